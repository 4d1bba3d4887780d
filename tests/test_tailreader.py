"""CPU suite: the pushback reader the decoders leave behind (no GPU needed)."""
import io

from rust_compress_amd.compress import TailReader


def test_tailreader_unread_and_topup():
    r = TailReader(io.BytesIO(b"0123456789"))
    assert r.read(3) == b"012"
    r.unread(b"12")
    assert r.read(1) == b"1" and r.read(4) == b"2345"        # tail first, then topped up from the inner reader
    r.unread(b"45"); r.unread(b"3")
    assert r.read(-1) == b"3456789" and r.read(5) == b"" and r.read(-1) == b""
    r.unread(b"")
    assert r.read(1) == b""
