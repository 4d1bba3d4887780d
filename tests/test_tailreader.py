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


def test_grow_caps_ends_at_the_largest_block_not_at_a_batch_error():
    """compress._grow_caps with a mocked decoder: slots grow 8x, the last attempt is 2^32 - 1 bytes (the largest slot
    run_batch accepts), and a block that still does not fit comes back with its own RCX_E_OUTPUT_TOO_SMALL status."""
    from rust_compress_amd import compress as Cm, _native as N

    class Res:
        def __init__(self, st): self.status = [st]
    seen = []

    def call(need):
        def f(cap):
            seen.append(cap)
            assert cap <= 0xFFFFFFFF, "a slot above 2^32 - 1 makes run_batch fail the whole batch"
            return Res(0 if cap >= need else N.E_OUTPUT_TOO_SMALL)
        return f
    assert Cm._grow_caps(call(3 << 30), 40 << 20).status[0] == 0            # 40 MB, 320 MB, 2.56 GB, then 2^32 - 1 (not 20 GB)
    assert seen == [40 << 20, 320 << 20, 2560 << 20, 0xFFFFFFFF]
    del seen[:]
    assert Cm._grow_caps(call(1 << 40), 1 << 16).status[0] == N.E_OUTPUT_TOO_SMALL and seen[-1] == 0xFFFFFFFF and len(set(seen)) == len(seen)
    del seen[:]
    assert Cm._grow_caps(call(10), 1 << 40).status[0] == 0 and seen == [0xFFFFFFFF]
