"""The command-line front end (mirror of the reference's src/main.rs): archive header and option parsing on the CPU,
pass chains on the GPU."""
import struct

import numpy as np
import pytest

from rust_compress_amd import cli


def test_header_and_options():
    cfg = cli.parse_args(["app", "-block4096", "bwt", "mtf", "-x", "ari"])
    assert cfg["methods"] == ["bwt", "mtf", "ari"] and cfg["block_size"] == 4096 and not cfg["decompress"]
    assert cfg["warnings"] == ["Warning: unrecognized option: -x"]                     # main.rs:49
    assert cli.parse_args(["app", "-d"])["decompress"]
    hdr = cli.write_header(["bwt", "ari"])
    assert hdr == struct.pack("<I", 0x73632172) + bytes([2, 3]) + b"bwt" + bytes([3]) + b"ari"      # main.rs:166-171
    assert hdr[:4] == b"r!cs"
    assert cli.read_header(hdr + b"body") == (["bwt", "ari"], len(hdr))
    with pytest.raises(ValueError):
        cli.read_header(b"\x1f\x8b\x08\x00\x00")
    assert set(cli.PASSES) == {"dummy", "ari", "bwt", "mtf", "lz4"}                    # main.rs:72-124


@pytest.mark.gpu
def test_pass_chains_roundtrip(ctx, oracle):
    from rust_compress_amd import compress as cz, synth
    cz.set_context(ctx)
    data = synth.gen("text", 200000, 9).tobytes()
    for methods, bs in ((["dummy"], 65536), (["mtf"], 65536), (["ari"], 65536), (["lz4"], 65536), (["bwt"], 4096),
                        (["ari", "mtf", "bwt"], 65536), (["lz4", "ari", "mtf", "bwt"], 32768), (["dummy", "bwt", "dummy"], 100000)):
        arc = cli.encode(data, methods, bs)
        assert cli.read_header(arc)[0] == methods
        assert cli.decode(arc) == data, methods
    # single passes are the crate's stream formats: the body equals the oracle's encoding
    arc = cli.encode(data, ["mtf"])
    assert arc[cli.read_header(arc)[1]:] == oracle.mtf_encode(data)
    arc = cli.encode(data, ["ari"])
    assert arc[cli.read_header(arc)[1]:] == oracle.ari_byte_encode(data)
    arc = cli.encode(data[:10000], ["bwt"], 4096)
    body = arc[cli.read_header(arc)[1]:]
    assert struct.unpack_from("<I", body, 0)[0] == 4096                                # bwt/mod.rs:463 block size header
    L, origin = oracle.bwt_encode(data[:4096])
    assert body[4:8] == struct.pack("<I", 4096) and body[8:8 + 4096] == L and struct.unpack_from("<I", body, 8 + 4096)[0] == origin
    # the nesting order: the LAST listed method sees the raw input first (main.rs:172-179)
    arc = cli.encode(data[:5000], ["ari", "mtf"])
    assert arc[cli.read_header(arc)[1]:] == oracle.ari_byte_encode(oracle.mtf_encode(data[:5000]))
