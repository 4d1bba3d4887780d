"""Command-line front end with the reference application's archive format and pass chaining (src/main.rs:20, :126-181),
running the block codecs on the GPU through rust_compress_amd.compress.  SURVEY.md 8(f) rank 2.

    python -m rust_compress_amd.cli <options> <method1> .. <methodN>  < input > archive
    python -m rust_compress_amd.cli -d                                 < archive > output
    options: -d (decompress), -block<N> (BWT block size, default 65536)          passes: dummy ari bwt mtf lz4

Archive: u32 LE 0x73632172 ("r!cs"), u8 method count, per method u8 length + ASCII name (main.rs:166-171), then the body.
Encoding nests the writers so that the LAST listed method sees the raw input first (main.rs:172-179); decoding applies
the decoders in listed order from the archive outward (main.rs:145-152).

One documented divergence: the reference application never calls finish() on its writers (main.rs:178-180 only flush()es),
so with `ari`, `bwt`, `mtf` or `lz4` in the chain its own archives are truncated (no range-coder tail, unflushed block).
This front end finishes every writer, innermost first; archives it writes decode with it and -- the format being the
crate's own stream formats -- with any consumer of those formats.
"""
import io
import struct
import sys

from . import compress as cz

MAGIC = 0x73632172


class _Sink(io.BytesIO):
    """write() target that survives being wrapped by several encoders"""


PASSES = {
    # name: (encoder factory (w, cfg) -> writer, decoder factory (r, cfg) -> reader, info)            main.rs:72-124
    "dummy": (lambda w, c: w, lambda r, c: r, "pass-through"),
    "ari": (lambda w, c: cz.entropy.ari.ByteEncoder(w), lambda r, c: cz.entropy.ari.ByteDecoder(r), "Adaptive arithmetic byte coder"),
    "bwt": (lambda w, c: cz.bwt.Encoder(w, c["block_size"]), lambda r, c: cz.bwt.Decoder(r, True), "Burrows-Wheeler Transformation"),
    "mtf": (lambda w, c: cz.bwt.mtf.Encoder(w), lambda r, c: cz.bwt.mtf.Decoder(r), "Move-To-Front Transformation"),
    "lz4": (lambda w, c: cz.lz4.Encoder(w), lambda r, c: cz.lz4.Decoder(r), "Ziv-Lempel derivative, focused at speed"),
}


def parse_args(argv):
    """Config::query, main.rs:29-55: options start with '-', everything else is a method name"""
    cfg = {"exe_name": argv[0] if argv else "app", "methods": [], "block_size": 1 << 16, "decompress": False, "warnings": []}
    for arg in argv[1:]:
        if arg.startswith("-"):
            body = arg[1:]
            if body.startswith("block"):
                cfg["block_size"] = int(body[len("block"):])
            elif body.startswith("d"):
                cfg["decompress"] = True
            else:
                cfg["warnings"].append("Warning: unrecognized option: %s" % arg)
        else:
            cfg["methods"].append(arg)
    return cfg


def write_header(methods):
    out = struct.pack("<IB", MAGIC, len(methods))
    for m in methods:
        out += bytes([len(m)]) + m.encode("ascii")
    return out


def read_header(blob):
    """-> (methods, body offset); raises ValueError on a foreign archive"""
    if len(blob) < 5 or struct.unpack_from("<I", blob, 0)[0] != MAGIC:
        raise ValueError("Input is not a rust-compress archive")
    n, p, methods = blob[4], 5, []
    for _ in range(n):
        ln = blob[p]
        methods.append(blob[p + 1:p + 1 + ln].decode("utf-8"))
        p += 1 + ln
    return methods, p


def encode(data, methods, block_size=1 << 16):
    cfg = {"block_size": block_size}
    sink = _Sink()
    sink.write(write_header(methods))
    chain, w = [], sink
    for m in methods:                                   # main.rs:172-177: each pass wraps the previous writer
        if m not in PASSES:
            raise KeyError("Pass %s is not implemented" % m)
        w = PASSES[m][0](w, cfg)
        chain.append(w)
    if w is sink:
        sink.write(data)
    else:
        w.write(data)
    for enc in reversed(chain):                         # finish the writers, the one nearest to the input first
        if enc is not sink and hasattr(enc, "finish"):
            enc.finish()
    return sink.getvalue()


def decode(blob):
    methods, p = read_header(blob)
    cfg = {"block_size": 1 << 16}
    r = io.BytesIO(blob[p:])
    for m in methods:                                   # main.rs:145-152
        if m not in PASSES:
            raise KeyError("Pass is not implemented")
        r = PASSES[m][1](r, cfg)
    return r.read() if isinstance(r, io.BytesIO) else r.read_to_end()


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    cfg = parse_args(argv)
    for w in cfg["warnings"]:
        print(w)
    if cfg["decompress"]:
        assert not cfg["methods"], "Decompression methods are set in stone"
        try:
            sys.stdout.buffer.write(decode(sys.stdin.buffer.read()))
        except ValueError as e:
            print(e, file=sys.stderr)
            return 1
    elif not cfg["methods"]:
        print("rust-compress test application (MI355X)")
        print("Usage:\n\t%s <options> <method1> .. <methodN> <input >output" % cfg["exe_name"])
        print("Options:\n\t-d (to decompress)\n\t-block<N> (BWT block size)\nPasses:")
        for name, (_, _, info) in PASSES.items():
            print("\t%s = %s" % (name, info))
    else:
        sys.stdout.buffer.write(encode(sys.stdin.buffer.read(), cfg["methods"], cfg["block_size"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
