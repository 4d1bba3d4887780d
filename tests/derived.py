"""Loader for tests/golden/derived/ (written by tests/gen_derived_golden.py: a plain-Python transliteration of SURVEY Appendix A,
independent of oracle/*.c): the expected encoder-side bytes per (input, codec) -- in full for the small inputs, as length +
sha256 for the large ones.  Inputs are regenerated with rust_compress_amd.synth and checked against the recorded sha256."""
import hashlib
import json
import os
import struct

from rust_compress_amd import synth

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "derived")


def records(max_n=None):
    man = json.load(open(os.path.join(DIR, "manifest.json")))
    for rec in man["records"]:
        if max_n is not None and rec["n"] > max_n:
            continue
        data = synth.gen(rec["kind"], rec["n"], rec["seed"]).tobytes()
        assert hashlib.sha256(data).hexdigest() == rec["input_sha256"], "synthetic input %s is not what the golden files were made from" % rec["input"]
        yield rec, data


def check(rec, codec, got):
    """`got` (bytes) against the committed expectation for (rec, codec)"""
    e = rec["expect"][codec]
    got = bytes(got)
    path = os.path.join(DIR, "%s.%s.bin" % (rec["input"], codec))
    if os.path.exists(path):
        want = open(path, "rb").read()
        assert len(want) == e["len"] and hashlib.sha256(want).hexdigest() == e["sha256"]
        assert got == want, "%s %s: bytes differ from tests/golden/derived" % (rec["input"], codec)
    assert len(got) == e["len"] and hashlib.sha256(got).hexdigest() == e["sha256"], "%s %s: length / sha256 differ from tests/golden/derived" % (rec["input"], codec)


def ctx_bytes(ctx):
    """[(symbol, last_rank, distance_limit)] -> the layout of the committed dc_ctx vectors (and of include/rcx.h)"""
    return b"".join(struct.pack("<BBHI", s, r, 0, lim) for s, r, lim in ctx)
