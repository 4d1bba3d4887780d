"""CPU suite: the Rust shim crate under rust/ is source only (no Rust toolchain in the image), so it is kept honest
mechanically: every export of include/rcx.h is declared in rust/src/rcx_sys.rs with the same argument count, the
#[repr(C)] structs have the header's fields in the header's order, the enum constants carry the header's values, and
every export the shim modules call is declared.  (The stream logic is transcribed from the tested C++ twin,
rust_compress_amd/host/compress.hpp.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def _header():
    h = _strip_c_comments(open(os.path.join(ROOT, "include", "rcx.h")).read())
    funcs = {}
    for m in re.finditer(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?\**\s+\**(rcx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.M):
        name, args = m.group(1), m.group(2).strip()
        funcs[name] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(rcx_[a-z_]+)\s*\{(.*?)\}\s*\1\s*;", h, flags=re.S):
        fields = [re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", f.strip()).group(1) for f in m.group(2).split(";") if f.strip()]
        structs[m.group(1)] = fields
    enums = {}
    for m in re.finditer(r"enum\s+rcx_[a-z]+\s*\{(.*?)\}\s*;", h, flags=re.S):
        val = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                val = int(v, 0)
            else:
                k, val = item, val + 1
            enums[k] = val
    return funcs, structs, enums


def _rust():
    s = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "src", "rcx_sys.rs")).read())
    funcs = {}
    ext = re.search(r'extern\s+"C"\s*\{(.*)\}', s, flags=re.S).group(1)
    for m in re.finditer(r"pub\s+fn\s+(rcx_[a-z0-9_]+)\s*\(([^)]*)\)", ext):
        args = m.group(2).strip()
        funcs[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub\s+struct\s+(rcx_[a-z_]+)\s*\{(.*?)\}", s, flags=re.S):
        structs[m.group(1)] = [f.group(1) for f in re.finditer(r"(?:pub\s+)?([a-z_][a-z0-9_]*)\s*:", m.group(2))]
    consts = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"pub\s+const\s+(RCX_[A-Z0-9_]+)\s*:\s*[a-z0-9_]+\s*=\s*(-?[0-9xa-fA-F]+)\s*;", s)}
    return funcs, structs, consts


def test_every_export_is_declared_with_the_same_arity():
    hf, _, _ = _header()
    rf, _, _ = _rust()
    assert len(hf) >= 36, sorted(hf)
    assert set(hf) == set(rf), (sorted(set(hf) - set(rf)), sorted(set(rf) - set(hf)))
    assert {k: hf[k] for k in hf} == {k: rf[k] for k in hf}


def test_header_exports_match_the_ctypes_list_and_the_library():
    from rust_compress_amd import _native as N
    hf, _, _ = _header()
    assert set(N.EXPORTS) == set(hf)


def test_repr_c_structs_have_the_headers_fields_in_order():
    _, hs, _ = _header()
    _, rs, _ = _rust()
    assert hs["rcx_batch"] == rs["rcx_batch"] and len(hs["rcx_batch"]) == 11
    assert hs["rcx_dev_batch"] == rs["rcx_dev_batch"] and len(hs["rcx_dev_batch"]) == 11
    assert rs["rcx_ctx"] == ["_private"]


def test_enum_constants_carry_the_headers_values():
    _, _, he = _header()
    _, _, rc = _rust()
    assert len(he) >= 50
    missing = [k for k in he if k not in rc]
    assert not missing, missing
    assert all(rc[k] == v for k, v in he.items()), [(k, v, rc[k]) for k, v in he.items() if rc[k] != v]
    assert rc["RCX_W_EMPTY_BLOCK_MIDSTREAM"] == 1


def test_shim_modules_only_call_declared_exports_and_cover_the_crates_surface():
    rf, _, _ = _rust()
    src = os.path.join(ROOT, "rust", "src")
    used = set()
    text = {}
    for dp, _, fs in os.walk(src):
        for f in fs:
            if f.endswith(".rs") and f != "rcx_sys.rs":
                t = open(os.path.join(dp, f)).read()
                text[os.path.relpath(os.path.join(dp, f), src)] = t
                used |= set(re.findall(r"\b(rcx_[a-z0-9_]+)\s*\(", t))
    assert used <= set(rf), sorted(used - set(rf))
    # the reference's public names, one per module
    want = {"lz4.rs": ["pub fn decode_block", "pub fn encode_block", "pub fn compression_bound", "pub struct Decoder", "pub struct Encoder"],
            "flate.rs": ["pub struct Decoder", "pub fn eof", "pub fn reset"], "zlib.rs": ["pub struct Decoder", "pub fn unwrap"],
            "bwt/mod.rs": ["pub fn encode_simple", "pub fn decode_simple", "pub struct Encoder", "pub struct Decoder", "pub fn encode(", "pub fn decode(",
                           "pub struct TransformIterator", "pub fn get_origin", "pub struct InverseIterator", "fn flush"],
            "bwt/mtf.rs": ["pub struct Encoder", "pub struct Decoder", "pub struct MTF", "pub r: TailReader<R>"],
            "bwt/dc.rs": ["pub fn encode_simple", "pub fn decode_simple", "pub struct Context", "pub symbol", "pub last_rank", "pub distance_limit",
                          "pub fn encode(", "pub fn decode(", "pub const TOTAL_SYMBOLS"],
            # the per-symbol surface of mod.rs:67-293 with its methods (host code), next to the per-stream device codecs
            "entropy/ari/mod.rs": ["pub struct ByteEncoder", "pub struct ByteDecoder", "pub mod bin", "pub mod table", "pub mod apm",
                                   "pub struct RangeEncoder", "pub fn process(&mut self, total: Border, from: Border, to: Border, output: &mut [Symbol]) -> usize",
                                   "pub fn query(&self, total: Border, code: Border) -> Border", "pub fn get_code_tail", "pub fn reset", "pub threshold",
                                   "pub trait Model<V: Copy>", "fn get_range(&self, value: V) -> (Border, Border)", "fn find_value(&self, offset: Border) -> (V, Border, Border)",
                                   "fn get_denominator(&self) -> Border", "pub struct Encoder<W>", "pub struct Decoder<R>",
                                   "pub fn encode<V: Copy, M: Model<V>>(&mut self, value: V, model: &M) -> io::Result<()>",
                                   "pub fn decode<V: Copy, M: Model<V>>(&mut self, model: &M) -> io::Result<V>", "pub const RANGE_DEFAULT_THRESHOLD"],
            "entropy/ari/bin.rs": ["pub struct Model", "pub fn new_flat(threshold: Border, rate: Border)", "pub fn new_custom", "pub fn reset_flat", "pub fn update(&mut self, value: bool)",
                                   "pub fn get_probability_zero", "pub fn get_probability_one", "impl AriModel<bool> for Model", "pub struct SumProxy", "impl<'a> AriModel<bool> for SumProxy<'a>",
                                   "rcx_ari_binary_encode_batch", "rcx_ari_binary_decode_batch"],
            "entropy/ari/table.rs": ["pub struct Model", "pub fn new_flat(num_values: usize, threshold: Border)", "pub fn new_custom", "pub fn reset_flat",
                                     "pub fn update(&mut self, value: usize, add_log: usize, add_const: Border)", "pub fn downscale", "pub fn get_frequencies",
                                     "impl AriModel<usize> for Model", "pub struct SumProxy", "impl<'a> AriModel<usize> for SumProxy<'a>",
                                     "rcx_ari_proxy_encode_batch", "rcx_ari_proxy_decode_batch"],
            "entropy/ari/apm.rs": ["pub struct Bit", "pub fn to_wide", "pub fn from_wide", "pub fn new_equal", "impl AriModel<bool> for Bit", "pub struct Gate", "pub fn pass(&self, bit: &Bit) -> (Bit, BinCoords)",
                                   "pub fn pass_wide", "pub fn update(&mut self, value: bool, bc: BinCoords, rate: isize, bias: isize)", "rcx_ari_apm_encode_batch", "rcx_ari_apm_decode_batch"],
            "rle.rs": ["pub struct Encoder", "pub struct Decoder", "in_run", "&buf[1..]"],
            "checksum/adler.rs": ["pub struct State32"],
            "lib.rs": ["pub struct TailReader", "pub use checksum::adler::State32 as Adler32", "impl<R: Read> std::ops::Deref for TailReader<R>", "pub fn into_inner"]}
    for f, names in want.items():
        for nm in names:
            assert nm in text[f], (f, nm)
    for f in ("entropy/ari/table.rs", "entropy/ari/bin.rs", "entropy/ari/apm.rs"):
        assert not re.search(r"pub struct \w+;", text[f]), f          # no marker types: the reference's names carry the reference's methods
    # braces balance in every file (a cheap syntax sanity check without a compiler)
    for f, t in text.items():
        t2 = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"//[^\n]*", "", t))
        assert t2.count("{") == t2.count("}") and t2.count("(") == t2.count(")"), f
