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


def test_the_crate_is_complete_on_paper():
    """What cannot be compiled here must at least be all there: the reference's feature table and [[bin]] (Cargo.toml:11-24), its module
    gates (src/lib.rs:19-50), the test application (src/main.rs) with the archive format and the passes of rust_compress_amd/cli.py,
    and the many-streams entry points of round 6."""
    ref = {"default": ["bwt", "checksum", "entropy", "flate", "lz4", "zlib", "rle"], "bwt": [], "checksum": [], "entropy": [], "flate": [], "lz4": [],
           "zlib": ["flate", "checksum"], "rle": [], "unstable": []}
    cargo = open(os.path.join(ROOT, "rust", "Cargo.toml")).read()
    feat = re.search(r"^\[features\]\n(.*?)(?=^\[|\Z)", cargo, flags=re.S | re.M).group(1)
    got = {m.group(1): re.findall(r'"([a-z0-9_]+)"', m.group(2)) for m in re.finditer(r"^([a-z0-9_]+)\s*=\s*\[(.*?)\]", feat, flags=re.M)}
    assert got == ref
    if os.path.exists("/root/reference/Cargo.toml"):                      # (the build container only: the reference's own table, parsed the same way)
        rc = open("/root/reference/Cargo.toml").read()
        rfeat = re.search(r"^\[features\]\n(.*?)(?=^\[|\Z)", rc, flags=re.S | re.M).group(1)
        assert got == {m.group(1): re.findall(r'"([a-z0-9_]+)"', m.group(2)) for m in re.finditer(r"^([a-z0-9_]+)\s*=\s*\[(.*?)\]", rfeat, flags=re.M)}
    binsec = re.search(r"^\[\[bin\]\]\n(.*?)(?=^\[|\Z)", cargo, flags=re.S | re.M).group(1)
    assert 'name = "compress"' in binsec and "doc = false" in binsec and 'path = "src/main.rs"' in binsec
    lib = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    for feature, item in (("bwt", "pub mod bwt;"), ("checksum", "pub mod checksum;"), ("entropy", "pub mod entropy;"), ("flate", "pub mod flate;"), ("lz4", "pub mod lz4;"),
                          ("rle", "pub mod rle;"), ("zlib", "pub mod zlib;"), ("checksum", "pub use checksum::adler::State32 as Adler32;")):
        assert re.search(r'#\[cfg\(feature\s*=\s*"%s"\)\]\s*\n%s' % (feature, re.escape(item)), lib), (feature, item)
    assert lib.count("pub fn into_inner") == 1                             # (two methods of one name do not compile)
    main = open(os.path.join(ROOT, "rust", "src", "main.rs")).read()
    from rust_compress_amd import cli
    assert "0x7363_2172" in main and cli.MAGIC == 0x73632172
    for name, (_, _, info) in cli.PASSES.items():
        assert '("%s", "%s")' % (name, info) in main, name
    for need in ("fn parse_args", "fn encode_pass", "fn decode_pass", "fn read_header", '"-block<N>"', "cfg.methods.iter().rev()", "Decompression methods are set in stone",
                 "Input is not a rust-compress archive", "bwt::Decoder::new(src, true)", "bwt::Encoder::new(Vec::new(), cfg.block_size)"):
        assert need in main, need
    t2 = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"//[^\n]*", "", main))
    assert t2.count("{") == t2.count("}") and t2.count("(") == t2.count(")")
    src = lambda f: open(os.path.join(ROOT, "rust", "src", f)).read()
    assert "pub fn decode_many(frames: &[&[u8]])" in src("lz4.rs") and "pub fn decode_many(streams: &[&[u8]])" in src("flate.rs") and "pub fn decode_many(members: &[&[u8]])" in src("zlib.rs")
    assert "pub(crate) fn decode_many_with" in lib
