//! The crate's test application (reference: src/main.rs): archives of chained passes, block codecs on the GPU.
//!
//!     compress <options> <method1> .. <methodN>  < input  > archive
//!     compress -d                                 < archive > output
//!     options: -d (decompress), -block<N> (BWT block size, default 65536)        passes: dummy ari bwt mtf lz4
//!
//! Archive (main.rs:20, :166-171): u32 LE 0x73632172 ("r!cs"), u8 method count, per method u8 length + name, then the body.
//! The LAST listed method sees the raw input first (main.rs:172-179: each pass wraps the previous writer); decoding applies the
//! listed decoders from the archive outward (main.rs:145-152).  Mirrors rust_compress_amd/cli.py stage by stage, and shares its one
//! documented divergence: every writer is finished (the reference only flushes, main.rs:178-180, which truncates its own archives
//! whenever `ari`, `bwt`, `mtf` or `lz4` is in the chain).
//! The shim's codecs buffer a whole stream per FFI call anyway, so the chain runs stage by stage over byte vectors.
extern crate compress;

use compress::entropy::ari;
use compress::{bwt, lz4};
use std::io::{self, Read, Write};
use std::{env, process};

const MAGIC: u32 = 0x7363_2172; // "r!cs"
const PASSES: [(&str, &str); 5] = [
    ("dummy", "pass-through"),
    ("ari", "Adaptive arithmetic byte coder"),
    ("bwt", "Burrows-Wheeler Transformation"),
    ("mtf", "Move-To-Front Transformation"),
    ("lz4", "Ziv-Lempel derivative, focused at speed"),
];

struct Config {
    exe_name: String,
    methods: Vec<String>,
    block_size: usize,
    decompress: bool,
}

/// Config::query (main.rs:29-55): options start with '-', everything else names a pass
fn parse_args(mut args: impl Iterator<Item = String>) -> Config {
    let mut cfg = Config { exe_name: args.next().unwrap_or_else(|| "compress".into()), methods: Vec::new(), block_size: 1 << 16, decompress: false };
    for arg in args {
        match arg.strip_prefix('-') {
            Some(body) if body.starts_with("block") => cfg.block_size = body["block".len()..].parse().expect("-block<N>"),
            Some(body) if body.starts_with('d') => cfg.decompress = true,
            Some(_) => println!("Warning: unrecognized option: {}", arg),
            None => cfg.methods.push(arg),
        }
    }
    cfg
}

fn finished<W>(pair: (W, io::Result<()>)) -> io::Result<W> {
    pair.1.map(|_| pair.0)
}

fn encode_pass(name: &str, data: Vec<u8>, cfg: &Config) -> io::Result<Vec<u8>> {
    match name {
        "dummy" => Ok(data),
        "ari" => {
            let mut e = ari::ByteEncoder::new(Vec::new());
            e.write_all(&data)?;
            finished(e.finish())
        }
        "bwt" => {
            let mut e = bwt::Encoder::new(Vec::new(), cfg.block_size);
            e.write_all(&data)?;
            finished(e.finish())
        }
        "mtf" => {
            let mut e = bwt::mtf::Encoder::new(Vec::new());
            e.write_all(&data)?;
            Ok(e.finish())
        }
        "lz4" => {
            let mut e = lz4::Encoder::new(Vec::new());
            e.write_all(&data)?;
            finished(e.finish())
        }
        other => panic!("Pass {} is not implemented", other),
    }
}

fn decode_pass(name: &str, data: Vec<u8>) -> io::Result<Vec<u8>> {
    let src = io::Cursor::new(data);
    let mut out = Vec::new();
    match name {
        "dummy" => return Ok(src.into_inner()),
        "ari" => ari::ByteDecoder::new(src).read_to_end(&mut out)?,
        "bwt" => bwt::Decoder::new(src, true).read_to_end(&mut out)?,
        "mtf" => bwt::mtf::Decoder::new(src).read_to_end(&mut out)?,
        "lz4" => lz4::Decoder::new(src).read_to_end(&mut out)?,
        _ => panic!("Pass is not implemented"),
    };
    Ok(out)
}

fn header(methods: &[String]) -> Vec<u8> {
    let mut h = MAGIC.to_le_bytes().to_vec();
    h.push(methods.len() as u8);
    for m in methods {
        h.push(m.len() as u8);
        h.extend_from_slice(m.as_bytes());
    }
    h
}

/// -> (methods, offset of the body); None: not one of our archives
fn read_header(blob: &[u8]) -> Option<(Vec<String>, usize)> {
    if blob.len() < 5 || u32::from_le_bytes([blob[0], blob[1], blob[2], blob[3]]) != MAGIC {
        return None;
    }
    let mut p = 5;
    let mut methods = Vec::new();
    for _ in 0..blob[4] {
        let len = *blob.get(p)? as usize;
        methods.push(String::from_utf8(blob.get(p + 1..p + 1 + len)?.to_vec()).ok()?);
        p += 1 + len;
    }
    Some((methods, p))
}

fn main() {
    let cfg = parse_args(env::args());
    let mut input = Vec::new();
    if cfg.decompress {
        assert!(cfg.methods.is_empty(), "Decompression methods are set in stone");
        io::stdin().read_to_end(&mut input).expect("Unable to read input");
        let (methods, at) = match read_header(&input) {
            Some(x) => x,
            None => {
                eprintln!("Input is not a rust-compress archive");
                process::exit(1);
            }
        };
        let mut data = input.split_off(at);
        for m in &methods {
            data = decode_pass(m, data).expect("decode");
        }
        io::stdout().write_all(&data).unwrap();
    } else if cfg.methods.is_empty() {
        println!("rust-compress test application (MI355X)");
        println!("Usage:\n\t{} <options> <method1> .. <methodN> <input >output", cfg.exe_name);
        println!("Options:\n\t-d (to decompress)\n\t-block<N> (BWT block size)\nPasses:");
        for (name, info) in PASSES.iter() {
            println!("\t{} = {}", name, info);
        }
    } else {
        io::stdin().read_to_end(&mut input).expect("Unable to read input");
        let mut data = input;
        for m in cfg.methods.iter().rev() {
            data = encode_pass(m, data, &cfg).expect("encode");
        }
        let mut out = io::stdout();
        out.write_all(&header(&cfg.methods)).unwrap();
        out.write_all(&data).unwrap();
        out.flush().unwrap();
    }
}
