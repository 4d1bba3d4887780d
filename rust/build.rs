// Links librcx.so (built in-tree by `python __graft_entry__.py`: hipcc --offload-arch=gfx950).
fn main() {
    let dir = std::env::var("RCX_LIB_DIR").unwrap_or_else(|_| "../rust_compress_amd/csrc".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rcx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
