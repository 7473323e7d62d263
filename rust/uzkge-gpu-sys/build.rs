// Locates libuzkge_gpu.so.  UZKGE_GPU_LIB_DIR (the directory holding the library, normally
// <this repo>/uzkge_amd) wins; otherwise the in-tree build next to this crate is used.
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=UZKGE_GPU_LIB_DIR");
    let dir = env::var("UZKGE_GPU_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../uzkge_amd")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    if !dir.join("libuzkge_gpu.so").exists() {
        panic!(
            "libuzkge_gpu.so not found in {} -- build it with `make -C uzkge_amd/csrc` (hipcc, gfx950) or set UZKGE_GPU_LIB_DIR",
            dir.display()
        );
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=uzkge_gpu");
    // the library is not installed system-wide: embed its directory so binaries find it at run time
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
