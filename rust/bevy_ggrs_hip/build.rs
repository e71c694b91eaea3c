// UN-BUILT SOURCE (no Rust toolchain in the build image).
// Links libggrs_hip.so; GGRS_HIP_LIB_DIR points at the directory holding it (bevy_ggrs_amd/ in this repo).
fn main() {
    if let Ok(dir) = std::env::var("GGRS_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=ggrs_hip");
    println!("cargo:rerun-if-env-changed=GGRS_HIP_LIB_DIR");
}
