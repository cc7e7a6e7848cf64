// Link against the in-tree build of the C ABI (sdk_amd/libspiral_hip.so; `make -C sdk_amd/csrc`).
// SPIRAL_HIP_LIB_DIR overrides the search path.
fn main() {
    let dir = std::env::var("SPIRAL_HIP_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../sdk_amd", std::env::var("CARGO_MANIFEST_DIR").unwrap()));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=spiral_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=SPIRAL_HIP_LIB_DIR");
}
