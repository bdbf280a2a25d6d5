// Links libtinyrenderer_hip.so (built by `make -C tinyrenderer_amd/csrc`, hipcc --offload-arch=gfx950).
// TINYRENDERER_HIP_DIR names the directory that holds it; the directory is also put on the run-time search path of
// binaries built in this workspace so that `cargo run` finds the library without LD_LIBRARY_PATH.
fn main() {
    let dir = std::env::var("TINYRENDERER_HIP_DIR")
        .expect("set TINYRENDERER_HIP_DIR to the directory holding libtinyrenderer_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tinyrenderer_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=TINYRENDERER_HIP_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
