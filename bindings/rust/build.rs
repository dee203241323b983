// UNCOMPILED (no Rust toolchain in this image).  Links the prebuilt libsfx.so.
fn main() {
    let dir = std::env::var("SFX_LIB_DIR").unwrap_or_else(|_| "../../starframe_amd/csrc".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sfx");
    println!("cargo:rustc-link-lib=dylib=sfx_rccl"); // the slab halo exchange (include/sfx_rccl.h); needs librccl at run time
}
