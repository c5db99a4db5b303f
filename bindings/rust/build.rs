// Links libfastlanes_amd.so when the `gpu` feature is enabled.  (Uncompiled: see README.md.)
fn main() {
    if std::env::var("CARGO_FEATURE_GPU").is_ok() {
        let dir = std::env::var("FASTLANES_AMD_LIB_DIR")
            .expect("set FASTLANES_AMD_LIB_DIR to the directory containing libfastlanes_amd.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=fastlanes_amd");
        println!("cargo:rerun-if-env-changed=FASTLANES_AMD_LIB_DIR");
    }
}
