// Links libcairom_hip.so.  CAIROM_HIP_DIR = directory holding the library built by
// `make -C cairo_m_amd/csrc` (default: ../../cairo_m_amd relative to this crate when vendored inside the HIP repo).
fn main() {
    let dir = std::env::var("CAIROM_HIP_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{manifest}/../../cairo_m_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=cairom_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=CAIROM_HIP_DIR");
}
