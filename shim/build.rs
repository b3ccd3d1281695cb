// Link against libdfgpu.so.  DFGPU_LIB_DIR = directory holding the library (datafusion-upstream_amd/ in this repository).
fn main() {
    let dir = std::env::var("DFGPU_LIB_DIR").unwrap_or_else(|_| "../datafusion-upstream_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=dfgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=DFGPU_LIB_DIR");
}
