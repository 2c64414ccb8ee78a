// Links the in-tree C-ABI library.  FLUIDSIM_LIB_DIR = directory holding libfluidsim_hip.so.
fn main() {
    let dir = std::env::var("FLUIDSIM_LIB_DIR").unwrap_or_else(|_| "..".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=fluidsim_hip");
}
