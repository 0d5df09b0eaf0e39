// Link against libaz_engine.so built by `python -c "import __graft_entry__ as g; g.build()"`.
fn main() {
    let dir = std::env::var("AZ_ENGINE_DIR").expect("set AZ_ENGINE_DIR to the directory holding libaz_engine.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=az_engine");
    println!("cargo:rerun-if-env-changed=AZ_ENGINE_DIR");
}
