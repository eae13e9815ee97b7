// Links the prebuilt C-ABI library (graph_amd/libgraph_mi355x.so, built by hipcc for gfx950).
fn main() {
    let dir = std::env::var("GRAPH_MI355X_LIB_DIR").unwrap_or_else(|_| "../../graph_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=graph_mi355x");
    println!("cargo:rerun-if-env-changed=GRAPH_MI355X_LIB_DIR");
}
