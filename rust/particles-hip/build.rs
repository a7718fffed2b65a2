// Builds libparticles_hip.so with hipcc for gfx950 and links it.
// NOT COMPILED OR TESTED here (no Rust toolchain in this repository's build
// environment); the source list below is checked against the Makefile's by
// tests/test_rust_binding.py.
use std::{env, path::PathBuf, process::Command};

const SOURCES: [&str; 4] = [
    "particles_amd/csrc/particles_hip.hip",
    "particles_amd/csrc/pair_sort.hip",
    "particles_amd/csrc/particles_comm.hip",
    "particles_amd/csrc/authoring.cpp",
];

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let repo = manifest.join("../..");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("libparticles_hip.so");
    let mut cmd = Command::new(&hipcc);
    cmd.args(["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c++17", "-shared"])
        .arg(format!("-I{}", repo.join("include").display()))
        .arg("-o")
        .arg(&lib);
    for s in SOURCES {
        cmd.arg(repo.join(s));
    }
    cmd.arg("-ldl"); // RCCL itself is loaded at run time
    let status = cmd.status().expect("failed to run hipcc");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=particles_hip");
    for f in SOURCES.iter().copied().chain([
        "particles_amd/csrc/step_kernels.hpp",
        "particles_amd/csrc/reorder_kernels.hpp",
        "particles_amd/csrc/brush_kernels.hpp",
        "particles_amd/csrc/handle_access.hpp",
        "include/particles_hip.h",
    ]) {
        println!("cargo:rerun-if-changed={}", repo.join(f).display());
    }
}
