// Builds libparticles_hip.so with hipcc for gfx950 and links it.
// NOT COMPILED OR TESTED here (no Rust toolchain in this repository's build
// environment).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let repo = manifest.join("../..");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("libparticles_hip.so");
    let status = Command::new(&hipcc)
        .args(["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c++17", "-shared"])
        .arg(format!("-I{}", repo.join("include").display()))
        .arg("-o")
        .arg(&lib)
        .arg(repo.join("particles_amd/csrc/particles_hip.hip"))
        .status()
        .expect("failed to run hipcc");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=particles_hip");
    for f in ["particles_amd/csrc/particles_hip.hip", "particles_amd/csrc/step_kernels.hpp",
              "particles_amd/csrc/reorder_kernels.hpp", "particles_amd/csrc/brush_kernels.hpp",
              "include/particles_hip.h"] {
        println!("cargo:rerun-if-changed={}", repo.join(f).display());
    }
}
