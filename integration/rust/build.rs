// build.rs for addrianyy/path-tracer: compiles the gfx950 HIP library and links it.
// (Source only: there is no Rust toolchain in the build image, so this file has not been
// compiled -- the C++ harness host/main.cpp proves the same ABI end to end.)
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let csrc = PathBuf::from(env::var("PT_AMD_CSRC").unwrap_or_else(|_| "pt_amd/path_tracer_amd/csrc".into()));
    let lib = out.join("libpt_amd.so");
    let status = Command::new(env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into()))
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off",       // Rust never fuses a*b+c; neither may the GPU (bit parity)
               "-fno-slp-vectorize",      // packed fp32 ops are slower than scalar ones on gfx950
               "-x", "hip"])
        .arg(csrc.join("pt_api.cpp")).arg(csrc.join("pt_host.cpp")).arg(csrc.join("pt_kernels.hip"))
        .arg("-o").arg(&lib)
        .status().expect("hipcc not found");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=pt_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    for f in ["pt_api.cpp", "pt_host.cpp", "pt_kernels.hip", "pt_device.hpp", "pt_kernels.hpp"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
}
