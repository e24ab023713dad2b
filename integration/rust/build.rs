// build.rs for addrianyy/path-tracer: compiles the gfx950 HIP library and links it.
// (Source only: there is no Rust toolchain in the build image, so this file has not been compiled.
// The same library, built with the same flags by path_tracer_amd/csrc/Makefile, is driven through the same
// ABI by the C++ host -- tests/test_gpu_host.py builds host/main.cpp, runs it on the GPU and checks its
// output.png against the oracle.)
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
        .arg(csrc.join("pt_api.cpp")).arg(csrc.join("pt_group.cpp")).arg(csrc.join("pt_host.cpp"))
        .arg(csrc.join("pt_kernels.hip")).arg(csrc.join("pt_kernels_fast.hip"))
        .args(["-L/opt/rocm/lib", "-lrccl"])      // pt_group_*: ncclCommInitAll + ncclReduce over xGMI
        .arg("-o").arg(&lib)
        .status().expect("hipcc not found");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=pt_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    for f in ["pt_api.cpp", "pt_group.cpp", "pt_host.cpp", "pt_kernels.hip", "pt_kernels_fast.hip", "pt_device.hpp",
              "pt_kernels.hpp", "pt_context.hpp"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
}
