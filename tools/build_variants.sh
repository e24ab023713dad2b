#!/bin/bash
# Builds tuning variants of libpt_amd.so into build/variants/<name>.so
# usage: tools/build_variants.sh name1:"-DX=1 -DY=2" name2:"..."
set -e
cd "$(dirname "$0")/../path_tracer_amd/csrc"
mkdir -p ../../build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=default $flags -shared -o ../../build/variants/$name.so -x hip pt_api.cpp pt_group.cpp pt_host.cpp pt_kernels.hip pt_kernels_fast.hip -L/opt/rocm/lib -lrccl &
done
wait
ls -la ../../build/variants
