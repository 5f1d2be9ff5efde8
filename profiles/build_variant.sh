#!/bin/bash
# Builds build_variants/libpt_<name>.so: the library with extra -D flags on the GPU file (A/B measurements).
# usage: profiles/build_variant.sh <name> [-DFLAG=VALUE ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_variants
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-memory-clause -Iinclude "$@" -c -o build_variants/pt_gpu_$name.o pathtracer_amd/csrc/pt_gpu.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_variants/libpt_$name.so pathtracer_amd/csrc/host_core.o pathtracer_amd/csrc/bvh_build.o pathtracer_amd/csrc/scenes.o pathtracer_amd/csrc/sah_build.o pathtracer_amd/csrc/presplit.o build_variants/pt_gpu_$name.o -pthread
rm build_variants/pt_gpu_$name.o
