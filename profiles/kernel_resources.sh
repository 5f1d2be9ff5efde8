#!/bin/bash
# Register, scratch and LDS use of the product kernel (trace_paths_kernel<0, true, false>) as the compiler reports it.
# usage: profiles/kernel_resources.sh [-DFLAG=VALUE ...]
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-memory-clause -Iinclude "$@" --cuda-device-only -S -o /tmp/pt_kernel_resources.s pathtracer_amd/csrc/pt_gpu.hip 2>/dev/null
awk '/^_ZN3ptd18trace_paths_kernelILi0ELb1ELb0EEEvNS_9TraceArgsE:/{f=1} f && /^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize)/{print} f && /^; Occupancy/{exit}' /tmp/pt_kernel_resources.s
awk '/\.name:.*trace_paths_kernelILi0ELb1ELb0EEE/{f=1} f && /(sgpr_spill_count|vgpr_spill_count|\.vgpr_count|\.sgpr_count)/{print} f && /vgpr_spill_count/{exit}' /tmp/pt_kernel_resources.s
