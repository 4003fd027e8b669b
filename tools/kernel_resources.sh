#!/bin/bash
# Registers, spills, occupancy and LDS of the library's kernels as the compiler reports them.
# Usage: tools/kernel_resources.sh [pattern] [extra compiler flags ...]
pat=${1:-.}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -shared -Wno-unused-parameter "$@" \
    orion_amd/csrc/kernels.hip -o /tmp/kernel_resources.so -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       / VGPRs:/ {v=$(NF-1)} /VGPRs Spill:/ {vs=$(NF-1)} /SGPRs Spill:/ {ss=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
       /LDS Size/ {printf "%-90s VGPRs %3s spill %2s sgpr-spill %2s waves/SIMD %s LDS %s\n", substr(name,1,90), v, vs, ss, oc, $(NF-1)}' | grep -E "$pat"
