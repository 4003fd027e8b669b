#!/bin/bash
# Build a variant of the HIP library for A/B runs: tools/ab/build_variant.sh <name> [extra hipcc flags]
# -> build_ab/liborion_hip_<name>.so, the device assembly in /tmp/isa/<name>/, and the register / LDS
# figures of the kernels that walk the wide tree.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/build_ab /tmp/isa/$name
cd /tmp/isa/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -shared -Wno-unused-parameter \
    -save-temps=obj "$@" $root/orion_amd/csrc/kernels.hip -o /tmp/isa/$name/lib.so
cp lib.so $root/build_ab/liborion_hip_$name.so
S=kernels-hip-amdgcn-amd-amdhsa-gfx950.s
grep -B3 -A14 "\.name:.*\(k_query_wideILb0ELb0\|k_extend_wide\|k_shadeILi5ELb1\|k_shade_wave_brdf\)" $S | grep "name:\|vgpr_count\|vgpr_spill\|group_segment" | sed 's/_ZN12_GLOBAL__N_1//; s/EN9orion.*//; s/EvN9orion.*//'
awk '/^_ZN12_GLOBAL__N_112k_query_wideILb0ELb0EEE/{p=1} /^\.Lfunc_end/{if(p){p=0}} p' $S > qw.s
echo "k_query_wide<0,0>: valu $(grep -c '^\s*v_' qw.s) salu $(grep -c '^\s*s_' qw.s) ds $(grep -c '^\s*ds_' qw.s) branches $(grep -c 's_cbranch' qw.s)"
