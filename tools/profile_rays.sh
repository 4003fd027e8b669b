#!/bin/bash
# Hardware counters for the ray-query kernels on one scene (tools/bench_rays.py <scene>).
# Usage (repository root, GPU box): tools/profile_rays.sh <scene> <tag>
set -u
scene=${1:-nanosuit}; tag=${2:-rays_}
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}pmc1 -o p -- python3 $root/tools/bench_rays.py $scene > $out/${tag}pmc1.log 2>&1 || { echo "pmc1 failed"; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/${tag}pmc2 -o p -- python3 $root/tools/bench_rays.py $scene > $out/${tag}pmc2.log 2>&1 || { echo "pmc2 failed"; exit 1; }
echo done
