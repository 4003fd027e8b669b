#!/bin/bash
# Hardware counters of the closest-hit ray-query kernel for one scene, one batch and one walk.
# Usage (repository root, GPU box): tools/profile_walk.sh <scene> "<batch name>" <traversal option> <tag>
# Three rocprofv3 passes (counter groups of their own) + one kernel trace; outputs under gpurun_out/<tag>*.
set -u
scene=$1; batch=$2; mode=$3; tag=$4
root=$(pwd); out=$root/gpurun_out
export RAYS_ONLY="$batch" RAYS_MODES=$mode
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc1 -o p -- python3 $root/tools/bench_rays.py $scene > $out/${tag}_pmc1.log 2>&1 || { echo "pmc1 failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_pmc2 -o p -- python3 $root/tools/bench_rays.py $scene > $out/${tag}_pmc2.log 2>&1 || { echo "pmc2 failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/${tag}_pmc3 -o p -- python3 $root/tools/bench_rays.py $scene > $out/${tag}_pmc3.log 2>&1 || { echo "pmc3 failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o p -- python3 $root/tools/bench_rays.py $scene > $out/${tag}_trace.log 2>&1 || { echo "trace failed"; exit 1; }
echo done
