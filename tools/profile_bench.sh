#!/bin/bash
# Profiles bench.py on the GPU box: one rocprofv3 kernel-trace pass, then hardware counters in
# separate passes (one rocprofv3 run each, no tracing options next to --pmc).
# Usage (from the repository root, on a GPU box): tools/profile_bench.sh <tag>
# Leaves gpurun_out/<tag>trace/, gpurun_out/<tag>pmc{1,2,3}/ for tools/summarize_profile.py.
set -u
tag=${1:-prof_}
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
# frames of 128 spp (4 batches) keep the passes short; the counter passes skip the profiled pass and the ray batches
args="--spp 128 --steps 2 --warmup 1 --no-cpu-baseline --no-ray-batches --no-configs"
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}trace -o p -- python3 $root/bench.py $args > $out/${tag}trace.log 2>&1 || { echo "trace pass failed"; exit 1; }
grep '^{' $out/${tag}trace.log | tail -1 > $out/${tag}trace/bench.json
pass() {  # name, counters
  timeout -s KILL 150 rocprofv3 --pmc $2 --output-format csv -d $out/${tag}$1 -o p -- python3 $root/bench.py $args --no-profile > $out/${tag}$1.log 2>&1 || { echo "$1 failed"; exit 1; }
}
pass pmc1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pass pmc2 "FETCH_SIZE"
pass pmc3 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
echo done
