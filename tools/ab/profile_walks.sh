#!/bin/bash
# The wide walk's measurements in one GPU call (repository root, GPU box): grazing stress with the guard, lane
# occupancy with and without it (a build with counters: build_ab/liborion_hip_stats.so, tools/ab/build_variant.sh stats
# -DORION_WIDE_STATS=1), hardware counters per batch.  Usage: tools/ab/profile_walks.sh [tag = walks_]
set -u
out=gpurun_out; tag=${1:-walks_}
STRESS_KINDS="grazing" timeout -k 10 400 python -u tools/ab/stress_wide.py 30 nanosuit soup hindu > $out/${tag}stress_guard.log 2>&1 || { echo "stress failed"; exit 1; }
ORION_HIP_LIB=$PWD/build_ab/liborion_hip_stats.so timeout -k 10 200 python -u tools/ab/wide_stats.py nanosuit soup > $out/${tag}wide_stats_guard.log 2>&1 || { echo "stats failed"; exit 1; }
ORION_HIP_LIB=$PWD/build_ab/liborion_hip_stats.so WIDE_GUARD=0 timeout -k 10 200 python -u tools/ab/wide_stats.py nanosuit soup > $out/${tag}wide_stats_noguard.log 2>&1 || { echo "stats2 failed"; exit 1; }
tools/profile_walk.sh nanosuit "R2 bounce" 4 ${tag}n2g || exit 1
tools/profile_walk.sh nanosuit "R3 shadow" 4 ${tag}n3g || exit 1
tools/profile_walk.sh soup "R2 bounce" 4 ${tag}s2g || exit 1
WIDE_GUARD=0 tools/profile_walk.sh nanosuit "R2 bounce" 4 ${tag}n2 || exit 1
WIDE_GUARD=0 tools/profile_walk.sh nanosuit "R3 shadow" 4 ${tag}n3 || exit 1
echo all done
