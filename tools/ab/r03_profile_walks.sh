#!/bin/bash
# The round's walk measurements in one GPU call (repository root, GPU box): grazing stress with the guard, lane
# occupancy with and without it (a build with counters: build_ab/liborion_hip_stats.so), hardware counters per batch.
set -u
out=gpurun_out
STRESS_KINDS="grazing" timeout -k 10 400 python -u tools/ab/stress_wide.py 30 nanosuit soup hindu > $out/r03f_stress_guard.log 2>&1 || { echo "stress failed"; exit 1; }
ORION_HIP_LIB=$PWD/build_ab/liborion_hip_stats.so timeout -k 10 200 python -u tools/ab/wide_stats.py nanosuit soup > $out/r03f_wide_stats_guard.log 2>&1 || { echo "stats failed"; exit 1; }
ORION_HIP_LIB=$PWD/build_ab/liborion_hip_stats.so WIDE_GUARD=0 timeout -k 10 200 python -u tools/ab/wide_stats.py nanosuit soup > $out/r03f_wide_stats_noguard.log 2>&1 || { echo "stats2 failed"; exit 1; }
tools/profile_walk.sh nanosuit "R2 bounce" 4 r03f_n2g || exit 1
tools/profile_walk.sh nanosuit "R3 shadow" 4 r03f_n3g || exit 1
tools/profile_walk.sh soup "R2 bounce" 4 r03f_s2g || exit 1
WIDE_GUARD=0 tools/profile_walk.sh nanosuit "R2 bounce" 4 r03f_n2 || exit 1
WIDE_GUARD=0 tools/profile_walk.sh nanosuit "R3 shadow" 4 r03f_n3 || exit 1
echo all done
