#!/bin/bash
# A/B of builds of the library under build_ab/ on full-size configs (GPU box, repository root).
# Usage: tools/ab/config_variants.sh <out log> "<config names>" <variant> [<variant> ...]
out=$1; shift; cfgs=$1; shift
: > $out
for v in "$@"; do
  export ORION_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/liborion_hip_$v.so
  echo "== $v" >> $out
  timeout -k 10 200 python -u tools/bench_configs.py $cfgs 2>&1 | grep -E "^\| c" >> $out || exit 1
done
