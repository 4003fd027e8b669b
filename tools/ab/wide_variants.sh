#!/bin/bash
# A/B of builds of the library under build_ab/ on the wide walk's ray batches (GPU box, repository root).
# Usage: [RAYS_LOG2=22] tools/ab/wide_variants.sh <out log> <variant> [<variant> ...]   (variant = build_ab/liborion_hip_<variant>.so)
out=$1; shift
: > $out
for v in "$@"; do
  export ORION_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/liborion_hip_$v.so
  echo "== $v" >> $out
  RAYS_MODES=4 timeout -k 10 200 python -u tools/bench_rays.py ${SCENES:-nanosuit soup} 2>&1 | grep -E "wide" >> $out || exit 1
done
