#!/bin/bash
# usage: ab_sweep.sh tag lib...
tag=$1; shift
for lib in "" "$@"; do
  name=${lib:-default}
  echo "== $name"
  ORION_HIP_LIB=${lib:+$PWD/$lib} RAYS_MODES=4 timeout -k 10 150 python -u tools/bench_rays.py nanosuit soup 2>&1 | grep -E "R2|R3|R5" || exit 1
done
