#!/bin/bash
# A/B of alternative builds of the kernel library on the bench step (unprofiled)
for v in "$@"; do
  export ORION_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/$v.so
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/ab_$v.log 2>&1 || { echo "$v failed"; exit 1; }
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$v.log)"
done
