#!/bin/bash
# Sweep of the two-lane overlap threshold on the bench step (unprofiled)
for v in "$@"; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --overlap-below $v > gpurun_out/ov_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/ov_$v.log; exit 1; }
  echo "overlap_below $v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ov_$v.log)"
done
