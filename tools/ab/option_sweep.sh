#!/bin/bash
# Sweep of one bench option on the bench step (unprofiled): option_sweep.sh --group-min 0 1048576 ...
opt=$1; shift
for v in "$@"; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile $opt $v > gpurun_out/sweep_$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/sweep_$v.log; exit 1; }
  echo "$opt $v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/sweep_$v.log)"
done
