#!/bin/bash
for t in 0 4096 16384 65536 131072 262144 524288; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --tail-paths $t > gpurun_out/tail_$t.log 2>&1 || { echo "$t failed"; exit 1; }
  echo "$t $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tail_$t.log)"
done
