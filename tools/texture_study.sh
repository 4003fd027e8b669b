#!/bin/bash
# Texture-size sweep on a GPU box: wall time per size, then one counter pass per size for the L2
# hit rate of the shade kernel.  Output: gpurun_out/tex_<edge>.json, gpurun_out/texpmc_<edge>/.
set -u
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for e in 0 64 256 1024 2048 4096 8192; do
  timeout -k 10 120 python3 $root/tools/texture_study.py $e 8 > $out/tex_$e.json 2> $out/tex_$e.err || { echo "size $e failed"; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/texpmc_$e -o p -- python3 $root/tools/texture_study.py $e 8 > $out/texpmc_$e.log 2>&1 || { echo "pmc $e failed"; exit 1; }
  grep '^{' $out/tex_$e.json
done
echo done
