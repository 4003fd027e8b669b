#!/bin/bash
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in 1 2 4; do
  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/sc_$L -o p -- python3 $root/tools/shade_cost.py $L 8 > $out/sc_$L.log 2>&1 || { echo "L=$L failed"; exit 1; }
  grep '^{' $out/sc_$L.log
done
