#!/bin/bash
# Instruction counters of the closest-hit kernel on one ray batch of tools/bench_rays.py:
#   walk_cost.sh "R4 tile tri"
root=$(pwd); out=$root/gpurun_out
export RAYS_ONLY="$1" RAYS_LOG2=22
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/wc -o p -- python3 $root/tools/bench_rays.py cornell > $out/wc.log 2>&1 || { echo failed; tail -5 $out/wc.log; exit 1; }
python3 - <<PY
import csv, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open("$out/wc/p_counter_collection.csv")):
    if "k_closest_api" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
for i in ids[-1:]:
    c = rows[i]
    print(i, {k: round(v / c["SQ_WAVES"], 1) for k, v in c.items() if k != "SQ_WAVES"}, "waves", int(c["SQ_WAVES"]))
PY
