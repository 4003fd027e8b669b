#!/usr/bin/env python3
"""Per-ray summary of tools/profile_walk.sh's counter passes for the closest-hit query kernel.
Usage: summarize_walk.py <gpurun_out/tag> <rays per launch>"""
import collections, csv, glob, sys

prefix, rays = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(prefix + "_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_closest_api" in r["Kernel_Name"] or "k_query_wide" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
L = max(n.values()) if n else 1
print(f"launches counted: {L}")
for k in sorted(agg):
    print(f"- {k}: {agg[k] / n[k]:.4g} per launch")
g = lambda k: agg[k] / max(1, n[k])
if "SQ_INSTS_VALU" in agg:
    print(f"- VALU wave instructions per ray: {g('SQ_INSTS_VALU') / rays:.1f}; SALU {g('SQ_INSTS_SALU') / rays:.1f}; vector loads {g('SQ_INSTS_VMEM_RD') / rays:.2f}; LDS {g('SQ_INSTS_LDS') / rays:.2f}")
    print(f"- waves resident per SIMD (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 4 per CU...): wave-cycles {g('SQ_WAVE_CYCLES'):.4g}, GRBM_GUI_ACTIVE {g('GRBM_GUI_ACTIVE'):.4g}")
if "SQ_WAIT_ANY" in agg:
    print(f"- SQ_WAIT_ANY / SQ_WAVE_CYCLES: {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}; SQ_WAIT_INST_ANY share {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}; ACTIVE_INST_VALU share {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f}")
if "TCP_TOTAL_CACHE_ACCESSES_sum" in agg:
    print(f"- L1 (TCP) accesses per ray: {g('TCP_TOTAL_CACHE_ACCESSES_sum') / rays:.1f}; L1->L2 read requests per ray: {g('TCP_TCC_READ_REQ_sum') / rays:.1f}; L2 hit rate {g('TCC_HIT_sum') / max(1, g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
for f in glob.glob(prefix + "_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_closest_api" in r["Name"] or "k_query_wide" in r["Name"]:
            print(f"- kernel trace: {r['Name'][:60]} calls {r['Calls']} avg {float(r['AverageNs']) / 1e3:.1f} us")
