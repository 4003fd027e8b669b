#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats CSV + PMC counter_collection CSVs) into a small
per-kernel text summary for profiles/.  Usage: summarize_profile.py <gpurun_out prefix> <out.md> [traffic.json]
With a third argument also writes the HBM bytes per ray / per shaded vertex that bench.py reports
as roofline.traffic (needs the run's bench.json for the unit counts)."""
import collections
import csv
import glob
import json
import os
import sys


def kname(n):
    for k in ("k_shade", "k_extend", "k_tail", "k_resolve", "k_closest_api", "k_anyhit_api"):
        if k in n:
            return k
    return None


def main():
    prefix, out = sys.argv[1], sys.argv[2]
    lines = []
    trace = glob.glob(prefix + "trace/**/*kernel_stats.csv", recursive=True)
    if trace:
        lines.append("## rocprofv3 --kernel-trace --stats (kernel, calls, total ms, average us, % of GPU time)\n")
        for r in csv.DictReader(open(trace[0])):
            lines.append(f"- `{r['Name'][:70]}` calls {r['Calls']}  total {int(r['TotalDurationNs']) / 1e6:.3f} ms  avg {float(r['AverageNs']) / 1e3:.1f} us  {float(r['Percentage']):.2f} %")
        b = prefix + "trace/bench.json"
        if os.path.exists(b):
            d = json.load(open(b))
            lines.append(f"\nbench.py under the profiler: {d['value'] / 1e9:.3f} Gsamples/s, HIP-event kernel totals: " +
                         ", ".join(f"{k} {v['ms']:.3f} ms / {v['launches']} launches" for k, v in d.get("kernels", {}).items()))
            # bench.py's roofline comes from its profiled pass, the last launches of the run (one batch at a time, no
            # overlap between the two lanes): the profiler's durations of those same launches, for comparison.
            kt = glob.glob(prefix + "trace/**/*kernel_trace.csv", recursive=True)
            if kt:
                spans = collections.defaultdict(list)
                for r in csv.DictReader(open(kt[0])):
                    k = kname(r["Kernel_Name"])
                    if k:
                        spans[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
                for k, v in d.get("kernels", {}).items():
                    last = [x[1] for x in sorted(spans.get(k, []))[-v["launches"]:]]
                    if len(last) == v["launches"]:
                        lines.append(f"- {k}: the profiled pass's {len(last)} launches average {sum(last) / len(last) / 1e3:.1f} us in the profiler's trace, "
                                     f"{v['avg_launch_ms'] * 1e3:.1f} us by bench.py's HIP events")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for f in glob.glob(prefix + "pmc*/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if not k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (f, r["Dispatch_Id"]) not in seen:
                seen.add((f, r["Dispatch_Id"]))
        for r in csv.DictReader(open(f)):
            pass
    if agg:
        lines.append("\n## PMC counters summed over all dispatches of the run (separate passes per counter group)\n")
        for k, v in agg.items():
            lines.append(f"### {k}")
            for c, x in sorted(v.items()):
                lines.append(f"- {c}: {x:.6g}")
            if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
                # FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
                rd = v.get("FETCH_SIZE", 0) * 1024 * 2
                wr = v.get("WRITE_SIZE", 0) * 1024
                lines.append(f"- HBM traffic (FETCH_SIZE x2 per the gfx950 correction, + WRITE_SIZE): read {rd / 1e9:.3f} GB, write {wr / 1e9:.3f} GB")
            if "TCC_HIT_sum" in v:
                lines.append(f"- L2 hit rate: {v['TCC_HIT_sum'] / max(1, v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.4f}")
            if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
                lines.append(f"- per wave: {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f} VALU, {v['SQ_INSTS_SALU'] / v['SQ_WAVES']:.0f} SALU, {v['SQ_INSTS_SMEM'] / v['SQ_WAVES']:.0f} SMEM instructions; "
                             f"waiting share SQ_WAIT_ANY/SQ_WAVE_CYCLES = {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.2f}")
            lines.append("")
    if len(sys.argv) > 3 and os.path.exists(prefix + "trace/bench.json"):
        d = json.load(open(prefix + "trace/bench.json"))
        # The counter passes run the same command as the traced one, and the counters also see the
        # warm-up steps, which bench.py's unit counts leave out.
        # bench.py's unit counts are those of its profiled pass (profiled_pass.spp samples of the frame); the counter passes run
        # steps + warmup frames of config.spp samples and no profiled pass.
        scale = (d["steps"] + d["warmup"]) * d["config"]["spp"] / float(d.get("profiled_pass", {}).get("spp", 128))
        units = {"k_extend": d["kernels"]["k_extend"]["units"] * scale, "k_shade": d["kernels"]["k_shade"]["units"] * scale}
        note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --spp 128 --steps 2 --warmup 1; FETCH_SIZE doubled "
                "(gfx950 counts half the bytes of wide coalesced reads); item = ray for k_extend, shaded vertex for k_shade")
        traffic = {}
        for k, n in units.items():
            if k in agg and "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k] and n:
                traffic[k] = {"hbm_read_bytes_per_item": agg[k]["FETCH_SIZE"] * 1024 * 2 / n,
                              "hbm_write_bytes_per_item": agg[k]["WRITE_SIZE"] * 1024 / n, "items": n, "note": note}
                if "SQ_INSTS_VALU" in agg[k]:   # wave64 vector instructions issued per item
                    traffic[k]["valu_wave_instructions_per_item"] = agg[k]["SQ_INSTS_VALU"] / n
        traffic["source_sha"] = d.get("source_sha")   # the kernel sources these per-item figures were taken with (bench.py: source_sha)
        json.dump(traffic, open(sys.argv[3], "w"), indent=1)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
