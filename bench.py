#!/usr/bin/env python3
"""Headline benchmark: the Cornell-box BRDF path tracer at 1920x1080, depth 100, 2 light samples
(the README render of the reference, BASELINE.json configs[1]) on N MI355X GPUs of one node.

A step renders `--spp-per-step` samples of every pixel on each rank (ranks own disjoint sample
ranges; weak scaling) into a device-resident accumulator, followed for N > 1 by one RCCL sum of
the accumulators -- the only exchange the workload has.  The sum runs asynchronously while the next
step renders into a second accumulator; the timed region ends when every sum has finished.
value = pixel-samples/s over all ranks.

    python bench.py --gpus 1 --steps 4 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline      HBM roofline of the dominant kernel from live HIP-event timings
  cpu_baseline  the oracle's render of a bounded sample of the same frame on the host cores
  rays          traversal throughput (closest-hit queries/s) over the timed region
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
MAX_DEPTH, LIGHT_SAMPLES, FULL_SPP = 100, 2, 2048
CAMERA = ([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0)   # assets/cornell-box.rtc
PUBLISHED_SAMPLES_PER_S = 1920 * 1080 * 2048 / 5126.4                        # reference README.md:38-46
HBM_PEAK_GBS = 8000.0                                                        # MI355X HBM3E peak
VALU_PEAK_GINSTR_S = 1024 * 2.4 / 2.84                                       # measured sustained wave64 VALU issue rate, G/s
# Algorithmic queue bytes: 32 B ray record read + 32 B hit record written per closest-hit query;
# a shaded path vertex adds L*E such queries for its light samples and 48 B + 48 B of path state.
BYTES_PER_RAY = 64
BYTES_PER_VERTEX_STATE = 96


def cpu_baseline(model, cam, target_s=12.0):
    """Time the oracle (CPU restatement, OpenMP over pixels) on a scaled-down frame of the same scene,
    camera, depth and light samples; returns samples/s on this host's cores, the same on one
    thread, and the work counters (box and triangle tests per pixel-sample) of the algorithm."""
    from oracle import bindings as ob
    sc = ob.OracleScene(model)
    threads = min(ob.oracle_lib().oracle_max_threads(), len(os.sched_getaffinity(0)))
    w, h = 480, 270
    t0 = time.time()
    _, counters = sc.render(cam, w, h, 0, 4, LIGHT_SAMPLES, MAX_DEPTH, seed=1, threads=threads, count=True)
    probe = time.time() - t0
    rate = w * h * 4 / max(probe, 1e-6)
    spp = int(max(4, min(8192, target_s * rate / (w * h))))
    t0 = time.time()
    sc.render(cam, w, h, 0, spp, LIGHT_SAMPLES, MAX_DEPTH, seed=1, threads=threads)
    dt = time.time() - t0
    multi = w * h * spp / dt
    spp1 = int(max(1, 3.0 * multi / threads / (w * h)))     # about 3 s on one thread
    t0 = time.time()
    sc.render(cam, w, h, 0, spp1, LIGHT_SAMPLES, MAX_DEPTH, seed=1, threads=1)
    dt1 = time.time() - t0
    return {"value": multi, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{w}x{h} x {spp} spp of the same Cornell frame (depth {MAX_DEPTH}, {LIGHT_SAMPLES} light samples), {dt:.1f} s",
            "one_thread_samples_per_s": w * h * spp1 / dt1,
            "box_tests_per_sample": float(counters[0]) / (w * h * 4), "triangle_tests_per_sample": float(counters[1]) / (w * h * 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=128, help="samples per pixel per rank per step (32 = one 64 Mi-path batch at FHD)")
    ap.add_argument("--traversal", type=int, default=0, help="0 auto, 1 per-lane, 2 wave-synchronous")
    ap.add_argument("--tail-paths", type=int, default=-1, help="override the library's tail hand-over threshold (A/B runs)")
    ap.add_argument("--overlap-below", type=int, default=-1, help="override the library's two-lane overlap threshold (A/B runs; 0 = one lane)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not collect per-kernel HIP-event timings")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from orion_amd import api, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    model = host.load_model(os.path.join(ROOT, "tests", "golden", "assets", "cornell-box", "cornell-box.obj"))
    eye, look, up, yv = CAMERA
    cam = api.camera_vectors(eye, look, up, yv, W / H)
    scene = api.Scene(model, device=local)
    scene.set_option("traversal", args.traversal)
    if args.tail_paths >= 0:
        scene.set_option("tail_paths", args.tail_paths)
    if args.overlap_below >= 0:
        scene.set_option("overlap_below", args.overlap_below)
    accums = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2 if world > 1 else 1)]
    pending = [None] * len(accums)
    spp = args.spp_per_step

    def step(index):
        k = index % len(accums)
        if pending[k] is not None:       # the sum that last used this accumulator
            pending[k].wait(); pending[k] = None
        accums[k].zero_()
        base = (index * world + rank) * spp
        scene.render(cam, W, H, base, base + spp, LIGHT_SAMPLES, MAX_DEPTH, seed=1, accum=accums[k])
        if world > 1:                    # the frame's one exchange: sum of the radiance accumulators
            pending[k] = dist.all_reduce(accums[k], async_op=True)

    def fence():
        for k, w in enumerate(pending):
            if w is not None:
                w.wait(); pending[k] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    scene.profile(not args.no_profile)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stats = scene.profile_read()
    scene.profile(False)

    samples = W * H * spp * args.steps * world
    value = samples / dt
    out = {
        "metric": "samples/s (pixel-samples per second, Cornell box 1920x1080 BRDF path tracer)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": value / PUBLISHED_SAMPLES_PER_S, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cornell-box BRDF path tracer 1920x1080, max depth 100, 2 light samples "
                               "(BASELINE.json configs[1]); one step = %d spp of the full frame per GPU" % spp,
                   "spp_per_step_per_gpu": spp, "seed": 1, "rng": "philox4x32-10 counter",
                   "sharding": "sample-index ranges per rank + one RCCL all-reduce of the accumulator per step"},
        "projected_wall_s_2048spp": W * H * FULL_SPP / value,
    }
    if rank == 0:
        E = sum(1 for m in model.meshes if np.any(model.materials[m.material].emissive != 0))
        # units per kernel: k_extend processes rays; k_shade processes the vertices that hit
        vertices = stats["shadow"]["items"] // max(1, LIGHT_SAMPLES * E)
        units = {"closest_hit": ("k_extend", stats["closest_hit"]["items"], BYTES_PER_RAY),
                 "shade": ("k_shade", vertices, BYTES_PER_RAY * LIGHT_SAMPLES * E + BYTES_PER_VERTEX_STATE)}
        pmc = {}
        try:   # HBM bytes per unit measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (see profiles/)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
        except OSError:
            pass
        kern = {}
        for name, (kname, items, per_item) in units.items():
            st = stats[name]
            if st["launches"] and st["ms"] > 0:
                launches = st["launches"]
                alg_per_launch = items * per_item / launches
                avg_ms = st["ms"] / launches
                kern[kname] = {"launches": launches, "units": items, "ms": st["ms"], "avg_launch_ms": avg_ms,
                               "algorithmic_bytes_per_unit": per_item, "algorithmic_bytes_per_launch": alg_per_launch,
                               "algorithmic_GBs": alg_per_launch / (avg_ms * 1e-3) / 1e9}
                if kname in pmc:
                    per = pmc[kname]["hbm_read_bytes_per_item"] + pmc[kname]["hbm_write_bytes_per_item"]
                    kern[kname]["pmc_hbm_bytes_per_launch"] = per * items / launches
        if kern:
            dom = max(kern, key=lambda k: kern[k]["ms"])
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["algorithmic_GBs"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": kern[dom]["algorithmic_GBs"] / HBM_PEAK_GBS,
                               "traffic": kern[dom].get("pmc_hbm_bytes_per_launch"),
                               "note": "achieved = algorithmic bytes per launch / average launch duration (HIP events on the "
                                       "library's stream over the timed region); traffic = HBM bytes per launch from the PMC passes "
                                       "in profiles/ (FETCH_SIZE doubled per the gfx950 correction). The kernel is bound by "
                                       "instruction issue (one VALU instruction per 3.2 cycles per SIMD against a measured "
                                       "sustained 2.84), not by HBM.  Two batches are in flight at batch boundaries, so a launch's "
                                       "duration includes the time it shares the chip with the other batch's launches."}
            out["kernels"] = kern
            if all("valu_wave_instructions_per_item" in pmc.get(k, {}) for k in kern):
                # what actually bounds the step: vector-instruction issue.  Instructions per ray / vertex from the
                # counter passes in profiles/, the chip's sustained rate from tools/micro/valu_rate.hip
                # (one wave64 instruction per 2.84 cycles per SIMD, 1024 SIMDs at 2.4 GHz).
                instr = sum(pmc[k]["valu_wave_instructions_per_item"] * kern[k]["units"] for k in kern)
                out["issue"] = {"bound": "valu issue", "achieved": instr / dt / 1e9, "peak": VALU_PEAK_GINSTR_S, "unit": "G wave64 instructions/s",
                                "frac": instr / dt / 1e9 / VALU_PEAK_GINSTR_S,
                                "note": "k_extend + k_shade vector instructions of rank 0 over the timed region / wall time"}
            rays = stats["closest_hit"]["items"] + stats["shadow"]["items"]
            out["rays"] = {"extend_rays": stats["closest_hit"]["items"], "light_sample_rays": stats["shadow"]["items"],
                           "Mrays_per_s_per_gpu": rays / dt / 1e6,
                           "note": "closest-hit queries traced by rank 0 over the timed region / wall time"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cb = cpu_baseline(model, cam)
            if "rays" in out:   # the algorithm's work per closest-hit query (reference counters: orion/avx/sbvh.cpp:8-12)
                rays_per_sample = (out["rays"]["extend_rays"] + out["rays"]["light_sample_rays"]) / (W * H * spp * args.steps)
                out["rays"]["per_sample"] = rays_per_sample
                out["rays"]["box_tests_per_ray"] = cb["box_tests_per_sample"] / rays_per_sample
                out["rays"]["triangle_tests_per_ray"] = cb["triangle_tests_per_sample"] / rays_per_sample
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
