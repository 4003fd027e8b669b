#!/usr/bin/env python3
"""Headline benchmark: the Cornell-box BRDF path tracer at 1920x1080, 2048 spp, depth 100, 2 light samples
(the README render of the reference, BASELINE.json configs[1]) on N MI355X GPUs of one node.

A step is ONE FULL FRAME: 2048 samples of every pixel, split over the ranks by sample index (rank r renders
samples [2048 r / N, 2048 (r + 1) / N) of the whole image: strong scaling), each rank into a device-resident
accumulator, followed for N > 1 by one RCCL sum of the accumulators -- the only exchange the workload has.
The sum runs asynchronously while the next frame renders into a second accumulator; the timed region ends
when every sum has finished.  ms_per_step is therefore the frame's wall time, value = pixel-samples/s over
all ranks.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline      HBM roofline of the dominant kernel, from HIP-event timings of a separate profiled pass
                (one batch at a time, so that no launch shares the chip with another batch's)
  issue         the bound that actually holds: vector-instruction issue, against the guide's and the measured rate
  ray_batches   closest-hit Mrays/s of the ray-query entry point on R1/R2/R3 batches of five scenes
  configs       the other BASELINE.json configs (c0, c2, c3, c3b, c4) at their full size: a few frames each, samples/s,
                the dominant kernel and its algorithmic GB/s
  cpu_baseline  the oracle's render of a bounded sample of the same frame on the host cores
"""
import argparse
import hashlib
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
# Vector-instruction issue, wave64 instructions per second over 1024 SIMDs: the guide's figure (one per 2 cycles per
# SIMD at the nominal 2.4 GHz) and what tools/micro/valu_rate.hip sustains on the box with independent FMA chains,
# 8 waves per SIMD (one per 2.975 nominal cycles; in-kernel counters show why: under that load the shader clock
# drops to 1.86 GHz, which makes it one per 2.3 real cycles).
VALU_GUIDE_GINSTR_S = 1024 * 2.4 / 2.0
VALU_MEASURED_GINSTR_S = 1024 * 2.4 / 2.975
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


def framing_camera(model):
    """eye / look-at / up / yview / aspect of a camera on the +z side whose frame is the model's bounding box (the
    reference's camera: the frame is yview high at distance 1, orion/raytracer.cpp:212-238), so that most primary rays
    meet the model."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raygen
    lo, hi = raygen.scene_bounds(model)
    c = (lo + hi) / 2
    dx, dy = float(hi[0] - lo[0]), float(hi[1] - lo[1])
    dist = 1.02 * dy                      # yview = 1: the frame is as high as it is far away
    eye = np.array([c[0], c[1], hi[2] + dist], np.float32)
    return eye, c.astype(np.float32), np.array([0, 1, 0], np.float32), 1.0, max(0.05, dx / dy)


def source_sha():
    """One hash over the kernel sources: the per-ray / per-vertex constants of the counter passes in profiles/ belong to
    the sources they were taken with."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "orion_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def git_blob_sha(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def config_frames(api, device, budget_s=2.5):
    """The other BASELINE.json configs at their full size (configs/*.rtc, the derived scene files): frames of the config's
    own spp, as many as fit into about budget_s seconds, then one frame with per-kernel HIP events for the split."""
    import torch
    configs = [("c0_cornell_point_256", 16, 1, "configs[0]: Cornell box, point light, 256x256, 16 spp, depth 8"),
               ("c2_view_test_fhd", 256, 1, "configs[2]: view_test, 1920x1080, 256 spp"),
               ("c3_nanosuit_fhd", 512, 1, "configs[3]: nanosuit (19 058 triangles, PNG colour maps), 1920x1080, 512 spp, depth 0 as the asset has it"),
               ("c3b_nanosuit_bounce_fhd", 512, 1, "configs[3] with four mirror bounces: incoherent secondary rays through the deep trees"),
               ("c4_hindu_4k", 1024, 1, "configs[4]: hindu, 3840x2160, 1024 spp (one GPU renders the whole frame here)")]
    out = {"note": "one step = one full frame of the config (its own resolution, spp and depth) on one GPU, accumulator in HBM; "
                   "dominant kernel and algorithmic GB/s from one more frame with per-kernel HIP events, one batch at a time; "
                   "hbm_frac = algorithmic GB/s / 8 TB/s (64 B per closest-hit query, 64 B per shadow query + 96 B of path state per shaded vertex)"}
    for name, spp, ls, what in configs:
        fr = api.load_frame(os.path.join(ROOT, "configs", name + ".rtc"), load_textures=True)
        w, h, depth = fr.rtc.xres, fr.rtc.yres, max(0, fr.rtc.recursion_level)
        sc = api.Scene(fr.model, device=device)
        acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")

        def frame():
            sc.render(fr.camera12, w, h, 0, spp, ls, depth, seed=1, lights=fr.rtc.lights, accum=acc)
        frame()                                # untimed: the queues are sized on the first call
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frame()
        torch.cuda.synchronize(); one = time.perf_counter() - t0
        frames = int(max(2, min(50, budget_s / max(one, 1e-4))))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(frames):
            frame()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        sc.set_option("overlap_below", 0)
        sc.profile(True)
        frame()
        torch.cuda.synchronize()
        st = sc.profile_read()
        sc.profile(False)
        per_vertex = BYTES_PER_RAY * (st["shadow"]["items"] / max(1, st["shade"]["items"])) + BYTES_PER_VERTEX_STATE
        alg = {"closest_hit": st["closest_hit"]["items"] * BYTES_PER_RAY, "shade": st["shade"]["items"] * per_vertex}
        kern = {k: v for k, v in st.items() if v["launches"] and v["ms"] > 0}
        dom = max(kern, key=lambda k: kern[k]["ms"]) if kern else None
        res = {"workload": what, "resolution": [w, h], "spp": spp, "max_depth": depth, "lights": len(fr.rtc.lights),
               "triangles": int(fr.model.n_triangles), "frames": frames, "ms_per_frame": 1e3 * dt / frames,
               "samples_per_s": w * h * spp * frames / dt,
               "rays_per_sample": (st["closest_hit"]["items"] + st["shadow"]["items"]) / (w * h * spp),
               "kernel_ms": {k: round(v["ms"], 3) for k, v in kern.items()}, "dominant_kernel": dom}
        if dom in alg and kern[dom]["ms"] > 0:
            res["algorithmic_GBs"] = alg[dom] / (kern[dom]["ms"] * 1e-3) / 1e9
            res["hbm_frac"] = res["algorithmic_GBs"] / HBM_PEAK_GBS
        out[name] = res
        sc.close()
        del acc
    return out


def ray_batches(api, log2_rays=20, log2_large=22):
    """Closest-hit throughput of the ray-query entry point (device-resident rays, the library's HIP events) on the
    R1 / R2 / R3 batches of five scenes, with the HBM fraction of the 64 algorithmic bytes per ray."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    import raygen
    n = 1 << log2_rays
    out = {"rays_per_batch": n, "bytes_per_ray": BYTES_PER_RAY,
           "note": "R1 jittered primary rays of random pixels (Cornell: the README camera; other scenes: a camera on the +z side whose frame "
                   "is the model's bounding box), R2 cosine-weighted bounce rays "
                   "from their hit points in random order, R3 rays from those points towards uniform surface points; "
                   "Mrays/s = rays / average launch time over 5 launches (HIP events); hbm_frac = Mrays/s x 64 B / 8 TB/s"}
    for name in ("cornell", "view_test", "hindu", "nanosuit", "soup"):
        model = conftest.load_model(name)
        sc = api.Scene(model, device=torch.cuda.current_device())
        if name == "cornell":
            eye, look, up, yv = CAMERA
            aspect = W / H
        else:
            eye, look, up, yv, aspect = framing_camera(model)
        cam = api.camera_vectors(eye, look, up, yv, aspect)
        o1, d1 = raygen.primary_rays(cam, max(16, int(round(H * aspect))), H, n, seed=1)
        h = sc.closest_hit(o1, d1)
        nrm = np.zeros((n, 3), np.float32)
        for mi, m in enumerate(model.meshes):
            sel = np.where(h["mesh"] == mi)[0]
            if len(sel):
                nrm[sel] = m.vertices[m.indices[3 * h["tri"][sel]], 3:6]
        batches = {"R1": (o1, d1)}
        o2, d2 = raygen.bounce_rays(o1, d1, np.stack([h["mesh"], h["tri"]], 1), np.stack([h["t"], h["u"], h["v"]], 1), nrm, seed=2)
        if len(o2) > 1000:
            rep = int(np.ceil(n / len(o2)))
            perm = np.random.default_rng(5).permutation(n)
            o2 = np.tile(o2, (rep, 1))[:n][perm]; d2 = np.tile(d2, (rep, 1))[:n][perm]
            batches["R2"] = (o2, d2)
            batches["R3"] = raygen.rays_to_surface(o2, model, seed=3)
        res = {"triangles": int(model.n_triangles)}
        for bname, (o, d) in batches.items():
            to, td = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
            hits = torch.empty((n, 5), dtype=torch.int32, device="cuda")
            sc.closest_hit(to, td, hits)
            sc.profile(True)
            for _ in range(5):
                sc.closest_hit(to, td, hits)
            st = sc.profile_read()["closest_hit"]
            sc.profile(False)
            mrays = 5 * n / st["ms"] / 1e3
            res[bname] = {"Mrays_per_s": mrays, "hbm_frac": mrays * 1e6 * BYTES_PER_RAY / (HBM_PEAK_GBS * 1e9),
                          "hit_rate": float((hits[:, 0] != -1).float().mean())}
            if model.n_triangles > 2000:
                # Large scenes: also 2^22 rays (the batch tiled four times in a fresh random order).  Their walk hands rays to
                # lanes as they finish, and a batch has to be several times the 0.4 M lanes of the chip before the longest
                # rays (ten times the median number of steps) stop deciding when the launch ends.
                big = 1 << log2_large
                perm = torch.randperm(big, device="cuda") % n
                bo, bd = to[perm].contiguous(), td[perm].contiguous()
                bh = torch.empty((big, 5), dtype=torch.int32, device="cuda")
                sc.closest_hit(bo, bd, bh)
                sc.profile(True)
                for _ in range(5):
                    sc.closest_hit(bo, bd, bh)
                st = sc.profile_read()["closest_hit"]
                sc.profile(False)
                mr = 5 * big / st["ms"] / 1e3
                res[bname]["Mrays_per_s_4Mi"] = mr
                res[bname]["hbm_frac_4Mi"] = mr * 1e6 * BYTES_PER_RAY / (HBM_PEAK_GBS * 1e9)
        out[name] = res
        sc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # (20 frames: a timed region of about 8 s on one GPU)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of a frame (one step); default: the config's (c1: 2048, c4: 1024)")
    ap.add_argument("--config", default="c1", choices=["c1", "c4"],
                    help="c1 (default, the headline): Cornell FHD 2048 spp, BASELINE.json configs[1]; c4: hindu 3840x2160 1024 spp, configs[4]")
    ap.add_argument("--tiles", default="", metavar="NXxNY", help="shard a frame by image tile x sample chunk over the ranks (e.g. 4x2) "
                    "instead of by sample range alone; BASELINE.json configs[4] asks for it")
    ap.add_argument("--traversal", type=int, default=0, help="0 auto, 1 per-lane, 2 wave-synchronous")
    ap.add_argument("--tail-paths", type=int, default=-1, help="override the library's tail hand-over threshold (A/B runs)")
    ap.add_argument("--overlap-below", type=int, default=-1, help="override the library's two-lane overlap threshold (A/B runs; 0 = one lane)")
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE", help="scene option for A/B runs (orion_hip_scene_set_option)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled pass (per-kernel HIP-event timings, roofline)")
    ap.add_argument("--no-ray-batches", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (c0, c2, c3, c3b, c4)")
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

    from orion_amd import dist as odist
    global W, H, MAX_DEPTH, LIGHT_SAMPLES
    lights = None
    if args.config == "c4":
        fr = api.load_frame(os.path.join(ROOT, "configs", "c4_hindu_4k.rtc"), load_textures=True)
        model, cam, lights = fr.model, fr.camera12, fr.rtc.lights
        W, H, MAX_DEPTH, LIGHT_SAMPLES = fr.rtc.xres, fr.rtc.yres, max(0, fr.rtc.recursion_level), 1
        full_spp = 1024
    else:
        model = host.load_model(os.path.join(ROOT, "tests", "golden", "assets", "cornell-box", "cornell-box.obj"))
        eye, look, up, yv = CAMERA
        cam = api.camera_vectors(eye, look, up, yv, W / H)
        full_spp = FULL_SPP
    tiles = tuple(int(x) for x in args.tiles.lower().split("x")) if args.tiles else None
    if tiles is not None and (len(tiles) != 2 or min(tiles) < 1):
        raise SystemExit("--tiles wants NXxNY, e.g. 4x2")
    scene = api.Scene(model, device=local)
    scene.set_option("traversal", args.traversal)
    if args.tail_paths >= 0:
        scene.set_option("tail_paths", args.tail_paths)
    if args.overlap_below >= 0:
        scene.set_option("overlap_below", args.overlap_below)
    for kv in args.set:
        name, value = kv.split("=")
        scene.set_option(name, int(value))
    if args.share_device and world > 1:     # rehearsal: the ranks share one GPU's memory
        scene.set_option("batch_paths", max(1 << 22, (64 << 20) // world))
        scene.set_option("overlap_below", 4 << 20)
    accums = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2 if world > 1 else 1)]
    pending = [None] * len(accums)
    spp = args.spp or full_spp
    # This rank's share of every frame: a range of sample indices of the whole image, or (--tiles) its cells of the
    # tile x sample-chunk grid (orion_amd/dist.py: shard_plan).
    plan = odist.shard_plan(spp, W, H, rank, world, tiles)
    my_samples = sum((e - b) * ((r[2] - r[0]) * (r[3] - r[1]) if r else W * H) for r, b, e in plan)

    def render_into(accum, b, e, tile):
        scene.render(cam, W, H, b, e, LIGHT_SAMPLES, MAX_DEPTH, seed=1, lights=lights, tile=tile, accum=accum)

    def step(index):
        k = index % len(accums)
        if pending[k] is not None:       # the sum that last used this accumulator
            pending[k].wait(); pending[k] = None
        accums[k].zero_()
        # (every frame is rendered with the same sample indices: the frames of the timed region are repetitions)
        odist.render_plan(render_into, plan, accums[k])
        pending[k] = odist.reduce_async(accums[k])    # the frame's one exchange: sum of the radiance accumulators (None on one rank)

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

    samples = W * H * spp * args.steps
    value = samples / dt
    if args.config == "c4":
        what = ("hindu 3840x2160 point-light frame, %d spp, depth %d (BASELINE.json configs[4]); one step = one full frame" % (spp, MAX_DEPTH))
        metric = "samples/s (pixel-samples per second, hindu 3840x2160, %d-spp frames)" % spp
    else:
        what = ("cornell-box BRDF path tracer 1920x1080, %d spp, max depth 100, 2 light samples "
                "(BASELINE.json configs[1], the reference's README render); one step = one full frame" % spp)
        metric = "samples/s (pixel-samples per second, Cornell box 1920x1080 BRDF path tracer, %d-spp frames)" % spp
    out = {
        "source_sha": source_sha(),
        "metric": metric,
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / max(1, args.steps), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": (value / PUBLISHED_SAMPLES_PER_S) if args.config == "c1" else None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": what,
                   "spp": spp, "samples_this_rank": my_samples, "seed": 1, "rng": "philox4x32-10 counter",
                   "sharding": ("%dx%d image tiles x %d sample chunks dealt round-robin over the ranks" % (tiles[0], tiles[1], world) if tiles
                                else "sample-index ranges per rank") + " + one RCCL all-reduce of the accumulator per frame"},
        "frame_wall_s": dt / max(1, args.steps),
        "per_gpu_samples_per_s": value / world,
    }
    if rank == 0:
        E = sum(1 for m in model.meshes if np.any(model.materials[m.material].emissive != 0))
        if not args.no_profile and args.config == "c1":
            # The profiled passes: PROF_SPP samples of the same frame (four batches) with per-kernel HIP events, first on the timed region's own
            # schedule (two lanes: the next batch's head runs beside the current batch's thin tail, so a launch's duration
            # includes the time it shares the chip), then one batch at a time (no launch shares the chip: these are the
            # durations the roofline is computed from).
            PROF_SPP = 512
            scene.profile(True)
            accums[0].zero_()
            t1 = time.perf_counter()
            scene.render(cam, W, H, 0, PROF_SPP, LIGHT_SAMPLES, MAX_DEPTH, seed=1, accum=accums[0])
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            stats2 = scene.profile_read()
            scene.profile(False)
            scene.set_option("overlap_below", 0)
            scene.profile(True)
            accums[0].zero_()
            t1 = time.perf_counter()
            scene.render(cam, W, H, 0, PROF_SPP, LIGHT_SAMPLES, MAX_DEPTH, seed=1, accum=accums[0])
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t1
            stats = scene.profile_read()
            scene.profile(False)
            # units per kernel: k_extend processes rays; k_shade processes the vertices that hit
            vertices = stats["shadow"]["items"] // max(1, LIGHT_SAMPLES * E)
            units = {"closest_hit": ("k_extend", stats["closest_hit"]["items"], BYTES_PER_RAY),
                     "shade": ("k_shade", vertices, BYTES_PER_RAY * LIGHT_SAMPLES * E + BYTES_PER_VERTEX_STATE)}
            pmc, pmc_file = {}, None
            for f in ("r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
                try:   # HBM bytes per unit measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (see profiles/)
                    pmc = json.load(open(os.path.join(ROOT, "profiles", f)))
                    pmc_file = f
                    break
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
                        kern[kname]["pmc_hbm_GBs"] = per * items / launches / (avg_ms * 1e-3) / 1e9
            if kern:
                dom = max(kern, key=lambda k: kern[k]["ms"])
                traffic = kern[dom].get("pmc_hbm_bytes_per_launch")
                out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["algorithmic_GBs"], "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": kern[dom]["algorithmic_GBs"] / HBM_PEAK_GBS,
                                   "traffic": traffic,
                                   "traffic_GBs": kern[dom].get("pmc_hbm_GBs"),
                                   "traffic_frac": (kern[dom]["pmc_hbm_GBs"] / HBM_PEAK_GBS) if traffic else None,
                                   "binding_limit": "valu issue (see `issue`), not HBM",
                                   "note": "achieved / frac = ALGORITHMIC bytes per launch / average launch duration (HIP events on the library's "
                                           "stream) in a profiled pass of 512 spp of the same frame, one batch at a time.  For k_shade 128 of the 224 "
                                           "algorithmic bytes per vertex are the ray and hit records of its two light-sample queries, which are traced "
                                           "in registers and never reach HBM: traffic / traffic_GBs / traffic_frac = the HBM bytes the kernel really "
                                           "moves per launch (PMC passes in profiles/, FETCH_SIZE doubled per the gfx950 correction) and that as a rate "
                                           "and as a fraction of 8 TB/s -- the number to read as HBM utilisation."}
                if pmc_file:
                    cur = source_sha()
                    taken = pmc.get("source_sha")
                    out["pmc_constants"] = {"file": "profiles/" + pmc_file, "git_blob": git_blob_sha(os.path.join(ROOT, "profiles", pmc_file)),
                                            "taken_at_source_sha": taken, "current_source_sha": cur, "stale": taken != cur,
                                            "note": "bytes and vector instructions per ray / per vertex come from counter passes of an earlier run; "
                                                    "stale = the kernel sources have changed since (traffic and `issue` then describe the older kernels)"}
                out["kernels"] = kern
                ksum = sum(kern[k]["ms"] for k in kern) * 1e-3
                ksum2 = sum(stats2[n]["ms"] for n in units if stats2[n]["launches"]) * 1e-3
                out["profiled_pass"] = {"spp": PROF_SPP, "wall_s": dtp, "samples_per_s": W * H * PROF_SPP / dtp, "schedule": "one batch at a time",
                                        "kernel_sum_s": ksum, "projected_frame_s": dtp * spp / PROF_SPP}
                out["profiled_pass_two_lanes"] = {"spp": PROF_SPP, "wall_s": dt2, "samples_per_s": W * H * PROF_SPP / dt2, "schedule": "as the timed region",
                                                  "kernel_sum_s": ksum2, "projected_frame_s": dt2 * spp / PROF_SPP,
                                                  "note": "the timed frames run on this schedule: the next batch's first two bounces run beside the current "
                                                          "batch's thin tail.  A frame is therefore SHORTER than the one-at-a-time pass projects (its tails are "
                                                          "hidden), while the sum of its kernels' HIP-event durations is LONGER (overlapped launches each count "
                                                          "the time they share the chip): the roofline and `issue` use the one-at-a-time durations"}
                if all("valu_wave_instructions_per_item" in pmc.get(k, {}) for k in kern):
                    # what actually bounds the step: vector-instruction issue.  Instructions per ray / vertex from the counter
                    # passes in profiles/, over the kernels' own time in the profiled pass.
                    instr = sum(pmc[k]["valu_wave_instructions_per_item"] * kern[k]["units"] for k in kern)
                    busy_s = sum(kern[k]["ms"] for k in kern) * 1e-3
                    ach = instr / busy_s / 1e9
                    out["issue"] = {"bound": "valu issue", "achieved": ach, "unit": "G wave64 instructions/s",
                                    "peak_guide": VALU_GUIDE_GINSTR_S, "frac_guide": ach / VALU_GUIDE_GINSTR_S,
                                    "peak_measured": VALU_MEASURED_GINSTR_S, "frac_measured": ach / VALU_MEASURED_GINSTR_S,
                                    "note": "k_extend + k_shade vector instructions (counter passes in profiles/) over their summed launch time in the "
                                            "profiled pass; peak_guide = one wave64 instruction per 2 cycles per SIMD at 2.4 GHz, peak_measured = "
                                            "what independent FMA chains sustain on the box (tools/micro/valu_rate.hip: one per 2.975 nominal cycles, "
                                            "the shader clock dropping to 1.86 GHz under that load)"}
                rays = stats["closest_hit"]["items"] + stats["shadow"]["items"]
                out["rays"] = {"extend_rays": stats["closest_hit"]["items"], "light_sample_rays": stats["shadow"]["items"],
                               "per_sample": rays / (W * H * PROF_SPP),
                               "Mrays_per_s_per_gpu": rays / (W * H * PROF_SPP) * value / world / 1e6,
                               "note": "closest-hit queries per pixel-sample (profiled pass) x the timed region's samples/s per GPU; "
                                       "a light-sample query whose shading term is surely zero is answered without a walk and still counted"}
        if world == 1 and not args.no_ray_batches:      # (single-GPU figures: the other ranks would only wait for them)
            out["ray_batches"] = ray_batches(api)
        if world == 1 and not args.no_configs:
            out["configs"] = config_frames(api, local)
        if world == 1 and not args.no_cpu_baseline and args.config == "c1":
            out["cpu_baseline"] = cb = cpu_baseline(model, cam)
            if "rays" in out:   # the algorithm's work per closest-hit query (reference counters: orion/avx/sbvh.cpp:8-12)
                out["rays"]["box_tests_per_ray"] = cb["box_tests_per_sample"] / out["rays"]["per_sample"]
                out["rays"]["triangle_tests_per_ray"] = cb["triangle_tests_per_sample"] / out["rays"]["per_sample"]
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
