#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path on MI355X.

One "step" = one complete render of BASELINE.json configs[1]: the book1 scene,
1920x1080, 512 samples per pixel, depth 50, through the C ABI (cr_render_device).
Inputs (scene, camera) are resident in HBM before the timed region; the output
stays in HBM.  With N GPUs the 512 sample indices are split across ranks
(rank r renders [r*512/N, (r+1)*512/N) of every pixel), the per-pixel f32 sums
are reduced to rank 0 with one RCCL reduce and divided by 512 there -- a fixed
job, so scaling is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` uses the counted algorithmic-bytes model
of DESIGN.md and the kernel time from HIP events on the launch stream;
`cpu_baseline` times the f64 oracle (a port of the reference, NOT the Rust binary)
on a bounded slice of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def algorithmic_bytes(st, width, height, entry_bytes=32, prim_bytes=16):
    """DESIGN.md / SURVEY.md 8(d): B = S*(2*48) + N*entry + P*s_prim + T*4 + W*H*12."""
    return (st["segments"] * 96 + st["node_tests"] * entry_bytes + st["prim_tests"] * prim_bytes +
            st["texel_fetches"] * 4 + width * height * 12)


def cpu_baseline(scene, seed, target_s=14.0):
    """Time the f64 oracle on sample indices [0, k) of the same workload (all pixels)."""
    from crucible_amd import _abi as A
    from oracle.oracle import Oracle
    threads = min(os.cpu_count() or 1, 16)
    o = Oracle(A.CR_REAL_F64)
    h = o.scene_create(scene.flatten())
    cam = scene.scene_cam
    try:
        k, dt = 1, 0.0
        for _ in range(4):   # grow the slice until it is >= ~10 s of CPU work (bounded at ~30 s)
            t0 = time.perf_counter()
            o.render(h, cam, seed=seed, sample_begin=0, sample_count=k, output_sum=True, n_threads=threads)
            dt = time.perf_counter() - t0
            if dt >= 0.8 * target_s or k >= cam.samples:
                break
            k = max(k + 1, min(cam.samples, int(k * target_s / max(dt, 1e-3))))
    finally:
        o.scene_destroy(h)
    n = cam.image_width * cam.image_height * k
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"f64 oracle, same scene {cam.image_width}x{cam.image_height}, sample indices [0,{k}) of {cam.samples} "
                      f"for every pixel ({n / 1e6:.2f} Msamples, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--real", choices=["f32", "f64"], default="f32")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["book1", "teapot", "million", "movie"], default="book1",
                    help="book1 = BASELINE configs[1] (the headline); teapot/million/movie = configs[2]/[3]/[4], extra lines")
    ap.add_argument("--bvh", choices=["reference", "sah", "ordered", "lbvh"], default="reference",
                    help="reference = the reference's median-split tree (parity mode, the headline); sah = the quality "
                         "builder of SURVEY 8(f) row 1 (same walk, other topology); ordered = that tree walked near child first. "
                         "Both are extra lines, not the headline")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N>1 (nccl = RCCL; gloo to rehearse)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from crucible_amd import _abi as A
    from crucible_amd.demo_builder import (book1_end_scene, load_teapot, million_spheres, procedural_sky,
                                           teapot_orbit_movie)
    from crucible_amd.distributed import reduce_to_mean, shard_range
    from crucible_amd.renderer import Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    real_type = A.CR_REAL_F32 if args.real == "f32" else A.CR_REAL_F64
    tdtype = torch.float32 if args.real == "f32" else torch.float64
    seed, scene_seed = 0xC0FFEE, 1
    prim_bytes = 16 if args.real == "f32" else 32
    frame_sharded = False
    if args.workload == "book1":
        scene = book1_end_scene(1, scene_seed=scene_seed, image_width=args.width, samples=args.spp)
        wl = "book1 (RTIOW final scene, seeded) {W}x{H} @ {spp} spp, depth {d} -- BASELINE.json configs[1]"
    elif args.workload == "teapot":
        spp_default = args.spp if args.spp != 512 else 1024
        scene = load_teapot(1, image_width=args.width, samples=spp_default, sky=procedural_sky())
        wl = "teapot.obj (6320 tris) + ground + procedural 2048x1024 env map {W}x{H} @ {spp} spp, depth {d} -- configs[2]"
        prim_bytes = 36 if args.real == "f32" else 72
    elif args.workload == "million":
        width = args.width if args.width != 1920 else 3840
        scene = million_spheres(1, scene_seed=scene_seed, image_width=width, samples=args.spp if args.spp != 512 else 256)
        wl = "1,000,001 procedural spheres {W}x{H} @ {spp} spp, depth {d} -- configs[3]"
    else:
        scene = teapot_orbit_movie(1, image_width=args.width, samples=args.spp)
        wl = "teapot orbit movie (240 frames at 24 fps), one frame per rank per step, {W}x{H} @ {spp} spp, depth {d} -- configs[4]"
        prim_bytes = 36 if args.real == "f32" else 72
        frame_sharded = True
    scene.bvh_mode = {"sah": A.CR_BVH_SAH, "ordered": A.CR_BVH_SAH_ORDERED, "lbvh": A.CR_BVH_LBVH}.get(args.bvh, A.CR_BVH_REFERENCE)
    cam = scene.scene_cam
    W, H, spp = cam.image_width, cam.image_height, cam.samples
    if frame_sharded:
        s_begin, s_count = 0, spp          # every rank renders whole frames; no collective
    else:
        s_begin, s_count = shard_range(rank, world, spp)
    reduce = world > 1 and not frame_sharded

    r = Renderer(dev_index)
    r.upload_scene(scene.flatten())
    out = torch.empty((H, W, 3), dtype=tdtype, device=dev)
    step_no = [0]

    def step():
        if frame_sharded:
            cam.frame = (step_no[0] * world + rank) % 240
            step_no[0] += 1
        r.render_device(cam, out.data_ptr(), seed=seed, real_type=real_type, sample_begin=s_begin, sample_count=s_count,
                        output_sum=reduce)
        ms = r.last_kernel_ms()          # waits for the launch (HIP events on the library's stream)
        if reduce:
            if args.backend == "nccl":
                reduce_to_mean(out, spp, dst=0)   # RCCL reduce of the RGB sums, then sum / count on rank 0
            else:                                  # rehearsal backends reduce a host copy
                host = out.cpu()
                reduce_to_mean(host, spp, dst=0)
                out.copy_(host)
        return ms

    # one counted launch (untimed) for the algorithmic-bytes model; also warms the build path
    st = r.render_device(cam, out.data_ptr(), seed=seed, real_type=real_type, sample_begin=s_begin, sample_count=s_count,
                         output_sum=reduce, want_stats=True)
    for _ in range(args.warmup):
        step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        kernel_ms += step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_samples = W * H * spp * args.steps * (world if frame_sharded else 1)
        value = total_samples / elapsed / 1e6
        entry_bytes = 32 if args.real == "f32" else 64
        B = algorithmic_bytes(st, W, H, entry_bytes, prim_bytes)
        k_ms = kernel_ms / max(args.steps, 1)
        achieved = B / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.workload}_{W}x{H}_spp{s_count}_{args.real}" + ("" if args.bvh == "reference" else "_" + args.bvh)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        valu = None
        vpath = os.path.join(ROOT, "profiles", "valu.json")
        if os.path.exists(vpath):
            try:
                valu = json.load(open(vpath)).get(f"{args.workload}_{W}x{H}_spp{s_count}_{args.real}" +
                                                  ("" if args.bvh == "reference" else "_" + args.bvh))
            except Exception:
                valu = None
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "Msamples/sec (whole node), book1 1920x1080"
        rec = {
            "metric": metric, "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak" if frame_sharded else "strong", "vs_baseline": None, "dtype": args.real, "data": "synthetic",
            "config": {"workload": wl.format(W=W, H=H, spp=spp, d=cam.max_depth),
                       "image": [W, H], "spp": spp, "max_depth": cam.max_depth, "scene_seed": scene_seed, "rng_seed": seed,
                       "primitives": len(scene.elements), "bvh_entries": st["bvh_entries"],
                       "bvh": "reference topology (median split, bvhwrapper.rs:46-78)" if args.bvh == "reference" else
                              {"sah": "binned SAH topology (CR_BVH_SAH; not the reference's tree)",
                               "ordered": "binned SAH topology walked near child first (CR_BVH_SAH_ORDERED; not the reference's tree or order)",
                               "lbvh": "Morton-code LBVH built on the device (CR_BVH_LBVH; not the reference's tree)"}[args.bvh],
                       "scene_residency": {0: "L2", 1: "whole scene in LDS", 2: "BVH top levels in LDS"}.get(st["scene_in_lds"]),
                       "parallelism": "1 GPU" if world == 1 else (f"frame-shard x{world}, no collective" if frame_sharded else
                                                                     f"spp-shard x{world} + {args.backend} reduce of the RGB sums")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "cr::pathtrace_kernel (kernel_ms also covers the ~2 ms ordered-sum kernel cr::sg_finalize_kernel)",
                         "kernel_ms": round(k_ms, 4),
                         "algorithmic_bytes_per_launch": int(B),
                         "counters_per_launch": {k: st[k] for k in ("samples", "segments", "node_tests", "prim_tests",
                                                                    "texel_fetches")}},
            "valu": valu,   # VALU busy vs the measured issue peak and lane utilisation, from the committed PMC passes (or null)
            "kernel_msamples_per_s": round(W * H * s_count / (k_ms * 1e-3) / 1e6, 2) if k_ms > 0 else None,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload != "million":
            rec["cpu_baseline"] = cpu_baseline(scene, seed)
        print(json.dumps(rec), flush=True)

    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
