#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path on MI355X.

One "step" = one complete render of BASELINE.json configs[1]: the book1 scene, 1920x1080, 512 samples per pixel,
depth 50, through the C ABI, in the reference's arithmetic (f64, src/utils.rs:72-74).  Inputs (scene, camera) are
resident in HBM before the timed region; the output stays in HBM.  With N GPUs the 512 sample indices are split across
ranks (rank r renders [r*512/N, (r+1)*512/N) of every pixel), the per-pixel sums are added on rank 0 by ONE RCCL
reduce issued inside the library (cr_group_render) and divided by 512 there -- a fixed job, so scaling is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.

Summation order: the library default (CR_SUM_RELAXED: the same paths, attenuations multiplied in path order, samples added
as 64-bit fixed point -- within 1e-12 of the reference order, equal work counters; include/crucible_hip.h).  The line
carries the reference-order (bit-exact parity mode) rate of the same frame as `reference_order` and the largest
per-channel difference between the two frames, measured on the device in this run.  `--sum-order reference` makes the
parity mode the measured one.

`roofline`: the path is a VALU-bound pointer-chasing walk (book1 lives in LDS), so the bound is
the vector ALU.  achieved = ALGORITHMIC floating-point operations per launch / live kernel time, where the
operation count is the reference algorithm's own arithmetic per unit of work (ALGO_FLOPS below: one IEEE add, sub, mul,
div, sqrt, compare, min, max or floor on the scalar type = 1; integer and control work = 0) times the work counters
of the launch (segments, box tests, primitive tests, texel fetches, samples -- equal to the oracle's in the parity
tests).  peak = the vector FLOP/s of the scalar type (guide: 157.3 TFLOP/s f32; f64 is half rate) which counts an FMA
as two -- the path may not fuse (parity forbids contraction), so half of it is the reachable ceiling; both are in the
line.  The algorithmic HBM bytes of SURVEY 8(d) and the PMC counters of the committed profile are secondary keys, the
latter labelled as coming from profiles/.

`cpu_baseline`: the f64 oracle (a port of the reference, NOT the Rust binary) on all host cores, on a bounded slice of
the same workload; `cpu_baseline_faithful` is the same port evaluating timelines, update_bb and material clones the way
the reference does at every hit.  Baselines, not targets.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.65}   # guide: 157.3 TFLOP/s f32 vector (FMA = 2); f64 vector is half rate
# What the chip SUSTAINS on this pool (scripts/calib/valu_peak.hip, profiles/r03_valu_calib.txt): whatever the occupancy, a SIMD
# retires one wave64 f64 add / mul / max per ~2.1 ns and one f32 add per ~1.2 ns (the clock drops as more waves issue), i.e.
# 1024 SIMDs x 64 lanes / 2.1 ns = 31 T f64 operations/s without FMA -- the ceiling a non-fusing f64 kernel can reach.
VALU_SUSTAINED_TOPS = {"f32": 1024 * 64 / 1.2e-9 / 1e12, "f64": 1024 * 64 / 2.1e-9 / 1e12}

WHY_VALU = {
    "book1": "book1 sits in LDS whole (tree, spheres, materials, f32 screening records): no global-memory read in the walk; HBM carries "
             "24 bytes per pixel of sums; the vector ALU under lane divergence is what binds",
    "teapot": "the 8191-wrapper tree is read through L2 below the 2048-wrapper LDS window (hit rate ~88 %): ~45 % of wave time waits on "
              "dependent 32-byte fetches at 4 waves/SIMD, the rest is the vector ALU under lane divergence; HBM is idle",
    "million": "1,048,575 wrappers: the walk makes ~80 box tests per segment, ~55 of them below the LDS window as dependent 32-byte "
               "fetches through L2 / Infinity Cache (the 32 MB of screening records exceed an XCD's 4 MB L2): memory latency at 4 waves/SIMD "
               "binds first, the vector ALU second; HBM bandwidth is not the bound",
    "movie": "the teapot frame with a keyed camera (per-sample camera basis from the keyframes): as the teapot frame, plus ~100 f64 "
             "operations per sample of camera set-up",
}

# Floating-point operations of the REFERENCE algorithm per unit of work (DESIGN.md section 4 derives each line):
ALGO_FLOPS = {
    "node_test": 24,        # Aabb::hit (bvh.rs:96-132): per axis 2 sub, 2 mul, t0<t1, max, min, max<=min
    "sphere_test": 26,      # Sphere::hit (sphere.rs:72-95): oc 3, h 5, c 7, disc 3, disc<0, sqrt, one root (sub, div, 2 compares) on ~half the tests
    "triangle_test": 45,    # Triangle::hit (triangle.rs:95-123): 63 on the full path, early exits on most tests
    "segment": 95,          # 1/dir 3, |d|^2 5, hit point 6, normal 6, front face 6, scatter ~60 (random unit vector 1.9 rounds), attenuation product 9
    "sample": 90,           # camera ray 46 (offsets, pixel position, lens disk 1.3 rounds), sky gradient 40, running sum 3
    "sample_sky_map": 70,   # spherical sky instead of the gradient: atan2 + asin + u,v (extra over "sample")
    "texel": 9,             # index math 6, /255 x3
}


def algorithmic_flops(st, triangles, sky_map):
    prim = ALGO_FLOPS["triangle_test"] if triangles else ALGO_FLOPS["sphere_test"]
    return (st["node_tests"] * ALGO_FLOPS["node_test"] + st["prim_tests"] * prim + st["segments"] * ALGO_FLOPS["segment"] +
            st["samples"] * (ALGO_FLOPS["sample"] + (ALGO_FLOPS["sample_sky_map"] if sky_map else 0)) +
            st["texel_fetches"] * ALGO_FLOPS["texel"])


def algorithmic_bytes(st, width, height, entry_bytes=32, prim_bytes=16):
    """DESIGN.md / SURVEY.md 8(d): B = S*(2*48) + N*entry + P*s_prim + T*4 + W*H*12."""
    return (st["segments"] * 96 + st["node_tests"] * entry_bytes + st["prim_tests"] * prim_bytes +
            st["texel_fetches"] * 4 + width * height * 12)


def hbm_probe():
    """What scripts/calib/hbm_peak.hip sustained on an MI355X of this pool (read / write / copy / triad over 2 GiB
    arrays), kept under profiles/: the measured figure SURVEY 8(d) asks for beside the nominal 8 TB/s."""
    path = os.path.join(ROOT, "profiles", "r02_hbm_peak.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    return {k: d[k] for k in ("read_GBps", "write_GBps", "copy_GBps", "triad_GBps", "measured_peak_GBps") if k in d} | {
        "source": "profiles/r02_hbm_peak.json (scripts/calib/hbm_peak.hip, a separate run -- not measured in this one)"}


def host_cores():
    """Cores this process may use: the affinity mask and the cgroup CPU quota, not the machine's logical CPU count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, (quota + period // 2) // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(scene, seed, target_s, faithful):
    """Time the f64 oracle on sample indices [0, k) of the same workload (all pixels), on the host cores this process
    may use.  A GPU box can report far more logical CPUs than its share (256 vs 16): the thread count is the one of
    {all, 32, 16} that renders a one-sample probe fastest, and it is stated in the line."""
    from crucible_amd import _abi as A
    from oracle.oracle import Oracle
    o = Oracle(A.CR_REAL_F64)
    h = o.scene_create(scene.flatten())
    cam = scene.scene_cam
    o.set_faithful(faithful)
    try:
        n_all = host_cores()
        best = None
        for cand in sorted({n_all, min(n_all, 32), min(n_all, 16)}, reverse=True):
            t0 = time.perf_counter()
            o.render(h, cam, seed=seed, sample_begin=0, sample_count=1, output_sum=True, n_threads=cand)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, cand)
        threads = best[1]
        k, dt = 1, 0.0
        for _ in range(4):   # grow the slice until it is about target_s of CPU work (bounded)
            t0 = time.perf_counter()
            o.render(h, cam, seed=seed, sample_begin=0, sample_count=k, output_sum=True, n_threads=threads)
            dt = time.perf_counter() - t0
            if dt >= 0.7 * target_s or k >= cam.samples:
                break
            k = max(k + 1, min(cam.samples, int(k * target_s / max(dt, 1e-3))))
    finally:
        o.set_faithful(False)
        o.scene_destroy(h)
    n = cam.image_width * cam.image_height * k
    how = ("timelines as 4x4 closure matrices at every hit, dead update_bb calls, material reference counts, per-pixel mutex, "
           "per-thread world copy (what Crucible does)") if faithful else "constants baked, direct evaluation (same results)"
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "host_logical_cpus": os.cpu_count(), "kind": "port",
            "variant": "faithful" if faithful else "sane",
            "sample": f"f64 oracle, {how}; same scene {cam.image_width}x{cam.image_height}, sample indices [0,{k}) of {cam.samples} "
                      f"for every pixel ({n / 1e6:.2f} Msamples, {dt:.1f} s)"}


def main():
    # stdout carries exactly one line, the result: everything else this process or its libraries print (RCCL's version banner at
    # communicator set-up, for one) goes to stderr.  fd 1 is pointed at stderr and the line is written to the saved descriptor.
    result_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--real", choices=["f32", "f64"], default="f64",
                    help="f64 = the reference's arithmetic (the headline); f32 = the opt-in fast mode (an extra line)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the short f32 timing carried as `f32_fast_mode`")
    ap.add_argument("--no-optin-line", action="store_true", help="skip the short CR_BVH_SAH_ORDERED timing carried as `opt_in_tree`")
    ap.add_argument("--workload", choices=["book1", "teapot", "million", "movie"], default="book1",
                    help="book1 = BASELINE configs[1] (the headline); teapot/million/movie = configs[2]/[3]/[4], extra lines")
    ap.add_argument("--bvh", choices=["reference", "sah", "ordered", "lbvh"], default="reference",
                    help="reference = the reference's median-split tree (parity mode, the headline); the others are the opt-in "
                         "trees of SURVEY 8(f) row 1 (extra lines, not the headline)")
    ap.add_argument("--sum-order", choices=["default", "reference", "relaxed"], default="default",
                    help="CrRenderParams.sum_order of the measured renders: default = the library's (relaxed sums); reference = the "
                         "reference's own order of products and sums, bit for bit (the parity mode)")
    ap.add_argument("--no-reference-line", action="store_true", help="skip the reference-order timing carried as `reference_order`")
    ap.add_argument("--end-to-end", type=int, default=0, metavar="FRAMES",
                    help="--workload movie: also render FRAMES consecutive frames (240 = all of configs[4]) with file output in P3, P6 "
                         "and PNG, a writer thread encoding frame k while frame k+1 renders; reported as `end_to_end`")
    ap.add_argument("--force-group", action="store_true",
                    help="diagnostic: go through cr_group_render even with one GPU (with CRUCIBLE_GROUP_FORCE_RCCL=1 the one-rank RCCL "
                         "communicator and its reduce are exercised too)")
    ap.add_argument("--reduce", choices=["library", "torch", "gloo"], default="library",
                    help="N>1: library = cr_group_render (RCCL reduce inside the C ABI); torch = torch.distributed (nccl) reduce of the "
                         "sums; gloo = reduce host copies (rehearsal of the multi-rank flow with ranks sharing one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from crucible_amd import _abi as A
    from crucible_amd.demo_builder import (book1_end_scene, load_teapot, million_spheres, procedural_sky,
                                           teapot_orbit_movie)
    from crucible_amd.distributed import shard_range
    from crucible_amd.group import RenderGroup
    from crucible_amd.renderer import CrucibleError, Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.reduce == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:   # torch's process group carries only the rendezvous, the barriers and the 128-byte RCCL id of the library's own communicator
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctl = dev if (world > 1 and args.reduce != "gloo") else torch.device("cpu")   # where control tensors of the process group live

    def real_of(name):
        return (A.CR_REAL_F32, torch.float32) if name == "f32" else (A.CR_REAL_F64, torch.float64)

    real_type, tdtype = real_of(args.real)
    sum_order = {"default": A.CR_SUM_DEFAULT, "reference": A.CR_SUM_REFERENCE_ORDER, "relaxed": A.CR_SUM_RELAXED}[args.sum_order]
    relaxed = args.sum_order == "relaxed" or (args.sum_order == "default" and os.environ.get("CRUCIBLE_SUM_ORDER") != "reference")
    seed, scene_seed = 0xC0FFEE, 1
    f32 = args.real == "f32"
    prim_bytes = 16 if f32 else 32
    frame_sharded, triangles, sky_map = False, False, False
    if args.workload == "book1":
        scene = book1_end_scene(1, scene_seed=scene_seed, image_width=args.width, samples=args.spp)
        wl = "book1 (RTIOW final scene, seeded) {W}x{H} @ {spp} spp, depth {d} -- BASELINE.json configs[1]"
    elif args.workload == "teapot":
        scene = load_teapot(1, image_width=args.width, samples=args.spp if args.spp != 512 else 1024, sky=procedural_sky())
        wl = "teapot.obj (6320 tris) + ground + procedural 2048x1024 env map {W}x{H} @ {spp} spp, depth {d} -- configs[2]"
        prim_bytes, triangles, sky_map = (36 if f32 else 72), True, True
    elif args.workload == "million":
        scene = million_spheres(1, scene_seed=scene_seed, image_width=args.width if args.width != 1920 else 3840,
                                samples=args.spp if args.spp != 512 else 256)
        wl = "1,000,001 procedural spheres {W}x{H} @ {spp} spp, depth {d} -- configs[3]"
    else:
        scene = teapot_orbit_movie(1, image_width=args.width, samples=args.spp)
        wl = "teapot orbit movie (240 frames at 24 fps), one frame per rank per step, {W}x{H} @ {spp} spp, depth {d} -- configs[4]"
        prim_bytes, triangles, sky_map, frame_sharded = (36 if f32 else 72), True, True, True
    scene.bvh_mode = {"sah": A.CR_BVH_SAH, "ordered": A.CR_BVH_SAH_ORDERED, "lbvh": A.CR_BVH_LBVH}.get(args.bvh, A.CR_BVH_REFERENCE)
    cam = scene.scene_cam
    W, H, spp = cam.image_width, cam.image_height, cam.samples
    flat = scene.flatten()
    s_begin, s_count = (0, spp) if frame_sharded else shard_range(rank, world, spp)
    spp_split = world > 1 and not frame_sharded

    # ---- the renderer: one handle, or (N > 1, spp split) a library group whose reduce is RCCL inside the C ABI
    group, group_error, reduce_how, torch_pg = None, None, None, None
    if (spp_split and args.reduce == "library") or (args.force_group and world == 1 and not frame_sharded):
        try:
            ident = None
            if world > 1 or os.environ.get("CRUCIBLE_GROUP_FORCE_RCCL"):
                ident = torch.zeros(A.CR_GROUP_ID_BYTES, dtype=torch.uint8, device=ctl)
                if rank == 0:
                    ident = torch.tensor(list(RenderGroup.unique_id()), dtype=torch.uint8, device=ctl)
                if world > 1:
                    dist.broadcast(ident, src=0)
                ident = bytes(ident.cpu().tolist())
            group = RenderGroup.rank(dev_index, rank, world, ident)
            group.upload_scene(flat)
        except (CrucibleError, OSError) as e:
            group, group_error = None, str(e)
        if world > 1:
            ok = torch.tensor([1 if group is not None else 0], dtype=torch.int32, device=ctl)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:          # any rank without RCCL inside the library: everybody falls back together
                if group is not None:
                    group.close()
                group = None
        elif group is None:
            raise SystemExit(f"--force-group: {group_error}")
    if spp_split and group is None and args.reduce == "gloo":
        reduce_how = "gloo reduce of host copies of the per-pixel sums (rehearsal)"
    elif spp_split and group is None:
        torch_pg = dist.group.WORLD
        reduce_how = "torch.distributed reduce (backend nccl = RCCL) of the per-pixel sums" + (
            f"; library group unavailable: {group_error}" if args.reduce == "library" else "")
    elif spp_split:
        reduce_how = "cr_group_render: ncclReduce(sum) of the per-pixel sums inside the library, divide on rank 0"
    r = None
    if group is None:
        r = Renderer(dev_index, sum_order=sum_order)
        r.upload_scene(flat)
    out = torch.empty((H, W, 3), dtype=tdtype, device=dev)
    step_no = [0]

    def step():
        """One render; returns the render kernels' time in ms (HIP events on the launch stream)."""
        if group is not None:
            st = group.render_device(cam, out.data_ptr(), seed=seed, real_type=real_type, sum_order=sum_order)
            return st["kernel_ms"]
        if frame_sharded:
            cam.frame = (step_no[0] * world + rank) % 240
            step_no[0] += 1
        r.render_device(cam, out.data_ptr(), seed=seed, real_type=real_type, sample_begin=s_begin, sample_count=s_count,
                        output_sum=spp_split)
        ms = r.last_kernel_ms()          # waits for the launch
        if spp_split and torch_pg is None:      # gloo rehearsal: host copies
            host = out.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                out.copy_(host.div_(float(spp)))
        elif spp_split:
            dist.reduce(out, dst=0, op=dist.ReduceOp.SUM, group=torch_pg)
            if rank == 0:
                out.div_(float(spp))
        return ms

    # one counted launch (untimed) for the work counters; also warms the build path
    if group is not None:
        st = group.render_device(cam, out.data_ptr(), seed=seed, real_type=real_type, sum_order=sum_order)
        st["bvh_entries"], st["scene_in_lds"] = None, None
    else:
        st = r.render_device(cam, out.data_ptr(), seed=seed, real_type=real_type, sample_begin=s_begin, sample_count=s_count,
                             output_sum=spp_split, want_stats=True)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the opt-in f32 mode, carried as an extra key (it is NOT the reference's arithmetic)
    f32_line = None
    if rank == 0 and world == 1 and args.real == "f64" and not args.no_f32_line and r is not None:
        o32 = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        ms32 = []
        for _ in range(3):
            r.render_device(cam, o32.data_ptr(), seed=seed, real_type=A.CR_REAL_F32)
            ms32.append(r.last_kernel_ms())
        best = min(ms32[1:])
        f32_line = {"value": round(W * H * spp / (best * 1e-3) / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(best, 3), "dtype": "f32",
                    "note": "CR_REAL_F32, kernel time of the best of 2 renders after 1 warm-up; narrower than the reference: bit-equal to the f32 "
                            "restatement only, ~29 % of pixels differ from the f64 image by more than 1e-4 (DESIGN.md section 2)"}
        del o32

    # the reference's own order of products and sums (the bit-exact parity mode) on the same frame, and how far the measured
    # frame is from it: largest per-channel difference, on the device
    reference_line = None
    if rank == 0 and world == 1 and relaxed and not args.no_reference_line and r is not None and not frame_sharded:
        fast = torch.empty_like(out)
        r.render_device(cam, fast.data_ptr(), seed=seed, real_type=real_type)
        r.last_kernel_ms()
        oref = torch.empty_like(out)
        msr = []
        for _ in range(3):
            r.render_device(cam, oref.data_ptr(), seed=seed, real_type=real_type, sum_order=A.CR_SUM_REFERENCE_ORDER)
            msr.append(r.last_kernel_ms())
        str_ = r.render_device(cam, oref.data_ptr(), seed=seed, real_type=real_type, sum_order=A.CR_SUM_REFERENCE_ORDER, want_stats=True)
        best = min(msr[1:])
        reference_line = {"sum_order": "CR_SUM_REFERENCE_ORDER (innermost-first products, samples added in draw order: bit-equal to the oracle)",
                          "value": round(W * H * spp / (best * 1e-3) / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(best, 3), "dtype": args.real,
                          "max_abs_difference_of_the_measured_frame": float((fast - oref).abs().max().item()),
                          "counters_equal": all(str_[k] == st[k] for k in ("segments", "node_tests", "prim_tests", "texel_fetches")),
                          "note": "kernel time of the best of 2 renders after 1 warm-up (includes the ordered-sum kernel); needs a per-sample "
                                  "colour buffer of W*H*spp*3 reals and a per-path attenuation stack"}
        del fast, oref

    # the opt-in tree of SURVEY 8(f) row 1 on the same frame, carried as an extra key: the same image bit for bit
    # (checked here, on the device), fewer box tests.  The headline stays on the reference's own topology.
    optin_line = None
    if rank == 0 and world == 1 and args.bvh == "reference" and not args.no_optin_line and r is not None and not frame_sharded:
        ref_img = torch.empty_like(out)
        r.render_device(cam, ref_img.data_ptr(), seed=seed, real_type=real_type)
        r.last_kernel_ms()
        flat_o = scene.flatten()
        flat_o.desc.bvh_mode = A.CR_BVH_SAH_ORDERED
        r2 = Renderer(dev_index)
        try:
            r2.upload_scene(flat_o)
            o2 = torch.empty_like(out)
            ms2 = []
            for _ in range(3):
                r2.render_device(cam, o2.data_ptr(), seed=seed, real_type=real_type)
                ms2.append(r2.last_kernel_ms())
            st2 = r2.render_device(cam, o2.data_ptr(), seed=seed, real_type=real_type, want_stats=True)
            best = min(ms2[1:])
            optin_line = {"bvh": "CR_BVH_SAH_ORDERED (binned SAH, near child first)", "value": round(W * H * spp / (best * 1e-3) / 1e6, 2),
                          "unit": "Msamples/s", "kernel_ms": round(best, 3), "dtype": args.real,
                          "image_identical_to_reference_topology": bool(torch.equal(o2, ref_img)),
                          "node_tests_per_segment": round(st2["node_tests"] / max(1, st2["segments"]), 2),
                          "note": "kernel time of the best of 2 renders after 1 warm-up; the frame is compared on the device with the "
                                  "reference-topology frame of the same seed"}
            del o2
        finally:
            r2.close()
        del ref_img

    # configs[4] end to end: FRAMES consecutive frames of the movie through the boundary WITH their files, the way
    # Scene::render_movie produces them (scene/mod.rs:295-322: render a frame, write it, next frame) -- except that a writer
    # thread encodes and writes frame k while frame k+1 renders.  Render-only first, then one pass per format.
    end_to_end = None
    if rank == 0 and world == 1 and frame_sharded and args.end_to_end > 0 and r is not None:
        import shutil
        import tempfile
        import threading
        import numpy as np
        nfr = args.end_to_end
        host = [np.empty((H, W, 3), dtype=np.float64 if args.real == "f64" else np.float32) for _ in range(2)]
        end_to_end = {"frames": nfr, "samples_per_frame": W * H * spp, "formats": {}}
        t0 = time.perf_counter()
        k_sum = 0.0
        for fr in range(nfr):
            cam.frame = fr % 240
            _, stf = r.render(cam, seed=seed, real_type=real_type)
            k_sum += stf["kernel_ms"]
        t_render = time.perf_counter() - t0
        end_to_end["render_only"] = {"wall_s": round(t_render, 3), "kernel_s": round(k_sum * 1e-3, 3),
                                     "msamples_per_s": round(nfr * W * H * spp / t_render / 1e6, 1),
                                     "what": "cr_render_host per frame (render + device->host copy + the Color::new check), no file"}
        for fmt, ext in (("P3 (the reference's ASCII PPM)", ".ppm"), ("P6 (binary PPM)", ".p6.ppm"), ("PNG", ".png")):
            tmp = tempfile.mkdtemp(prefix="crucible_e2e_")
            writer, nbytes = None, 0
            t0 = time.perf_counter()
            for fr in range(nfr):
                cam.frame = fr % 240
                img, _ = r.render(cam, seed=seed, real_type=real_type)
                if writer is not None:
                    writer.join()
                host[fr & 1][...] = img
                path = os.path.join(tmp, f"image{fr:03d}{ext}")
                writer = threading.Thread(target=r.write_image, args=(path, host[fr & 1]))
                writer.start()
            writer.join()
            wall = time.perf_counter() - t0
            nbytes = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
            shutil.rmtree(tmp, ignore_errors=True)
            end_to_end["formats"][fmt] = {"wall_s": round(wall, 3), "msamples_per_s": round(nfr * W * H * spp / wall / 1e6, 1),
                                          "vs_render_only": round(wall / t_render, 4), "file_MB_per_frame": round(nbytes / nfr / 1e6, 2)}
        cam.frame = 0

    if rank == 0:
        total_samples = W * H * spp * args.steps * (world if frame_sharded else 1)
        value = total_samples / elapsed / 1e6
        entry_bytes = 32 if f32 else 64
        k_ms = kernel_ms / max(args.steps, 1)
        flops = algorithmic_flops(st, triangles, sky_map)          # of this rank's launch
        B = algorithmic_bytes(st, W, H, entry_bytes, prim_bytes)
        tflops = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        peak = VALU_PEAK_TFLOPS[args.real]
        key = f"{args.workload}_{W}x{H}_spp{s_count}_{args.real}" + ("" if args.bvh == "reference" else "_" + args.bvh) + ("" if relaxed else "_reference_order")
        pmc, traffic = None, None
        ppath = os.path.join(ROOT, "profiles", "r03_pmc.json")
        if os.path.exists(ppath):
            try:
                pmc = json.load(open(ppath)).get(key)
            except Exception:
                pmc = None
        if pmc:
            traffic = pmc["hbm_bytes"]
            pmc = dict(pmc, from_committed_profile=f"profiles/r03_pmc.json[{key}] -- collected in separate rocprofv3 --pmc runs of this "
                                                   "workload, NOT measured in this run")
        # which kernel ran, and what the compiler allocated for it (rocprofv3's arch_vgpr_count halves the allocation on gfx950)
        b = lambda v: "true" if v else "false"
        res_code = st["scene_in_lds"] if st["scene_in_lds"] is not None else 1
        anim = flat.desc.n_keys > 0
        cam_keyed = bool(cam.look_from_tl.keyframes() or cam.look_at_tl.keyframes())
        # f64: f32 screening records; f32, unordered trees: the wrappers in the same link layout
        screened = not (f32 and args.bvh == "ordered") and os.environ.get("CRUCIBLE_SCREEN", "1") != "0" and os.environ.get("CRUCIBLE_PIPELINE", "mega") == "mega" \
            and not (res_code == 1 and os.environ.get("CRUCIBLE_SCREEN_LDS", "1") == "0")
        kname = (f"cr::pathtrace_kernel<{'float' if f32 else 'double'}, {res_code}, {b(anim)}, {b(args.bvh == 'ordered')}, "
                 f"{b(cam_keyed and not anim)}, {b(relaxed)}, {b(screened)}>")
        resources = None
        try:
            resources = json.load(open(os.path.join(ROOT, "profiles", "r03_kernel_resources.json")))["kernels"].get(kname)
        except Exception:
            resources = None
        sustained = VALU_SUSTAINED_TOPS[args.real]
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
                       "sum_order": ("CR_SUM_RELAXED (library default): path-order products, fixed-point per-pixel sums; within 1e-12 of the reference "
                                     "order, equal work counters" if relaxed else
                                     "CR_SUM_REFERENCE_ORDER: the reference's own order of products and sums, bit-equal to the oracle"),
                       "box_tests": "f64 walk on f32 screening records, f64 test where f32 cannot decide (exact)" if screened else "in the scalar type",
                       "parallelism": ("1 GPU" + (" through cr_group_render" if group is not None else "")) if world == 1 else (f"frame-shard x{world}, no collective" if frame_sharded else
                                                                     f"spp-shard x{world}: {reduce_how}")},
            "roofline": {"bound": "valu", "achieved": round(tflops, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                         "traffic": traffic,
                         "kernel": kname + (" (+ cr::fx_finalize_kernel, ~0.03 ms: fixed-point sums -> frame)" if relaxed else
                                            " (+ cr::sg_finalize_kernel, the ordered sum)"),
                         "kernel_resources": resources and dict(resources, source="profiles/r03_kernel_resources.json (compiler remarks of the "
                                                                "same sources; not rocprofv3's halved arch_vgpr_count)"),
                         "kernel_ms": round(k_ms, 4),
                         "algorithmic_flops_per_launch": int(flops), "flops_per_unit": ALGO_FLOPS,
                         "peak_without_fma": peak / 2, "frac_of_peak_without_fma": round(tflops / (peak / 2), 4),
                         "peak_sustained": round(sustained, 2), "frac_of_sustained": round(tflops / sustained, 4),
                         "peak_sustained_note": "operations/s of the scalar type the chip sustains WITHOUT fusing, from scripts/calib/valu_peak.hip "
                                                "(profiles/r03_valu_calib.txt): one wave64 f64 add/mul/max per ~2.1 ns per SIMD at any occupancy "
                                                "(the clock drops to 1.0-1.4 GHz as more waves issue), f32 add ~1.2 ns",
                         "why_valu": WHY_VALU[args.workload],
                         "counters_per_launch": {k: st[k] for k in ("samples", "segments", "node_tests", "prim_tests", "texel_fetches")},
                         "hbm": {"algorithmic_bytes_per_launch": int(B), "achieved_algorithmic_GBps": round(B / (k_ms * 1e-3) / 1e9, 2) if k_ms > 0 else None,
                                 "peak_GBps": HBM_PEAK_GBS, "note": "SURVEY 8(d) byte model; exceeds the HBM peak wherever the scene is served from LDS / L2 -- "
                                                                   "not a bound of this kernel",
                                 "measured_GBps_from_profile": round(traffic / (k_ms * 1e-3) / 1e9, 1) if traffic and k_ms > 0 else None,
                                 "probed_peak": hbm_probe()},
                         "executed": pmc},
            "kernel_msamples_per_s": round(W * H * s_count / (k_ms * 1e-3) / 1e6, 2) if k_ms > 0 else None,
        }
        if end_to_end:
            rec["end_to_end"] = end_to_end
        if reference_line:
            rec["reference_order"] = reference_line
        if f32_line:
            rec["f32_fast_mode"] = f32_line
        if optin_line:
            rec["opt_in_tree"] = optin_line
        if world == 1 and not args.no_cpu_baseline and args.workload != "million":
            rec["cpu_baseline"] = cpu_baseline(scene, seed, 14.0, False)
            rec["cpu_baseline_faithful"] = cpu_baseline(scene, seed, 10.0, True)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(rec) + "\n").encode())

    if r is not None:
        r.close()
    if group is not None:
        group.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
