#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: cornellBox PathTracing at 1920x1080 (BASELINE configs[2]).

  python bench.py --gpus N --steps K --warmup W                        one process; N > 1: one context over N GPUs
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W     one process per GPU

One step = one wavefront pass of SPP_PER_STEP samples per pixel over the whole 1080p frame on every GPU
(raygen, then extend/shade/shadow/probe for each bounce up to depth 15, then the film accumulation). GPU g
renders sample indices [g*K*P, (g+1)*K*P) of a (N*K*P)-spp job; after the K steps the float film (one packed buffer of
4 W H floats) is sum-reduced to GPU 0 over RCCL inside the timed region. The scene is resident in HBM and camera rays
are generated on the device, so no host data moves in the timed region.

Two hosts for N > 1 (the same split, the same single reduce):
  group   plain `python bench.py --gpus N` (no WORLD_SIZE in the environment): ONE process, as the reference's main is one
          process that calls render() once (reference src/RayTracing.cpp:32-33); the library's multi-GPU context
          (rt_create_multi) shards the job and runs its own ncclReduce. On a machine with fewer than N GPUs device 0 is
          listed N times, which is the library's one-GPU rehearsal of that path ("rehearsal": true in the line).
  ranks   under torch.distributed.run: one process and one rt_create context per GPU, reduce through torch.distributed.

Prints ONE JSON line (rank 0). value = whole-job Mrays/s, one ray = one KD-tree traversal of either flavour
(extend + MIS-probe + shadow), counted by the kernels themselves. Besides the contract's fields the line carries
  roofline        k_extend, the dominant traversal kernel, against the roof that binds it on this 8 KB scene (VALU issue),
                  with its HBM side and its lane occupancy from the hardware counters
  roofline_shade  k_shade against HBM
  strong_1024spp  BASELINE configs[2] as worded: ONE 1024-spp job split over the N ranks, reduce included (after the
                  weak-scaling region the driver times)
  secondary       BASELINE configs[3], the 1 M-triangle soup whose tables live in HBM (1-GPU runs only): the
                  incoherent-ray batch and a 1080p x 16 spp path trace, with the HBM roofline of the streaming traversal kernel
  cpu_baseline    the CPU oracle on the host cores, on bounded samples of the same workloads
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WIDTH, HEIGHT = 1920, 1080
SPP_PER_STEP = int(os.environ.get("RT_BENCH_SPP_PER_STEP", "64"))
MAX_DEPTH = 15
STRONG_SPP = 1024
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak
FP32_VALU_PEAK_TFLOPS = 157.3
# Figures that only the hardware counters give (rocprofv3 --pmc runs of this same command, summaries under profiles/);
# counters cannot be read inside a timed run, so the per-ray figures measured under the profiler are scaled by this run's
# own ray counts and HIP-event times.
PMC = {
    "source": "profiles/r03_pmc_bench_steps4_noslp.md",
    # HBM-side bytes per extend ray, launches weighted by their rays (FETCH_SIZE doubled: gfx950 counts 16-B-per-lane coalesced
    # reads at half their bytes): camera rays (lockstep kernel, no origin records) 16 B read + 16 B written; bounce rays
    # (refilling kernel) 31 B read + 30 B written - a lane stores its hit record when its ray ends, so those 16-byte stores
    # are not coalesced
    "extend_hbm_bytes_per_ray": 50.0,
    "extend_lanes_active": 18.8,            # refilling kernel: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, of 64 (camera rays in lockstep: 42.1)
    "extend_lds_conflict_cycles": 1.51,     # refilling kernel: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS (lanes stand in different leaves)
    "extend_valu_per_ray": 50.9,            # refilling kernel on the pipeline's own bounce rays: SQ_INSTS_VALU per ray (lockstep, round 2: 61.5 over all bounces)
    "shade_hbm_bytes_per_path": 109.0,      # k_shade: (FETCH_SIZE x 2 + WRITE_SIZE) per ray entering the kernel
    "shade_lanes_active": 54.4,
    "valu_per_camera_sample": 167.5,        # SQ_INSTS_VALU of every kernel of the run / camera samples (599.3 M: nine 32-spp passes and one of 1 spp)
}
# Cycles a SIMD needs to issue one wave64 vector instruction with two or more waves resident (tools/valu_rate.hip,
# profiles/r03_valu_issue_rates.txt): 2.2 for plain two-operand f32 / integer instructions on VGPRs, 4.1 for fma, min / max,
# compares, conversions, packed and f64 instructions and anything with an SGPR operand. 40 % of the instructions of the three
# big kernels are of the first kind (static ISA: k_extend_refill 0.37, k_shade 0.47, k_shadow 0.29, weighted by their shares).
ISSUE_CYCLES_MIX = 0.40 * 2.2 + 0.60 * 4.1
ISSUE_CYCLES_SLOW = 4.1
SIMDS, CLOCK_HZ = 1024, 2.4e9
PMC_SOUP = {
    "source": "profiles/r03_soup_refill_pmc_incoherent.md",
    "fetch_bytes_per_ray_raw": 4232.0,      # FETCH_SIZE per ray (64 B per request, no correction: 8-byte gathers)
    "l2_hit_rate": 0.53,
    "lanes_active": 17.4,
}


def bytes_per_ray(ni, nl, nidx, ntri, closest=True):
    """Algorithmic traffic of one traversal: ray in (28 B), hit out (16 B; 1 B for any-hit), 8 B per node visited,
    4 B per leaf index, 36 B per triangle tested."""
    return 28 + (16 if closest else 1) + 8 * (ni + nl) + 4 * nidx + 36 * ntri


def flops_per_ray(ni, ntri, ncand):
    return 40 + 6 * ni + 44 * ntri + 35 * ncand


# k_shade per ray entering it: the hit record (16 B) is read for every ray; a ray that hit (0.53 of the camera rays, 0.77 to
# 0.86 of later rays: 0.70 over a pass) reads its direction record, throughput and generator state (48 B) and writes the
# next ray (32 B), throughput and generator (32 B) and a shadow entry (48 B). Scene tables (80 B per primitive) stay in L2.
SHADE_BYTES_PER_PATH = 16 + 0.70 * (48 + 32 + 32 + 48)


def host_cores():
    """Threads the CPU baseline uses: the job's share of the machine (OMP_NUM_THREADS on the GPU box), at most 16."""
    return int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)


def cpu_baseline(scene, cfg, target_seconds=12.0):
    """Times the CPU oracle (a port of the reference's path; the reference's sources cannot travel to this
    machine) on the host cores given to this job, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    cores = host_cores()
    orc = Oracle(scene, cfg)
    _, _, st = orc.render(integrator=0, rng_mode=0, seed=1, spp=1, threads=cores)     # calibration pass
    rate = st["camera_samples"] / max(st["seconds"], 1e-9)
    spp = int(max(1, min(64, round(target_seconds * rate / (cfg.width * cfg.height)))))
    _, _, st = orc.render(integrator=0, rng_mode=0, seed=1, spp=spp, threads=cores)
    rays = st["rays_closest"] + st["rays_any"]
    return {
        "value": rays / st["seconds"] / 1e6,
        "unit": "Mrays/s",
        "samples_per_sec": st["camera_samples"] / st["seconds"],
        "seconds": st["seconds"],
        "cores": cores,
        "host_cpu_count": os.cpu_count(),
        "kind": "port",
        "sample": f"cornellBox PathTracing {cfg.width}x{cfg.height}, depth {MAX_DEPTH}, {spp} spp (of the GPU job's), {cores} threads",
    }


def soup_secondary(torch, dev, base_cfg, with_cpu):
    """BASELINE configs[3]: 1 M-triangle soup (tables in HBM). Incoherent closest-hit batch with the library's default
    schedule for such scenes, timed with HIP events on the library's own stream, and a 1080p x 16 spp path trace."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raygen
    from raytracing_amd.device import Context
    n_tris, n_rays, reps = 1_000_000, 1 << 22, 5
    t0 = time.perf_counter()
    scene, cfg = raygen.soup_scene(n_tris, base_cfg)
    build_s = time.perf_counter() - t0
    ctx = Context(dev.index)
    ctx.upload_scene(scene)
    o_np, d_np = raygen.incoherent_rays(scene.bounds, n_rays, 3)
    # per-ray work from the instrumented kernel on a sample of the batch
    step = n_rays // 200_000
    ctx.reset_stats()
    ctx.trace_closest(np.ascontiguousarray(o_np[::step]), np.ascontiguousarray(d_np[::step]))
    st = ctx.stats()
    cr = max(1, st["counted_rays"])
    ni, nl, nidx, ntri = st["n_interior"] / cr, st["n_leaf"] / cr, st["n_leaf_idx"] / cr, st["n_tri_tests"] / cr
    bpr = bytes_per_ray(ni, nl, nidx, ntri)
    o, d = torch.tensor(o_np, device=dev), torch.tensor(d_np, device=dev)
    prim = torch.empty(n_rays, dtype=torch.int32, device=dev)
    t = torch.empty(n_rays, dtype=torch.float32, device=dev)
    bary = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)   # the stream the kernel is launched on

    def launch():
        ctx.trace_closest_dev(n_rays, o.data_ptr(), d.data_ptr(), 0, prim.data_ptr(), t.data_ptr(), bary.data_ptr())

    launch()
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        launch()
    e1.record(stream)
    ctx.synchronize()
    ms = e0.elapsed_time(e1) / reps
    achieved = bpr * n_rays / (ms * 1e-3) / 1e9
    out = {
        "workload": f"{n_tris}-triangle soup (BASELINE configs[3]): {scene.n_nodes} KD nodes, {len(scene.leaf_idx)} leaf indices, "
                    f"{(scene.n_nodes * 8 + len(scene.leaf_idx) * 4 + scene.n_prims * 48) / 1e6:.0f} MB of traversal tables in HBM",
        "host_build_seconds": build_s,
        "incoherent": {
            "n_rays": n_rays,
            "Mrays_per_s": n_rays / (ms * 1e-3) / 1e6,
            "hit_fraction": float((prim >= 0).float().mean()),
            "roofline": {
                "kernel": "k_trace_closest_refill (the traversal of k_extend_refill on a caller's batch)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": PMC_SOUP["fetch_bytes_per_ray_raw"] * n_rays,
                "traffic_unit": "bytes per launch fetched by L2 from the fabric (PMC FETCH_SIZE, uncorrected; includes Infinity Cache hits)",
                "bytes_per_ray": bpr,
                "avg_launch_ms": ms,
                "launches": reps,
                "per_ray": {"interior": ni, "leaf": nl, "leaf_idx": nidx, "tri_tests": ntri},
                "l2_hit_rate_pmc": PMC_SOUP["l2_hit_rate"],
                "lanes_active_pmc": PMC_SOUP["lanes_active"],
                "pmc_source": PMC_SOUP["source"],
            },
        },
    }
    # path tracing on the same scene with the library's own pass sizes: three passes each, overlapping on three pipelines
    ctx.set_camera(cfg)
    ctx.set_kernel_times(True)   # these two jobs report launch times; their seconds include the (small) cost of the events
    rgb = torch.zeros((cfg.height, cfg.width, 3), dtype=torch.float32, device=dev)
    w = torch.zeros((cfg.height, cfg.width), dtype=torch.float32, device=dev)
    for spp in (16, 128):
        kw = dict(integrator=0, rng_mode=0, seed=1, spp=spp, max_depth=MAX_DEPTH)
        ctx.render_dev(rgb.data_ptr(), w.data_ptr(), **kw)
        ctx.synchronize()
        rgb.zero_()
        w.zero_()
        ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.render_dev(rgb.data_ptr(), w.data_ptr(), **kw)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        rays = st["rays_closest"] + st["rays_any"]
        assert float(w.min()) >= spp
        out[f"path_1080p_{spp}spp"] = {"seconds": dt, "Mrays_per_s": rays / dt / 1e6, "samples_per_sec": st["camera_samples"] / dt,
                                       "rays_per_sample": rays / max(1, st["camera_samples"]),
                                       "extend_avg_launch_ms": st["extend_ms"] / max(1, st["extend_launches"])}
    ctx.close()
    if with_cpu:
        from concurrent.futures import ThreadPoolExecutor
        from oracle_lib import Oracle
        cores = host_cores()
        orc = Oracle(scene, cfg)
        n_cpu = n_rays   # ~1.3 s on 16 threads, ~20 core-seconds
        chunks = np.array_split(np.arange(n_cpu), cores)

        def work(idx):
            orc.trace_closest(o_np[idx], d_np[idx])

        with ThreadPoolExecutor(cores) as pool:
            list(pool.map(work, [c[:1000] for c in chunks]))   # touch the tables once
            t0 = time.perf_counter()
            list(pool.map(work, chunks))
            cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_cpu / cpu_s / 1e6, "unit": "Mrays/s", "seconds": cpu_s, "cores": cores,
                               "host_cpu_count": os.cpu_count(), "kind": "port",
                               "sample": f"the same {n_cpu} incoherent rays, closest hit, {cores} threads"}
        out["incoherent"]["gpu_over_cpu"] = out["incoherent"]["Mrays_per_s"] / out["cpu_baseline"]["value"]
    return out


def plan(gpus, environ):
    """How this invocation runs: 'single' (one GPU), 'group' (one process, one context over `gpus` GPUs) or 'ranks' (one
    process per GPU, started by torch.distributed.run). Never refuses: `--gpus N` without a launcher is the group form."""
    world = int(environ.get("WORLD_SIZE", "1"))
    if world > 1:
        return {"mode": "ranks", "world": world, "rank": int(environ.get("RANK", "0")), "local_rank": int(environ.get("LOCAL_RANK", "0"))}
    if gpus > 1:
        return {"mode": "group", "world": gpus, "rank": 0, "local_rank": 0}
    return {"mode": "single", "world": 1, "rank": 0, "local_rank": 0}


def c1_job(torch, dev, scene):
    """BASELINE configs[0] on the GPU: the reference's own CPU-runnable case (cornellBox PathTracing 800x800, 512 spp, depth
    15), the one job for which a time of the reference itself exists (202.6 s on 8 vCPU, measured by the survey)."""
    from raytracing_amd.device import Context
    cfg = scene.config.with_resolution(800, 800, spp=512)
    ctx = Context(dev.index)
    ctx.upload_scene(scene)
    ctx.set_camera(cfg)
    rgb = torch.zeros((800, 800, 3), dtype=torch.float32, device=dev)
    w = torch.zeros((800, 800), dtype=torch.float32, device=dev)
    kw = dict(integrator=0, rng_mode=0, seed=1, spp=512, max_depth=MAX_DEPTH)
    ctx.render_dev(rgb.data_ptr(), w.data_ptr(), **kw)   # warm-up: allocates the queues
    ctx.synchronize()
    rgb.zero_()
    w.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.render_dev(rgb.data_ptr(), w.data_ptr(), **kw)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    assert float(w.min()) >= 512
    ctx.close()
    return {"job": "cornellBox PathTracing 800x800 x 512 spp, depth 15 (BASELINE configs[0]), counter RNG, film in device memory",
            "gpu_seconds": dt, "reference_cpu_seconds": 202.6, "reference_cpu_cores": 8,
            "ratio": 202.6 / dt,
            "note": "the reference itself, compiled and timed by the survey on 8 vCPU; the only like-for-like GPU / reference-CPU figure"}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 1 M-triangle soup (BASELINE configs[3])")
    ap.add_argument("--no-strong", action="store_true", help="skip the 1024-spp strong-scaling job")
    ap.add_argument("--no-c1", action="store_true", help="skip the 800x800 x 512 spp job (BASELINE configs[0]) that is timed against the reference")
    args = ap.parse_args(argv)
    pl = plan(args.gpus, os.environ)   # decided before anything touches a GPU
    mode, world, rank, local_rank = pl["mode"], pl["world"], pl["rank"], pl["local_rank"]

    import torch
    import torch.distributed as dist
    from raytracing_amd.device import Context
    from raytracing_amd.distributed import shard_range
    from raytracing_amd.scene import FlatScene

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the device path has no CPU fallback")
    n_visible = torch.cuda.device_count()
    local_dev = local_rank % n_visible   # one rank per GPU; wraps only in a rehearsal with fewer GPUs than ranks
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # ranks: RCCL is the transport; "gloo" (RT_BENCH_BACKEND) only rehearses the multi-rank logic on a box with fewer GPUs than
    # ranks, where RCCL refuses two ranks on one device - it stages the film through host memory and is not a measurement
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if mode == "ranks":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def reduce_to_rank0(tensor):
        if backend == "nccl":
            dist.reduce(tensor, dst=0, op=dist.ReduceOp.SUM)
        else:
            staged = tensor.cpu()
            dist.reduce(staged, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                tensor.copy_(staged)

    scene = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz"))
    cfg = scene.config.with_resolution(WIDTH, HEIGHT, spp=SPP_PER_STEP)
    rehearsal = False
    if mode == "group":
        # one context over `world` GPUs; with fewer GPUs than that, the library's one-GPU rehearsal of the same code path
        rehearsal = n_visible < world
        ctx = Context(device_ids=[0] * world if rehearsal else list(range(world)))
    else:
        ctx = Context(local_dev)
    ctx.upload_scene(scene)
    ctx.set_camera(cfg)
    PIPELINES = int(os.environ.get("RT_BENCH_PIPELINES", "3"))   # for comparisons; 3 is the library's default
    ctx.set_pipelines(PIPELINES)

    K, W, P = args.steps, args.warmup, SPP_PER_STEP
    spp_total = world * K * P
    # the film as ONE buffer - rgb sums (3 W H) then weight sums (W H) - so that the exchange step is one reduce
    npix = WIDTH * HEIGHT
    film = torch.zeros(4 * npix, dtype=torch.float32, device=dev)
    rgb = film[:3 * npix].view(HEIGHT, WIDTH, 3)
    wsum = film[3 * npix:].view(HEIGHT, WIDTH)

    def run(spp_begin, spp_end, total, collect=False, per_pass=P):
        ctx.render_dev(rgb.data_ptr(), wsum.data_ptr(), integrator=0, rng_mode=0, seed=1, spp=total, max_depth=MAX_DEPTH,
                       spp_begin=spp_begin, spp_end=spp_end, spp_per_pass=per_pass, collect_stats=collect)

    def library_passes(spp_job):
        """How rt_render cuts a job when spp_per_pass is 0: equal passes of at most 2^27 paths, a multiple of three of them."""
        n = -(-spp_job // max(1, (1 << 27) // npix))
        if n > 1 or (spp_job >= 3 and spp_job * npix >= 1 << 23):
            n = -(-n // 3) * 3
        return n

    def job(begin_of_rank0, per_gpu, total):
        """`per_gpu` samples per pixel on every GPU of a `total`-spp job, film summed on GPU 0: in ranks mode this rank's
        shard and the reduce; in group mode ONE call, the library shards it and runs its own reduce."""
        if mode == "ranks":
            run(begin_of_rank0 + rank * per_gpu, begin_of_rank0 + (rank + 1) * per_gpu, total)
            ctx.synchronize()   # the render runs on the library's own streams; RCCL runs on torch's
            reduce_to_rank0(film)
        else:
            run(begin_of_rank0, begin_of_rank0 + world * per_gpu, total)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if mode == "ranks":
            dist.barrier()
            torch.cuda.synchronize()

    # per-ray traversal work (node / leaf / triangle counts), measured once outside the timed region
    ctx.reset_stats()
    run(0, 1, max(1, W * P) + 1, collect=True)
    ctx.synchronize()
    st = ctx.stats()
    cr = max(1, st["counted_rays"])
    ni, nl, nidx, ntri = st["n_interior"] / cr, st["n_leaf"] / cr, st["n_leaf_idx"] / cr, st["n_tri_tests"] / cr
    # warmup steps (untimed); with more than one GPU they include the first use of the collective (communicator set-up,
    # channel buffers)
    if W > 0 or world > 1:
        job(0, max(1, W) * P, world * max(1, W) * P)
    fence()
    film.zero_()
    ctx.reset_stats()
    fence()

    # timed region: no HIP events in the render's streams (RT_OPTION_KERNEL_TIMES is off)
    t0 = time.perf_counter()
    job(0, K * P, spp_total)
    fence()
    elapsed = time.perf_counter() - t0

    st = ctx.stats()
    wmin = float(wsum.min()) if rank == 0 else 0.0

    # BASELINE configs[2] as worded: one 1024-spp job, spp-sharded over the GPUs, reduce included (strong scaling)
    strong = None
    if not args.no_strong and world <= STRONG_SPP:
        film.zero_()
        ctx.reset_stats()
        fence()
        ts = time.perf_counter()
        if mode == "ranks":
            sb, se = shard_range(STRONG_SPP, rank, world)
            run(sb, se, STRONG_SPP, per_pass=0)   # the library's own pass sizes
            ctx.synchronize()
            reduce_to_rank0(film)
        else:
            sb, se = shard_range(STRONG_SPP, 0, world)   # the library cuts the job by the same rule
            run(0, STRONG_SPP, STRONG_SPP, per_pass=0)
        fence()
        strong_elapsed = time.perf_counter() - ts
        sst = ctx.stats()
        strong = (strong_elapsed, float(sst["rays_closest"] + sst["rays_any"]), float(sst["camera_samples"]),
                  float(wsum.min()) if rank == 0 else 0.0, library_passes(se - sb))

    # Kernel launch times, outside the timed region: the same steps once more with HIP events around the extend and shade
    # launches (three passes share the CUs, as in the timed region), then one step with the passes run one after the other
    # ("exclusive": a kernel alone on the device).
    timed_launches = exclusive = None
    if rank == 0:
        members = world if mode == "group" else 1   # a group cuts the job by sample index: the same passes per GPU as above
        ctx.set_kernel_times(True)
        ctx.reset_stats()
        run(0, min(K, 8) * P * members, min(K, 8) * P * members)
        ctx.synchronize()
        timed_launches = ctx.stats()
        ctx.set_pipelines(1)
        try:
            ctx.reset_stats()
            run(0, P * members, P * members)
            ctx.synchronize()
            sx = ctx.stats()
            if sx["extend_ms"] > 0:
                exclusive = sx
        finally:
            ctx.set_pipelines(PIPELINES)
            ctx.set_kernel_times(False)
    vals = [elapsed, float(st["rays_closest"] + st["rays_any"]), float(st["camera_samples"])]
    vals += list(strong[:3]) if strong else [0.0, 0.0, 0.0]
    local = torch.tensor(vals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    tmax = local.clone()
    tsum = local.clone()
    if mode == "ranks":
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    t = float(tmax[0])
    rays, samples = float(tsum[1]), float(tsum[2])

    if rank == 0:
        # sanity: every pixel received exactly spp_total samples (plus rare border-rounded neighbours)
        assert wmin >= spp_total, f"film incomplete: min weight {wmin} < {spp_total}"
        if mode == "group":
            how = ("rt_create_multi, one ncclReduce" if not rehearsal else
                   "rt_create_multi on one GPU listed %d times (rehearsal: device-local sums stand in for the ncclReduce)" % world)
        else:
            how = f"one process per GPU, one {'RCCL' if backend == 'nccl' else backend + ' (rehearsal)'} sum-reduce through torch.distributed"
        out = {
            "metric": "Mrays/s (traversal+isect) and samples/sec, cornellBox 1080p, 1/2/4/8 GPUs",
            "value": rays / t / 1e6,
            "unit": "Mrays/s",
            "samples_per_sec": samples / t,
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": t * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "cornellBox scene of the reference (38 triangles, 133 KD nodes) from a committed fixture; camera rays generated on the device",
            "config": {
                "workload": f"cornellBox PathTracing {WIDTH}x{HEIGHT}, depth {MAX_DEPTH}, counter RNG (BASELINE configs[2]); a step is one "
                            f"pass of {P} spp: this run renders {K * P} spp per GPU, {spp_total} spp in total (the configuration's own "
                            f"1024-spp job is timed separately: strong_1024spp)",
                "spp_per_gpu": K * P,
                "spp_total": spp_total,
                "paths_in_flight_per_gpu": WIDTH * HEIGHT * P,
                "parallelism": f"spp-sharded x{world}: {how}; the film travels as one packed {WIDTH}x{HEIGHT}x4 float buffer",
                "host": mode,
            },
            "rays_per_sample": rays / samples,
        }
        if world > 1:
            out["rccl_ranks"] = 0 if rehearsal else world
            out["rehearsal"] = bool(rehearsal or (mode == "ranks" and backend != "nccl"))
        if timed_launches is not None:
            tl = timed_launches
            bpr = bytes_per_ray(ni, nl, nidx, ntri, closest=True)
            fpr = flops_per_ray(ni, ntri, 1.0)
            ext_rays, ext_ms, ext_launches = float(tl["extend_rays"]), tl["extend_ms"], max(1, tl["extend_launches"])
            sh_ms, sh_launches = tl["shade_ms"], max(1, tl["shade_launches"])
            ext_s = max(ext_ms, 1e-9) * 1e-3
            valu = fpr * ext_rays / ext_s / 1e12
            hbm = PMC["extend_hbm_bytes_per_ray"] * ext_rays / ext_s / 1e9
            shade_gbs = SHADE_BYTES_PER_PATH * ext_rays / (max(sh_ms, 1e-9) * 1e-3) / 1e9
            out["roofline"] = {
                "kernel": "k_extend / k_extend_refill (closest-hit KD traversal + triangle tests: lockstep for camera rays, refilling from the first "
                          "bounce on); the 8 KB of scene tables are staged in LDS",
                "bound": "valu",
                "achieved": valu,
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": valu / FP32_VALU_PEAK_TFLOPS,
                "flops_per_ray": fpr,
                "traffic": PMC["extend_hbm_bytes_per_ray"] * ext_rays / ext_launches,
                "traffic_unit": "HBM-side bytes per launch (PMC bytes per ray x this run's rays per launch)",
                "rays_per_launch": ext_rays / ext_launches,
                "avg_launch_ms": ext_ms / ext_launches,
                "launches": int(ext_launches),
                "timing": "HIP events on the launching streams, in a repeat of the first steps outside the timed region (the timed region "
                          "carries no events); three passes share the CUs as they do there",
                "per_ray": {"interior": ni, "leaf": nl, "leaf_idx": nidx, "tri_tests": ntri},
                "lanes_active_pmc": PMC["extend_lanes_active"],
                "vector_instructions_per_bounce_ray_pmc": PMC["extend_valu_per_ray"],
                "lds_bank_conflict_share_pmc": PMC["extend_lds_conflict_cycles"],
                "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                        "bytes_per_ray_pmc": PMC["extend_hbm_bytes_per_ray"], "bytes_per_ray_unavoidable": 41.7},
                "lds": {"bytes_per_ray_algorithmic": bpr - 44.0, "achieved_GBs": (bpr - 44.0) * ext_rays / ext_s / 1e9,
                        "note": "node, leaf-index and triangle bytes: served from LDS, not HBM traffic"},
                "pmc_source": PMC["source"],
                "note": "what binds is instruction issue under SIMD divergence (a wave issues every vector instruction for all 64 lanes "
                        "whatever the number of live ones), so frac is the useful-flop fraction of the FP32 vector peak; valu_issue is "
                        "the roof the step actually runs against",
                "valu_issue": {
                    "scope": "all kernels of a step, one device (the job's samples / n_gpus)",
                    "vector_instructions_per_camera_sample_pmc": PMC["valu_per_camera_sample"],
                    "achieved": PMC["valu_per_camera_sample"] * samples / (1 if rehearsal else world) / t / 1e9,
                    "peak": SIMDS * CLOCK_HZ / ISSUE_CYCLES_MIX / 1e9,
                    "unit": "G wave-instructions/s",
                    "frac": PMC["valu_per_camera_sample"] * samples / (1 if rehearsal else world) / t * ISSUE_CYCLES_MIX / (SIMDS * CLOCK_HZ),
                    "frac_if_every_instruction_cost_4_1_cycles": PMC["valu_per_camera_sample"] * samples / (1 if rehearsal else world) / t * ISSUE_CYCLES_SLOW / (SIMDS * CLOCK_HZ),
                    "cycles_per_instruction": {"plain two-operand f32 / integer on VGPRs": 2.2, "fma, min / max, compare, convert, packed, f64, "
                                               "any SGPR operand": 4.1, "rcp, sqrt": 8.1, "mix of the hot loops": ISSUE_CYCLES_MIX},
                    "source": "profiles/r03_valu_issue_rates.txt (tools/valu_rate.hip), " + PMC["source"],
                },
                "launch_overlap": "durations include time shared with the kernels of two other passes (three pipelines); 'exclusive' is "
                                  "the same kernels on the same queues with the passes run one after the other",
            }
            out["roofline_shade"] = {
                "kernel": "k_shade (one bounce of PathRenderer::Li per ray: NEE + MIS, BSDF sample, Russian roulette, queue appends)",
                "bound": "valu-issue",
                "bound_note": "large launches: a wave's round of 256 paths takes 19 k cycles, of which its SIMD spends 17 k issuing the ~1 380 "
                              "shading instructions of its three waves at 4.1 cycles each (profiles/r03_shade_phase_stamps.txt, "
                              "profiles/r03_valu_issue_rates.txt); the eleven small launches of a pass are latency-bound, which is what the "
                              "70 % waiting of the counters' average shows. 54 of 64 lanes live. The HBM figures below say how far the "
                              "kernel is from the byte roof",
                "achieved": shade_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": shade_gbs / HBM_PEAK_GBS,
                "bytes_per_path": SHADE_BYTES_PER_PATH,
                "traffic": PMC["shade_hbm_bytes_per_path"] * ext_rays / sh_launches,
                "traffic_unit": "HBM-side bytes per launch (PMC)",
                "avg_launch_ms": sh_ms / sh_launches,
                "launches": int(sh_launches),
                "lanes_active_pmc": PMC["shade_lanes_active"],
            }
            if exclusive is not None:
                xr, xms, xl = float(exclusive["extend_rays"]), exclusive["extend_ms"], max(1, exclusive["extend_launches"])
                xv = fpr * xr / (xms * 1e-3) / 1e12
                out["roofline"]["exclusive"] = {"achieved": xv, "unit": "TFLOP/s", "frac": xv / FP32_VALU_PEAK_TFLOPS, "avg_launch_ms": xms / xl,
                                                "rays_per_launch": xr / xl,
                                                "hbm_frac": PMC["extend_hbm_bytes_per_ray"] * xr / (xms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                if exclusive["shade_ms"] > 0:
                    xs = SHADE_BYTES_PER_PATH * xr / (exclusive["shade_ms"] * 1e-3) / 1e9
                    out["roofline_shade"]["exclusive"] = {"achieved": xs, "unit": "GB/s", "frac": xs / HBM_PEAK_GBS,
                                                          "avg_launch_ms": exclusive["shade_ms"] / max(1, exclusive["shade_launches"])}
        if strong:
            ts_, srays, ssamples = float(tmax[3]), float(tsum[4]), float(tsum[5])
            assert strong[3] >= STRONG_SPP, f"strong-scaling film incomplete: min weight {strong[3]}"
            out["strong_1024spp"] = {"seconds": ts_, "samples_per_sec": ssamples / ts_, "Mrays_per_s": srays / ts_ / 1e6,
                                     "per_gpu_passes": strong[4], "spp_per_gpu": STRONG_SPP // world, "n_gpus": world,
                                     "note": "one 1024-spp job split by sample index over the GPUs, film reduce included; max over ranks"}
        if world == 1 and not args.no_secondary:
            out["secondary"] = soup_secondary(torch, dev, scene.config, with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_c1:
            out["gpu_over_reference_cpu_c1"] = c1_job(torch, dev, scene)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, cfg)
            out["gpu_over_cpu"] = {"ratio": out["value"] / out["cpu_baseline"]["value"], "cores": out["cpu_baseline"]["cores"],
                                   "host_cpu_count": os.cpu_count(), "baseline": "the CPU port (oracle) on this host, same frame"}
            out["gpu_over_reference_cpu_cross_resolution"] = {
                "ratio": out["samples_per_sec"] / 1.617e6, "cores": 8,
                "note": "1080p samples per second here over the reference's 800x800 rate on 8 vCPU (1.617 M samples/s, measured by the "
                        "survey): half of the 16:9 frame's camera rays miss the box (3.67 rays per sample against 5.75), so this is not a "
                        "like-for-like ratio; gpu_over_reference_cpu_c1 is"}
        print(json.dumps(out))
    if mode == "ranks":
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
