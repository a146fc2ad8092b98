#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: cornellBox PathTracing at 1920x1080 (BASELINE config 3).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one wavefront pass of SPP_PER_STEP samples per pixel over the whole 1080p frame on every rank
(raygen, then extend/shade/shadow/probe for each bounce up to depth 15, then the film accumulation). Rank g
renders sample indices [g*K*P, (g+1)*K*P) of a (N*K*P)-spp job; after the K steps the float film is
sum-reduced to rank 0 over RCCL inside the timed region. The scene is resident in HBM and camera rays are
generated on the device, so no host data moves in the timed region.

Prints ONE JSON line (rank 0). value = whole-job Mrays/s, one ray = one KD-tree traversal of either flavour
(extend + MIS-probe + shadow), counted by the kernels themselves.
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
SPP_PER_STEP = int(os.environ.get("RT_BENCH_SPP_PER_STEP", "32"))
MAX_DEPTH = 15
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak
FP32_VALU_PEAK_TFLOPS = 157.3
# HBM-side bytes per ray of k_extend from the hardware counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# passes over this workload, profiles/r01_pmc_bench_steps4_v13.md): 2 x 13.5 B read (gfx950 counts the 16-B-per-lane
# coalesced reads at half their bytes; camera rays have no origin record) + 17.0 B written. Counters cannot be read
# inside a timed run, so the figure measured under the profiler is scaled by this run's rays per launch.
EXTEND_HBM_BYTES_PER_RAY_PMC = 44.0


def bytes_per_ray(ni, nl, nidx, ntri, closest=True):
    """Algorithmic traffic of one traversal: ray in (28 B), hit out (16 B; 1 B for any-hit), 8 B per node visited,
    4 B per leaf index, 36 B per triangle tested."""
    return 28 + (16 if closest else 1) + 8 * (ni + nl) + 4 * nidx + 36 * ntri


def flops_per_ray(ni, ntri, ncand):
    return 40 + 6 * ni + 44 * ntri + 35 * ncand


def cpu_baseline(scene, cfg, target_seconds=15.0):
    """Times the CPU oracle (a port of the reference's path; the reference's sources cannot travel to this
    machine) on all host cores given to this job, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
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
        "kind": "port",
        "sample": f"cornellBox PathTracing {cfg.width}x{cfg.height}, depth {MAX_DEPTH}, {spp} spp (of the GPU job's), {cores} threads",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd.device import Context
    from raytracing_amd.scene import FlatScene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the device path has no CPU fallback")
    local_dev = local_rank % torch.cuda.device_count()   # one rank per GPU; wraps only in a rehearsal with fewer GPUs than ranks
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # RCCL is the transport; "gloo" (RT_BENCH_BACKEND) only rehearses the multi-rank logic on a box with fewer GPUs than
    # ranks, where RCCL refuses two ranks on one device - it stages the film through host memory and is not a measurement
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if world > 1:
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
    ctx = Context(local_dev)
    ctx.upload_scene(scene)
    ctx.set_camera(cfg)

    K, W, P = args.steps, args.warmup, SPP_PER_STEP
    spp_total = world * K * P
    rgb = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=dev)
    wsum = torch.zeros((HEIGHT, WIDTH), dtype=torch.float32, device=dev)

    def run(spp_begin, spp_end, total, collect=False):
        ctx.render_dev(rgb.data_ptr(), wsum.data_ptr(), integrator=0, rng_mode=0, seed=1, spp=total, max_depth=MAX_DEPTH,
                       spp_begin=spp_begin, spp_end=spp_end, spp_per_pass=P, collect_stats=collect)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # per-ray traversal work (node / leaf / triangle counts), measured once outside the timed region
    ctx.reset_stats()
    run(0, 1, max(1, W * P) + 1, collect=True)
    ctx.synchronize()
    st = ctx.stats()
    cr = max(1, st["counted_rays"])
    ni, nl, nidx, ntri = st["n_interior"] / cr, st["n_leaf"] / cr, st["n_leaf_idx"] / cr, st["n_tri_tests"] / cr
    # warmup steps (untimed)
    if W > 0:
        run(0, W * P, W * P)
    if world > 1:
        # first use of the collectives (communicator set-up, channel buffers) belongs to the warm-up, not to the timed steps
        ctx.synchronize()
        reduce_to_rank0(rgb)
        reduce_to_rank0(wsum)
    fence()
    rgb.zero_()
    wsum.zero_()
    ctx.reset_stats()
    fence()

    t0 = time.perf_counter()
    run(rank * K * P, (rank + 1) * K * P, spp_total)
    if world > 1:
        ctx.synchronize()   # the render runs on the library's own streams; RCCL runs on torch's
        reduce_to_rank0(rgb)
        reduce_to_rank0(wsum)
    fence()
    elapsed = time.perf_counter() - t0

    st = ctx.stats()
    # k_extend alone on the device: one more step with the passes run one after the other (untimed). In the timed region
    # kernels of three passes share the CUs, which is faster overall but stretches every single launch.
    exclusive = None
    if rank == 0:
        ctx.set_pipelines(1)
        try:
            ctx.reset_stats()
            run(0, P, P)
            ctx.synchronize()
            sx = ctx.stats()
            if sx["extend_ms"] > 0:
                exclusive = (float(sx["extend_rays"]), sx["extend_ms"], max(1, sx["extend_launches"]))
        finally:
            ctx.set_pipelines(3)
    local = torch.tensor([elapsed, float(st["rays_closest"] + st["rays_any"]), float(st["camera_samples"]), float(st["extend_rays"]),
                          st["extend_ms"], float(st["extend_launches"])], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    tmax = local.clone()
    tsum = local.clone()
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    t = float(tmax[0])
    rays, samples = float(tsum[1]), float(tsum[2])

    if rank == 0:
        # sanity: every pixel received exactly spp_total samples (plus rare border-rounded neighbours)
        wmin = float(wsum.min())
        assert wmin >= spp_total, f"film incomplete: min weight {wmin} < {spp_total}"
        bpr = bytes_per_ray(ni, nl, nidx, ntri, closest=True)
        ext_rays, ext_ms, ext_launches = float(st["extend_rays"]), st["extend_ms"], max(1, st["extend_launches"])
        achieved = bpr * ext_rays / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
        fpr = flops_per_ray(ni, ntri, 1.0)
        valu = fpr * ext_rays / (ext_ms * 1e-3) / 1e12 if ext_ms > 0 else 0.0
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
                "workload": f"cornellBox PathTracing {WIDTH}x{HEIGHT}, depth {MAX_DEPTH}, {P} spp per step, counter RNG (BASELINE configs[2])",
                "spp_per_gpu": K * P,
                "spp_total": spp_total,
                "paths_in_flight_per_gpu": WIDTH * HEIGHT * P,
                "parallelism": f"spp-sharded x{world}, one {'RCCL' if backend == 'nccl' else backend + ' (rehearsal)'} sum-reduce of the {WIDTH}x{HEIGHT}x4 float film",
            },
            "rays_per_sample": rays / samples,
            "roofline": {
                "kernel": "k_extend (closest-hit KD traversal + triangle tests)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": EXTEND_HBM_BYTES_PER_RAY_PMC * ext_rays / ext_launches,
                "traffic_unit": "bytes per launch (HBM side, PMC)",
                "bytes_per_ray": bpr,
                "rays_per_launch": ext_rays / ext_launches,
                "avg_launch_ms": ext_ms / ext_launches,
                "launches": int(ext_launches),
                "per_ray": {"interior": ni, "leaf": nl, "leaf_idx": nidx, "tri_tests": ntri},
                "note": "algorithmic bytes (nodes, leaf indices and triangles are served from LDS for this 8 KB scene, so only the "
                        "ray and hit records are HBM traffic); the kernel is bound by SIMD divergence (12.6 of 64 lanes active per "
                        "VALU instruction, PMC), see valu",
                "valu": {"achieved": valu, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu / FP32_VALU_PEAK_TFLOPS,
                         "flops_per_ray": fpr},
                "extend_share_of_step": ext_ms * 1e-3 / t,
                "launch_overlap": "durations in the timed region include time shared with the kernels of two other passes "
                                  "(three pipelines); 'exclusive' is the same kernel on the same queues with the passes "
                                  "run one after the other, outside the timed region",
            },
        }
        if exclusive is not None:
            xr, xms, xl = exclusive
            xa = bpr * xr / (xms * 1e-3) / 1e9
            out["roofline"]["exclusive"] = {"achieved": xa, "unit": "GB/s", "frac": xa / HBM_PEAK_GBS, "avg_launch_ms": xms / xl,
                                            "rays_per_launch": xr / xl}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, cfg)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
