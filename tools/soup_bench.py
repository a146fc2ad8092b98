#!/usr/bin/env python3
"""Traversal throughput on a triangle soup that does NOT fit in LDS (the scene tables live in HBM / L2 /
Infinity Cache): n random small triangles, centres uniform in [50,500]^3, edges 0.5..5 long, SAH KD-tree
built by the host front-end. Prints one JSON line per ray distribution.
usage: tools/soup_bench.py [n_triangles=1000000] [n_rays=4194304] [repeats=5]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracing_amd.device import Context  # noqa: E402
from raytracing_amd.scene import FlatScene, camera_ray  # noqa: E402
import raygen  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main():
    n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    t0 = time.perf_counter()
    base = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz")).config
    scene, cfg = raygen.soup_scene(n_tris, base)
    build_s = time.perf_counter() - t0
    print(f"# soup: {n_tris} triangles, {len(scene.nodes)} nodes, {len(scene.leaf_idx)} leaf indices, build {build_s:.1f} s "
          f"({scene.tree_stats})", flush=True)
    measure(scene, cfg, f"{n_tris}-triangle soup", n, reps, 15)


def measure(scene, cfg, label, n, reps, depth):
    # RT_HBM_WHAT: comma list of primary, incoherent, path (default: all); RT_HBM_SCHEDULES: lockstep, stream (default: both).
    # Profiler runs name one of each so that the counters belong to one kernel on one workload.
    what = set(os.environ.get("RT_HBM_WHAT", "primary,incoherent,path").split(","))
    schedules = os.environ.get("RT_HBM_SCHEDULES", "lockstep,stream").split(",")
    ctx = Context(0)
    ctx.upload_scene(scene)
    dev = torch.device("cuda", 0)
    prim = torch.empty(n, dtype=torch.int32, device=dev)
    t = torch.empty(n, dtype=torch.float32, device=dev)
    bary = torch.empty((n, 3), dtype=torch.float32, device=dev)

    def timed(fn):
        fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps

    def run(name, o_np, d_np):
        # per-ray work from the instrumented kernels on a sample of the batch
        step = max(1, len(o_np) // 200_000)
        ctx.reset_stats()
        ctx.trace_closest(np.ascontiguousarray(o_np[::step]), np.ascontiguousarray(d_np[::step]))
        st = ctx.stats()
        cr = max(1, st["counted_rays"])
        ni, nl, nidx, ntri = st["n_interior"] / cr, st["n_leaf"] / cr, st["n_leaf_idx"] / cr, st["n_tri_tests"] / cr
        bpr = 28 + 16 + 8 * (ni + nl) + 4 * nidx + 36 * ntri
        o = torch.tensor(o_np, device=dev)
        d = torch.tensor(d_np, device=dev)
        out = {"workload": f"{label}, {name} rays", "n_rays": n,
               "per_ray": {"interior": ni, "leaf": nl, "leaf_idx": nidx, "tri_tests": ntri}, "bytes_per_ray": bpr}
        for key in schedules:
            ctx.set_trace_schedule(key)
            dt = timed(lambda: ctx.trace_closest_dev(n, o.data_ptr(), d.data_ptr(), 0, prim.data_ptr(), t.data_ptr(), bary.data_ptr()))
            out[key] = {"Mrays_per_s": n / dt / 1e6, "ms": dt * 1e3, "achieved_GBs": bpr * n / dt / 1e9,
                        "frac_of_hbm_peak": bpr * n / dt / 1e9 / HBM_PEAK_GBS}
        ctx.set_trace_schedule("auto")
        out["hit_fraction"] = float((prim >= 0).float().mean())
        print(json.dumps(out), flush=True)

    # primary rays: the camera's pinhole, regular grid of film positions in scanline order
    side = int(np.sqrt(n))
    u = (np.arange(side, dtype=np.float32) + 0.5) / side
    gx, gy = np.meshgrid(u, u)
    eye, d00 = [np.asarray(v, np.float32) for v in camera_ray(cfg, 0.5, 0.5)]
    d10 = np.asarray(camera_ray(cfg, cfg.width - 0.5, 0.5)[1], np.float32)
    d01 = np.asarray(camera_ray(cfg, 0.5, cfg.height - 0.5)[1], np.float32)
    d = d00 + gx.reshape(-1, 1) * (d10 - d00) + gy.reshape(-1, 1) * (d01 - d00)
    d /= np.sqrt((d * d).sum(1, keepdims=True)).astype(np.float32)
    if len(d) < n:
        d = np.concatenate([d, d[: n - len(d)]])
    o = np.broadcast_to(eye, d.shape).copy()
    if "primary" in what:
        run("primary", o, d.astype(np.float32))
    if "incoherent" in what:
        o, d = raygen.incoherent_rays(scene.bounds, n, 3)
        run("incoherent", o, d)
    if "path" not in what:
        ctx.close()
        return

    # path tracing (PathRenderer::Li) at 1920x1080, 16 spp in one pass (RT_HBM_SPP / RT_HBM_SPP_PASS: other job and pass sizes)
    spp = int(os.environ.get("RT_HBM_SPP", "16"))
    spp_pass = int(os.environ.get("RT_HBM_SPP_PASS", "0"))
    ctx.set_camera(cfg)
    rgb = torch.zeros((cfg.height, cfg.width, 3), dtype=torch.float32, device=dev)
    w = torch.zeros((cfg.height, cfg.width), dtype=torch.float32, device=dev)
    for schedule in schedules:
        ctx.set_trace_schedule(schedule)
        ctx.render_dev(rgb.data_ptr(), w.data_ptr(), integrator=0, rng_mode=0, seed=1, spp=spp, max_depth=depth, spp_per_pass=spp_pass)
        ctx.synchronize()
        rgb.zero_()
        w.zero_()
        ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.render_dev(rgb.data_ptr(), w.data_ptr(), integrator=0, rng_mode=0, seed=1, spp=spp, max_depth=depth, spp_per_pass=spp_pass)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        rays = st["rays_closest"] + st["rays_any"]
        print(json.dumps({"workload": f"{label}, PathTracing 1920x1080, {spp} spp, depth {depth}", "schedule": schedule, "seconds": dt,
                          "Mrays_per_s": rays / dt / 1e6, "Msamples_per_s": st["camera_samples"] / dt / 1e6,
                          "rays_per_sample": rays / max(1, st["camera_samples"]), "extend_ms": st["extend_ms"],
                          "mean_rgb": [float(v) for v in (rgb.sum((0, 1)) / w.sum()).cpu()]}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
