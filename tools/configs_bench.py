#!/usr/bin/env python3
"""The five BASELINE configurations at their stated sizes on one MI355X, one JSON line each: render time of the whole job
(device-resident film, inputs uploaded before), samples/s and rays/s, next to the reference's own time where the survey
measured one. usage: tools/configs_bench.py [configs, e.g. 1,2,3,4,5]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import raygen  # noqa: E402
from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM  # noqa: E402
from raytracing_amd.device import Context  # noqa: E402
from raytracing_amd.scene import FlatScene  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def job(ctx, name, scene, cfg, reference_seconds=None, host_build_seconds=None, repeats=2, **kw):
    dev = torch.device("cuda", 0)
    ctx.upload_scene(scene)
    ctx.set_camera(cfg)
    rgb = torch.zeros((cfg.height, cfg.width, 3), dtype=torch.float32, device=dev)
    w = torch.zeros((cfg.height, cfg.width), dtype=torch.float32, device=dev)
    best = None
    for _ in range(repeats + 1):   # the first run allocates the queues
        rgb.zero_()
        w.zero_()
        ctx.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.render_dev(rgb.data_ptr(), w.data_ptr(), **kw)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        best = dt if best is None else min(best, dt)
    rays = st["rays_closest"] + st["rays_any"]
    out = {"config": name, "resolution": [cfg.width, cfg.height], "spp": kw["spp"], "max_depth": kw["max_depth"], "seconds": best,
           "Msamples_per_s": st["camera_samples"] / best / 1e6, "Mrays_per_s": rays / best / 1e6, "rays_per_sample": rays / st["camera_samples"],
           "n_prims": scene.n_prims, "n_nodes": scene.n_nodes, "tables_in_lds": bool(st["scene_in_lds"]),
           "mean_rgb": [float(v) for v in (rgb.sum((0, 1)) / w.sum()).cpu()]}
    if host_build_seconds is not None:
        out["host_build_seconds"] = host_build_seconds
    if reference_seconds is not None:
        out["reference_seconds_8vcpu"] = reference_seconds
        out["gpu_over_reference"] = reference_seconds / best
    print(json.dumps(out), flush=True)


def main():
    which = set(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,5").split(","))
    path_scene = FlatScene.load_npz(os.path.join(GOLDEN, "cornellBox_path.npz"))
    whitted_scene = FlatScene.load_npz(os.path.join(GOLDEN, "cornellBox_whitted.npz"))
    ctx = Context(0)
    if 1 in which:   # the reference's own job, counter streams (202.6 s on 8 vCPU, measured on the compiled reference by the survey)
        job(ctx, "C1 cornellBox PathTracing 800x800 x 512 spp, depth 15", path_scene, path_scene.config, reference_seconds=202.6,
            integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=1, spp=512, max_depth=15)
    if 2 in which:   # deterministic: the reference's tile streams (2.05 s on 8 vCPU)
        job(ctx, "C2 cornellBox WhittedTracing 666x500 x 64 spp, depth 15, reference tile streams", whitted_scene, whitted_scene.config,
            reference_seconds=2.05, integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, seed=0, spp=64, max_depth=15)
    if 3 in which:
        job(ctx, "C3 cornellBox PathTracing 1920x1080 x 1024 spp, depth 15", path_scene, path_scene.config.with_resolution(1920, 1080),
            integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=1, spp=1024, max_depth=15)
    if 4 in which:
        t0 = time.perf_counter()
        scene, cfg = raygen.soup_scene(1_000_000, path_scene.config)
        job(ctx, "C4 1 M-triangle soup PathTracing 1920x1080 x 1024 spp, depth 15", scene, cfg, host_build_seconds=time.perf_counter() - t0, repeats=1,
            integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=1, spp=1024, max_depth=15)
    if 5 in which:
        t0 = time.perf_counter()
        scene, cfg = raygen.grid_scene(2048, 1024, path_scene.config)
        job(ctx, "C5 4.19 M-triangle mesh, Lambertian / mirror / glass, PathTracing 1920x1080 x 1024 spp, depth 8", scene, cfg,
            host_build_seconds=time.perf_counter() - t0, repeats=1, integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=1, spp=1024,
            max_depth=8)
    ctx.close()


if __name__ == "__main__":
    main()
