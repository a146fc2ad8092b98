#!/usr/bin/env python3
"""Times the Whitted configuration (cornellBox-WhittedTracing: 666x500, 64 spp, depth 15) on the device, with the
reference's per-tile random stream and with counter streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd.device import Context  # noqa: E402
from raytracing_amd.scene import FlatScene  # noqa: E402

scene = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_whitted.npz"))
ctx = Context(0)
ctx.upload_scene(scene)
ctx.set_camera(scene.config)
for mode, name in ((1, "reference tile stream"), (0, "counter streams")):
    ctx.render(rng_mode=mode)
    t0 = time.perf_counter()
    ctx.render(rng_mode=mode)
    dt = time.perf_counter() - t0
    n = scene.config.width * scene.config.height * scene.config.spp
    print(f"{name}: {dt * 1e3:.1f} ms, {n / dt / 1e6:.1f} M samples/s, prepass count passes {ctx._lib.rt_debug_whitted_passes(ctx._h)}")
