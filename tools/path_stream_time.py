#!/usr/bin/env python3
"""Times the reference's default scene (cornellBox-PathTracing: 800x800, 512 spp, depth 15) on the device with the
reference's per-tile random stream and with counter streams, and compares both with the image the reference publishes.
usage: tools/path_stream_time.py [spp=512]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd.device import Context  # noqa: E402
from raytracing_amd.pngio import read_png  # noqa: E402
from raytracing_amd.scene import FlatScene  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
scene = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz"))
ref = read_png(os.path.join(ROOT, "tests", "golden", "reference_cornellBox_path_800x800_512spp.png")).astype(np.float64)
ctx = Context(0)
ctx.upload_scene(scene)
ctx.set_camera(scene.config)
for mode, name in ((1, "reference tile stream"), (0, "counter streams")):
    ctx.render(rng_mode=mode, spp=2)   # warm-up
    t0 = time.perf_counter()
    rgb, w = ctx.render(rng_mode=mode, spp=spp)
    dt = time.perf_counter() - t0
    img = ctx.develop(rgb, w, scene.config.film_scale).astype(np.float64)
    psnr = 10 * np.log10(255.0 ** 2 / ((img - ref) ** 2).mean())
    n = scene.config.width * scene.config.height * spp
    print(f"{name}: {dt:.2f} s, {n / dt / 1e6:.1f} M samples/s, PSNR against the published image {psnr:.2f} dB", flush=True)
