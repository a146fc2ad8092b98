#!/usr/bin/env python3
"""Time of rt_upload_scene (validation, device tables, copies) on the two large synthetic scenes. usage: tools/upload_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import raygen  # noqa: E402
from raytracing_amd.device import Context  # noqa: E402
from raytracing_amd.scene import FlatScene  # noqa: E402

base = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz")).config
ctx = Context(0)
for name, make in (("1 M soup", lambda: raygen.soup_scene(1_000_000, base)), ("4.19 M mesh", lambda: raygen.grid_scene(2048, 1024, base))):
    t0 = time.perf_counter()
    scene, cfg = make()
    t1 = time.perf_counter()
    ctx.upload_scene(scene)
    ctx.synchronize()
    t2 = time.perf_counter()
    ctx.upload_scene(scene)
    ctx.synchronize()
    t3 = time.perf_counter()
    print(f"{name}: generate + host build {t1 - t0:.2f} s, first upload {t2 - t1:.2f} s, second upload {t3 - t2:.2f} s", flush=True)
ctx.close()
