#!/usr/bin/env python3
"""Work-queue tracer against the lockstep tracer on a soup, with the integrity guard printed. usage: queue_check.py [n_tris] [n_rays]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raygen
from raytracing_amd.device import Context
from raytracing_amd.scene import FlatScene

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
base = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz")).config
scene, cfg = raygen.soup_scene(n_tris, base)
print("built", scene.n_nodes, flush=True)
ctx = Context(0)
ctx.upload_scene(scene)
from raytracing_amd.scene import camera_ray
side = int(np.sqrt(n))
u = (np.arange(side, dtype=np.float32) + 0.5) / side
gx, gy = np.meshgrid(u, u)
eye, d00 = [np.asarray(v, np.float32) for v in camera_ray(cfg, 0.5, 0.5)]
d10 = np.asarray(camera_ray(cfg, cfg.width - 0.5, 0.5)[1], np.float32)
d01 = np.asarray(camera_ray(cfg, 0.5, cfg.height - 0.5)[1], np.float32)
dc = d00 + gx.reshape(-1, 1) * (d10 - d00) + gy.reshape(-1, 1) * (d01 - d00)
dc = (dc / np.sqrt((dc * dc).sum(1, keepdims=True))).astype(np.float32)
oc = np.broadcast_to(eye, dc.shape).copy()
n = len(oc)
for name, (o, d) in (("camera", (oc, dc)), ("incoherent", raygen.incoherent_rays(scene.bounds, n, 22))):
    ctx.set_trace_schedule("lockstep")
    lp, lt, lb = ctx.trace_closest(o, d)
    ctx.set_trace_schedule("queue")
    ctx.reset_stats()
    qp, qt, qb = ctx.trace_closest(o, d)
    st = ctx.stats()
    print(name, "guard", st["queue_guard"], "ids differ", int((lp != qp).sum()), "t differ", int((lt.view(np.uint32) != qt.view(np.uint32)).sum()), flush=True)
    qa = ctx.trace_any(o, d, np.full(n, 40.0, np.float32))
    ctx.set_trace_schedule("lockstep")
    la = ctx.trace_any(o, d, np.full(n, 40.0, np.float32))
    print(name, "any differ", int((la != qa).sum()), "guard", ctx.stats()["queue_guard"], flush=True)
ctx.close()
