#!/usr/bin/env python3
"""Ray batches for tools/sched_sim.cpp, made on the CPU with the oracle: the path tracer's camera rays of a band of the
1080p cornellBox frame in queue order (pixel order, one sample), the rays of its second bounce (hits only, compacted in
the same order, cosine-distributed about the shading side of the surface) and of its third bounce.
usage: tools/sched_rays.py out_dir [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import flatbin  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from raytracing_amd.scene import FlatScene  # noqa: E402


def normalize(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def cosine_about(n, rng):
    u = normalize(rng.standard_normal(n.shape).astype(np.float32))
    return normalize(n + 0.999 * u).astype(np.float32)


def main():
    out = sys.argv[1]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    os.makedirs(out, exist_ok=True)
    scene = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz"))
    cfg = scene.config.with_resolution(1920, 1080)
    orc = Oracle(scene, cfg)
    rng = np.random.default_rng(7)
    y0 = (1080 - rows) // 2
    ys, xs = np.meshgrid(np.arange(y0, y0 + rows), np.arange(1920), indexing="ij")
    film = np.stack([xs.ravel() + rng.random(xs.size), ys.ravel() + rng.random(xs.size)], axis=1).astype(np.float32)
    o, d = orc.camera_rays(film)
    flatbin.write_scene(os.path.join(out, "scene.bin"), scene, cfg, 0, 1, 15, 0, 1)
    tri = scene.tri_p.reshape(-1, 3, 3)
    ng = normalize(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])).astype(np.float32)
    for b in range(4):
        flatbin.write_rays(os.path.join(out, f"rays_b{b}.bin"), o, d)
        prim, t, _ = orc.trace_closest(o, d)[:3]
        hit = prim >= 0
        print(f"bounce {b}: {len(o)} rays, hit fraction {hit.mean():.3f}")
        p = (o + t[:, None] * d)[hit].astype(np.float32)
        n = ng[prim[hit]]
        n = np.where(((n * d[hit]).sum(1) > 0)[:, None], -n, n)
        o, d = p, cosine_about(n, rng)


if __name__ == "__main__":
    main()
