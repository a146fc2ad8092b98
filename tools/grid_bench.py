#!/usr/bin/env python3
"""Traversal and path-tracing throughput on a large connected mesh with mixed materials (tables in HBM): a displaced
height field of nx x nz cells (default 2048 x 1024 = 4.19 M triangles), materials by stripe across x - 60 % Lambertian,
25 % mirror, 15 % glass (eta 1.5) - under the cornellBox luminaire, depth 8, 1920x1080. The KD-tree is built by the
host front-end. usage: tools/grid_bench.py [nx=2048] [nz=1024] [n_rays=4194304] [repeats=5]"""
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402

from raytracing_amd.scene import FlatScene, build_scene, make_lights, make_materials  # noqa: E402
import raygen  # noqa: E402
from soup_bench import measure  # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    nz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    t0 = time.perf_counter()
    tri, stripe = raygen.displaced_grid(nx, nz, seed=5)
    n_tris = len(tri)
    # 20 stripes, each split 60/25/15 between the three materials
    phase = (stripe * 20.0) % 1.0
    mat = np.where(phase < 0.60, 0, np.where(phase < 0.85, 1, 2)).astype(np.int32)
    lum = np.array([[343, 548.7, 227, 343, 548.7, 332, 213, 548.7, 332], [343, 548.7, 227, 213, 548.7, 332, 213, 548.7, 227]], np.float32)
    tri = np.concatenate([tri, lum])
    mat = np.concatenate([mat, np.zeros(2, np.int32)])
    light_id = np.full(n_tris + 2, -1, np.int32)
    light_id[n_tris:] = [0, 1]
    base = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz")).config
    cfg = dataclasses.replace(base, eye=(278.0, 520.0, -420.0), focus=(278.0, 120.0, 278.0)).with_resolution(1920, 1080, spp=4)
    materials = make_materials([(0, (0.73, 0.73, 0.73), 1.0), (1, (0.9, 0.9, 0.9), 1.0), (2, (1.0, 1.0, 1.0), 1.5)])
    scene = build_scene(tri, mat, materials, light_id=light_id,
                        lights=make_lights([(n_tris, (17, 12, 4), 0), (n_tris + 1, (17, 12, 4), 0)]), config=cfg)
    build_s = time.perf_counter() - t0
    print(f"# grid: {n_tris} triangles, {len(scene.nodes)} nodes, {len(scene.leaf_idx)} leaf indices, build {build_s:.1f} s "
          f"({scene.tree_stats})", flush=True)
    measure(scene, cfg, f"{n_tris}-triangle displaced grid (60/25/15 Lambertian/mirror/glass)", n, reps, 8)


if __name__ == "__main__":
    main()
