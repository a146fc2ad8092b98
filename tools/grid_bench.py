#!/usr/bin/env python3
"""Traversal and path-tracing throughput on a large connected mesh with mixed materials (tables in HBM): a displaced
height field of nx x nz cells (default 2048 x 1024 = 4.19 M triangles), materials by stripe across x - 60 % Lambertian,
25 % mirror, 15 % glass (eta 1.5) - under the cornellBox luminaire, depth 8, 1920x1080. The KD-tree is built by the
host front-end. usage: tools/grid_bench.py [nx=2048] [nz=1024] [n_rays=4194304] [repeats=5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402

from raytracing_amd.scene import FlatScene  # noqa: E402
import raygen  # noqa: E402
from soup_bench import measure  # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    nz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    t0 = time.perf_counter()
    base = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz")).config
    scene, cfg = raygen.grid_scene(nx, nz, base)
    n_tris = scene.n_prims - 2
    build_s = time.perf_counter() - t0
    print(f"# grid: {n_tris} triangles, {len(scene.nodes)} nodes, {len(scene.leaf_idx)} leaf indices, build {build_s:.1f} s "
          f"({scene.tree_stats})", flush=True)
    measure(scene, cfg, f"{n_tris}-triangle displaced grid (60/25/15 Lambertian/mirror/glass)", n, reps, 8)


if __name__ == "__main__":
    main()
