#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the reference's scene files (run in the build container, where
/root/reference is mounted). The .npz files hold flattened scene DATA only: KD-tree node words, leaf index list,
world-space triangles, material / light tables and the render configuration, produced by this repository's own
host front-end (librt_host.so). The GPU box has no /root/reference; tests and bench.py load these files instead."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracing_amd.scene import load_scene_json  # noqa: E402

REF = os.environ.get("RT_REFERENCE", "/root/reference")
SCENES = {
    "cornellBox_path": "scenes/cornellBox-PathTracing/cornellBox.json",
    "cornellBox_whitted": "scenes/cornellBox-WhittedTracing/cornellBox.json",
    "cornell_spheres": "scenes/cornellBox-PathTracing/cornell.json",
}

if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, rel in SCENES.items():
        s = load_scene_json(os.path.join(REF, rel))
        s.save_npz(os.path.join(out, name + ".npz"))
        print(name, s.tree_stats, "prims", s.n_prims, "lights", len(s.lights), "res", s.config.width, s.config.height, "spp", s.config.spp)
