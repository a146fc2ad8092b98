#!/usr/bin/env python3
"""One 32-spp pass of the cornellBox 1080p path tracer with the passes' kernels one after the other (one pipeline), for a
rocprofv3 --kernel-trace run: tools/pass_trace.sh prints every launch of the pass with its duration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracing_amd.device import Context  # noqa: E402
from raytracing_amd.scene import FlatScene  # noqa: E402

scene = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz"))
cfg = scene.config.with_resolution(1920, 1080, spp=32)
ctx = Context(0)
ctx.upload_scene(scene)
ctx.set_camera(cfg)
ctx.set_pipelines(int(os.environ.get("RT_PT_PIPELINES", "1")))
dev = torch.device("cuda", 0)
rgb = torch.zeros((1080, 1920, 3), dtype=torch.float32, device=dev)
w = torch.zeros((1080, 1920), dtype=torch.float32, device=dev)
for _ in range(2):
    ctx.render_dev(rgb.data_ptr(), w.data_ptr(), integrator=0, rng_mode=0, seed=1, spp=32, max_depth=15, spp_per_pass=32)
    ctx.synchronize()
ctx.close()
