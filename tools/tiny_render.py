import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd.scene import FlatScene
from raytracing_amd.device import Context
s = FlatScene.load_npz(os.path.join(ROOT, 'tests/golden/cornellBox_path.npz'))
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = s.config.with_resolution(res, res, spp=8)
ctx = Context(0); ctx.upload_scene(s); ctx.set_camera(cfg)
for i in range(6):
    ctx.render(integrator=0, rng_mode=0, seed=i, spp=8, spp_per_pass=8)
print(ctx.stats())
