import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from raytracing_amd.scene import FlatScene
from raytracing_amd.device import Context
from oracle_lib import Oracle
s = FlatScene.load_npz(os.path.join(ROOT, 'tests/golden/cornellBox_path.npz'))
ctx = Context(0)
ctx.upload_scene(s)
for (res, spp, depth, pp) in [(64,1,0,1),(64,1,1,1),(64,1,2,1),(64,1,3,1),(64,1,15,1),(64,4,15,4),(64,4,15,1),(64,4,15,2),(128,16,15,4),(128,16,15,16)]:
    cfg = s.config.with_resolution(res, res, spp=spp)
    orc = Oracle(s, cfg)
    ctx.set_camera(cfg)
    ctx.reset_stats()
    g, gw = ctx.render(integrator=0, rng_mode=0, seed=1, max_depth=depth, spp_per_pass=pp)
    st = ctx.stats()
    c, cw, cs = orc.render(integrator=0, rng_mode=0, seed=1, max_depth=depth)
    eq = (g == c).all(axis=2)
    close = np.isclose(g, c, rtol=1e-5, atol=1e-7).all(axis=2)
    print(f"res {res} spp {spp} depth {depth} pass {pp}: exact {eq.mean():.4f} close {close.mean():.4f} maxabs {np.abs(g-c).max():.4g} meanG {g.mean():.5f} meanC {c.mean():.5f} rays g {st['rays_closest']},{st['rays_any']} c {cs['rays_closest']},{cs['rays_any']}")
