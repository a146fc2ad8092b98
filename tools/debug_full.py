import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from raytracing_amd.scene import FlatScene, develop
from raytracing_amd.device import Context
from raytracing_amd.pngio import read_png, write_png
s = FlatScene.load_npz(os.path.join(ROOT, 'tests/golden/cornellBox_path.npz'))
ref = read_png(os.path.join(ROOT, 'tests/golden/reference_cornellBox_path_800x800_512spp.png'))[..., :3].astype(np.float64)
ctx = Context(0)
ctx.upload_scene(s)
def psnr(a,b):
    d=a.astype(np.float64)-b; return 10*np.log10(255**2/(d**2).mean())
for spp, pp in [(512,0),(512,4),(64,0),(64,1)]:
    ctx.reset_stats()
    t=time.time()
    rgb, w = ctx.render(integrator=0, rng_mode=0, seed=2024, spp=spp, spp_per_pass=pp)
    dt=time.time()-t
    st=ctx.stats()
    img = develop(rgb, w)
    d = img.astype(np.float64)-ref
    print(f"spp {spp} pass {pp}: {dt:.3f}s render_ms {st['render_ms']:.1f} extend_ms {st['extend_ms']:.1f} launches {st['extend_launches']} samples {st['camera_samples']} rays {st['rays_closest']} {st['rays_any']} wmin {w.min()} wmax {w.max()} PSNR {psnr(img,ref):.2f} meandiff {d.mean(axis=(0,1))} nan {np.isnan(rgb).sum()}")
    write_png(os.path.join(ROOT, f'gpurun_out/gpu_{spp}_{pp}.png'), img)
    # block-level diff map
    blk = np.abs(d).reshape(8,100,8,100,3).mean(axis=(1,3,4))
    print(np.round(blk,2))
