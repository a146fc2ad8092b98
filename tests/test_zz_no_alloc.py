"""rt_render_noalloc_dev - the entry point that allocates nothing (its queues, counters and spill buffers live in a
static array inside the library) - against rt_render_dev on the same inputs, through the C ABI. Same kernels, same
arithmetic: film sums must be bit-equal, on two calls in a row, the second with a smaller film."""
import numpy as np
import pytest

import raygen
from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM

pytestmark = pytest.mark.gpu


def _pair(ctx, torch, cfg, **kw):
    """The same job through rt_render_dev and rt_render_noalloc_dev; returns the four film arrays."""
    dev = torch.device("cuda", 0)
    out = []
    for call in (ctx.render_dev, ctx.render_noalloc_dev):
        rgb = torch.zeros((cfg.height, cfg.width, 3), dtype=torch.float32, device=dev)
        w = torch.zeros((cfg.height, cfg.width), dtype=torch.float32, device=dev)
        call(rgb.data_ptr(), w.data_ptr(), **kw)
        ctx.synchronize()
        out += [rgb.cpu().numpy(), w.cpu().numpy()]
    return out


def test_noalloc_render_equals_render_dev(path_scene):
    import torch
    from raytracing_amd.device import Context, RtError
    with Context(0) as ctx:
        ctx.upload_scene(path_scene)
        big = path_scene.config.with_resolution(320, 200, spp=6)
        small = path_scene.config.with_resolution(96, 64, spp=5)
        dev = torch.device("cuda", 0)

        # first call on a fresh context, before any other render: nothing it needs may depend on an earlier allocation
        ctx.set_camera(big)
        kw = dict(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=11, spp=6, max_depth=15)
        rgb_n = torch.zeros((200, 320, 3), dtype=torch.float32, device=dev)
        w_n = torch.zeros((200, 320), dtype=torch.float32, device=dev)
        ctx.render_noalloc_dev(rgb_n.data_ptr(), w_n.data_ptr(), **kw)
        # second call in a row, smaller film: the scratch still holds the first call's queues and counters
        ctx.set_camera(small)
        kw_s = dict(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=12, spp=5, max_depth=15)
        rgb_s = torch.zeros((64, 96, 3), dtype=torch.float32, device=dev)
        w_s = torch.zeros((64, 96), dtype=torch.float32, device=dev)
        free_before = torch.cuda.mem_get_info(0)[0]
        ctx.render_noalloc_dev(rgb_s.data_ptr(), w_s.data_ptr(), **kw_s)
        ctx.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free_before, "the call changed the amount of free device memory"
        # the existing entry point on the same inputs
        ctx.set_camera(big)
        rgb_d = torch.zeros_like(rgb_n)
        w_d = torch.zeros_like(w_n)
        ctx.render_dev(rgb_d.data_ptr(), w_d.data_ptr(), **kw)
        ctx.set_camera(small)
        rgb_ds = torch.zeros_like(rgb_s)
        w_ds = torch.zeros_like(w_s)
        ctx.render_dev(rgb_ds.data_ptr(), w_ds.data_ptr(), **kw_s)
        ctx.synchronize()
        assert float(w_d.min()) >= 6 and float(rgb_d.sum()) > 0
        assert torch.equal(w_n, w_d) and torch.equal(rgb_n, rgb_d), "first call (320x200) differs from rt_render_dev"
        assert torch.equal(w_s, w_ds) and torch.equal(rgb_s, rgb_ds), "second, smaller call (96x64) differs from rt_render_dev"

        # Whitted, and a pass size the sample count does not divide
        a, aw, b, bw = _pair(ctx, torch, small, integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_COUNTER, seed=3, spp=3, max_depth=6)
        assert np.array_equal(aw, bw) and np.array_equal(a, b)
        a, aw, b, bw = _pair(ctx, torch, small, integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=4, spp=7, max_depth=5, spp_per_pass=3)
        assert np.array_equal(aw, bw) and np.array_equal(a, b)

        # a scene whose tables live in HBM (streaming extend / shadow stages), mixed materials, depth 8
        scene, cfg = raygen.grid_scene(128, 64, path_scene.config, width=160, height=90)
        ctx.upload_scene(scene)
        ctx.set_camera(cfg)
        a, aw, b, bw = _pair(ctx, torch, cfg, integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=5, spp=4, max_depth=8)
        assert np.array_equal(aw, bw) and np.array_equal(a, b) and a.sum() > 0

        # beyond the limits: a clear error, nothing rendered
        ctx.upload_scene(path_scene)
        too_big = path_scene.config.with_resolution(640, 480, spp=1)   # 307 200 pixels > RT_NOALLOC_MAX_PATHS
        ctx.set_camera(too_big)
        rgb = torch.zeros((480, 640, 3), dtype=torch.float32, device=dev)
        w = torch.zeros((480, 640), dtype=torch.float32, device=dev)
        with pytest.raises(RtError, match="RT_NOALLOC_MAX_PATHS"):
            ctx.render_noalloc_dev(rgb.data_ptr(), w.data_ptr(), integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, spp=1, max_depth=3)
        ctx.set_camera(small)
        with pytest.raises(RtError, match="RT_RNG_COUNTER"):
            ctx.render_noalloc_dev(rgb_s.data_ptr(), w_s.data_ptr(), integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, spp=2, max_depth=3)
        with pytest.raises(RtError, match="RT_NOALLOC_MAX_DEPTH"):
            ctx.render_noalloc_dev(rgb_s.data_ptr(), w_s.data_ptr(), integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, spp=2, max_depth=65)
        ctx.synchronize()
        assert float(rgb.sum()) == 0.0
