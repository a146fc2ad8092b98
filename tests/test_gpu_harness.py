"""The C ABI driven by a compiled C program (tests/abi_harness.c, built by `make harness`) with no Python in the
process - the nearest thing to the reference-side Renderer binding that can be built without the reference. Its outputs
must equal what the ctypes path returns for the same inputs, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import flatbin
import raygen
from conftest import ROOT
from oracle_lib import Oracle
from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_RNG_COUNTER

pytestmark = pytest.mark.gpu
HARNESS = os.path.join(ROOT, "tests", "bin", "abi_harness")


@pytest.mark.parametrize("devices", ["0", "0,0"])
def test_c_caller_matches_the_ctypes_path(gpu_ctx, path_scene, tmp_path, devices):
    if not os.path.exists(HARNESS):
        pytest.fail(f"{HARNESS} is missing: build it with `make harness`")
    cfg = path_scene.config.with_resolution(128, 96, spp=6)
    orc = Oracle(path_scene, cfg)
    o, d = raygen.camera_rays(orc, cfg, 50_000, seed=41)
    o2, d2 = raygen.incoherent_rays(path_scene.bounds, 50_000, seed=42)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    scene_bin, rays_bin, out_bin = (str(tmp_path / n) for n in ("scene.bin", "rays.bin", "out.bin"))
    flatbin.write_scene(scene_bin, path_scene, cfg, RT_INTEGRATOR_PATH, 6, 15, RT_RNG_COUNTER, seed=17)
    flatbin.write_rays(rays_bin, o, d)
    run = subprocess.run([HARNESS, scene_bin, rays_bin, out_bin, devices], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    n_dev = devices.count(",") + 1
    assert f"{n_dev} device(s)" in run.stdout and f"{128 * 96 * 6} camera samples" in run.stdout
    prim, t, bary, occ, rgb, w, rgb8 = flatbin.read_outputs(out_bin, len(o), cfg.width, cfg.height)

    gpu_ctx.upload_scene(path_scene)
    gpu_ctx.set_camera(cfg)
    gp, gt, gb = gpu_ctx.trace_closest(o, d)
    assert np.array_equal(prim, gp) and np.array_equal(t.view(np.uint32), gt.view(np.uint32)) and np.array_equal(bary.view(np.uint32), gb.view(np.uint32))
    assert np.array_equal(occ, gpu_ctx.trace_any(o, d))
    cp, _, _ = orc.trace_closest(o, d)
    assert np.array_equal(prim, cp)
    grgb, gw = gpu_ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=17, spp=6, max_depth=15)
    assert np.array_equal(w, gw)
    if n_dev == 1:
        assert np.array_equal(rgb, grgb)
        assert np.array_equal(rgb8, gpu_ctx.develop(grgb, gw))
    else:   # two members: the same samples summed in another order
        assert np.abs(rgb - grgb).max() <= 4e-6 * np.abs(grgb).max()
        assert np.abs(rgb8.astype(np.int16) - gpu_ctx.develop(grgb, gw).astype(np.int16)).max() <= 1
