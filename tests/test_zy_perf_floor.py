"""Wall-clock floor of the benchmark workload. Opt-in (RT_PERF_TESTS=1): a timing assertion says nothing about parity
and fails on a shared or throttled box with no defect in the code, so it stays out of the default GPU run."""
import os
import time

import pytest

from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_RNG_COUNTER

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("RT_PERF_TESTS") != "1", reason="set RT_PERF_TESTS=1 to run")]


def test_throughput_floor(gpu_ctx, path_scene):
    """cornellBox PathTracing 1920x1080 x 16 spp through the host-buffer entry point (the time includes 2 x 33 MB over
    PCIe) must stay above a floor far below what an MI355X measures (10 G rays/s)."""
    cfg = path_scene.config.with_resolution(1920, 1080, spp=16)
    gpu_ctx.upload_scene(path_scene)
    gpu_ctx.set_camera(cfg)
    kw = dict(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=1, spp=16, max_depth=15)
    rgb, w = gpu_ctx.render(**kw)                       # warm-up: allocates the queues
    gpu_ctx.reset_stats()
    t0 = time.perf_counter()
    gpu_ctx.render(rgb_sum=rgb, weight_sum=w, **kw)
    dt = time.perf_counter() - t0
    st = gpu_ctx.stats()
    mrays = (st["rays_closest"] + st["rays_any"]) / dt / 1e6
    assert float(w.min()) >= 32.0
    assert mrays > 2500, f"{mrays:.0f} Mrays/s"
