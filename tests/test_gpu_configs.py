"""GPU parity on the two synthetic BASELINE configurations whose traversal tables live in HBM:
configs[3], the 1 M-triangle soup, and configs[4], the ~4 M-triangle connected mesh with Lambertian / Mirror / Glass
stripes at depth 8. The connected mesh is where the triangle test's ties decide the primitive id: the reference accepts
tScaled == tMax * det (Shape/TriangleShape.cpp:354-357), so of several triangles that a ray meets at the same t - a
shared edge, a shared vertex - the LAST one tested wins, and the test order is the order of the leaf lists. Random soups
never produce such ties. Bar as everywhere: primitive ids, t and barycentrics bit-equal to the CPU oracle; Path sums
bit-equal to the oracle with correctly rounded trig (up to reordered border sums); Whitted within 1e-4 per pixel."""
import numpy as np
import pytest

import oracle_lib
import raygen
from oracle_lib import Oracle
from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM

pytestmark = pytest.mark.gpu


def _mean(rgb, w):
    return rgb / np.maximum(w, 1)[..., None]


def _closest_both_schedules(ctx, orc, o, d, label, tmax=None):
    """ids / t / barycentrics bit-equal to the oracle under both ray-to-lane schedules; returns the oracle's ids."""
    cp, ct, cb = orc.trace_closest(o, d, tmax)
    try:
        for schedule in ("lockstep", "stream", "refill"):
            ctx.set_trace_schedule(schedule)
            gp, gt, gb = ctx.trace_closest(o, d, tmax)
            assert np.array_equal(gp, cp), f"{label}/{schedule}: {np.count_nonzero(gp != cp)} of {len(gp)} primitive ids differ"
            assert np.array_equal(gt.view(np.uint32), ct.view(np.uint32)), f"{label}/{schedule}: t not bit-equal"
            assert np.array_equal(gb.view(np.uint32), cb.view(np.uint32)), f"{label}/{schedule}: barycentrics not bit-equal"
    finally:
        ctx.set_trace_schedule("auto")
    return cp, cb


def _any_both_schedules(ctx, orc, o, d, tmax, label):
    c = orc.trace_any(o, d, tmax)
    try:
        for schedule in ("lockstep", "stream", "refill"):
            ctx.set_trace_schedule(schedule)
            assert np.array_equal(ctx.trace_any(o, d, tmax), c), f"{label}/{schedule}"
    finally:
        ctx.set_trace_schedule("auto")
    return c


def test_connected_mesh_ties_and_mixed_materials(gpu_ctx, path_scene):
    """A 128 x 64 x 2 = 16 384-triangle displaced grid with the material stripes of configs[4], depth 8.
    (a) closest / any-hit batches aimed at shared vertices and shared edges (mesh_feature_rays): in 4 of 10 hits one
        barycentric is exactly 1, i.e. up to six triangles are met at one t and the leaf order decides the id;
    (b) Path and Whitted renders against the oracle under both schedules."""
    scene, cfg = raygen.grid_scene(128, 64, path_scene.config, width=96, height=72)
    assert scene.n_prims == 128 * 64 * 2 + 2
    orc = Oracle(scene, cfg)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cfg)
    assert gpu_ctx.stats()["scene_in_lds"] == 0
    o, d = raygen.mesh_feature_rays(scene.tri_p[:-2], 300_000, seed=7)
    prim, bary = _closest_both_schedules(gpu_ctx, orc, o, d, "mesh features")
    hit = prim >= 0
    assert hit.mean() > 0.95
    assert (bary[hit] == 1).any(axis=1).mean() > 0.25, "vertex hits (ties between up to six triangles) must be exercised"
    assert (bary[hit] == 0).any(axis=1).mean() > 0.5, "edge hits (ties between two triangles) must be exercised"
    # the shadow flavour on the same features, and rays that start on the mesh (second generation)
    tmax = np.random.RandomState(8).uniform(0, 500, len(o)).astype(np.float32)
    _any_both_schedules(gpu_ctx, orc, o, d, tmax, "mesh features, any-hit")
    o2, d2 = raygen.surface_rays(orc, o, d, seed=9)
    _closest_both_schedules(gpu_ctx, orc, o2, d2, "from the mesh")
    oc, dc = raygen.camera_rays(orc, cfg, 200_000, seed=10)
    _closest_both_schedules(gpu_ctx, orc, oc, dc, "camera")

    oracle_lib.set_trig_mode(True)
    try:
        crgb, cw, st = orc.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=6, max_depth=8, spp=8)
    finally:
        oracle_lib.set_trig_mode(False)
    wrgb_c, ww_c, wst = orc.render(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, max_depth=8, spp=2)
    assert _mean(crgb, cw).mean() > 0.01 and st["rays_closest"] > 1.5 * 96 * 72 * 8
    assert wst["rays_closest"] > 1.15 * 96 * 72 * 2, "the specular recursion must have happened"
    images = {}
    try:
        for schedule in ("lockstep", "stream"):
            gpu_ctx.set_trace_schedule(schedule)
            grgb, gw = gpu_ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=6, max_depth=8, spp=8, spp_per_pass=3)
            assert np.array_equal(gw, cw)
            neq = (grgb != crgb).any(axis=2)
            assert neq.mean() <= 2e-3, f"{schedule}: {np.count_nonzero(neq)} of {neq.size} pixels differ, max {np.abs(grgb - crgb).max()}"
            images[schedule] = grgb
            wrgb_g, ww_g = gpu_ctx.render(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, max_depth=8, spp=2)
            assert np.array_equal(ww_g, ww_c)
            assert np.abs(_mean(wrgb_g, ww_g) - _mean(wrgb_c, ww_c)).max() <= 1e-4, schedule
    finally:
        gpu_ctx.set_trace_schedule("auto")
    assert np.array_equal(images["lockstep"], images["stream"])


def _full_size_frame_properties(ctx, cfg, depth):
    """1920x1080 x 2 spp where the oracle is too slow to be the checker: every pixel receives exactly its samples, all
    sums are finite, and two sample-index shards add up to the unsharded frame."""
    ctx.set_camera(cfg)
    ctx.reset_stats()
    full, fw = ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=4, spp=2, max_depth=depth)
    st = ctx.stats()
    assert st["camera_samples"] == 1920 * 1080 * 2
    assert fw.min() >= 2 and (fw != 2).mean() < 2e-3 and np.isfinite(full).all() and full.min() >= 0
    assert full.sum() > 0
    rgb = np.zeros_like(full)
    w = np.zeros_like(fw)
    ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=4, spp=2, spp_begin=0, spp_end=1, max_depth=depth, rgb_sum=rgb, weight_sum=w)
    ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=4, spp=2, spp_begin=1, spp_end=2, max_depth=depth, rgb_sum=rgb, weight_sum=w)
    assert np.array_equal(w, fw)
    neq = (rgb != full).any(axis=2)
    assert neq.mean() < 2e-3 and np.abs(rgb - full).max() <= 2e-6 * np.abs(full).max()
    return st


def test_config4_one_million_triangle_soup(gpu_ctx, path_scene):
    """BASELINE configs[3] at its stated size: 1 000 000 random triangles, KD-tree built by the host front-end with the
    reference's parameters (maxDepth = round(8 + 1.3 log2 N) = 34, Accelerators/KDTree.cpp:97)."""
    scene, cfg = raygen.soup_scene(1_000_000, path_scene.config)
    assert scene.tree_stats["max_depth_param"] == 34 and scene.n_nodes > 10_000_000
    orc = Oracle(scene, cfg)
    gpu_ctx.upload_scene(scene)
    oc, dc = raygen.camera_rays(orc, cfg, 200_000, seed=21)
    _closest_both_schedules(gpu_ctx, orc, oc, dc, "1M soup, camera")
    o, d = raygen.incoherent_rays(scene.bounds, 200_000, seed=22)
    _, _, _, cn = orc.trace_closest(o, d, counters=True)
    assert cn["max_stack"] > 12, "the batch must reach past the todo entries kept in LDS"
    prim, _ = _closest_both_schedules(gpu_ctx, orc, o, d, "1M soup, incoherent")
    assert 0.5 < (prim >= 0).mean() < 1.0
    _any_both_schedules(gpu_ctx, orc, o, d, np.full(len(o), 40.0, np.float32), "1M soup, any-hit")
    st = _full_size_frame_properties(gpu_ctx, cfg, depth=15)
    assert (st["rays_closest"] + st["rays_any"]) / st["camera_samples"] > 1.2


def test_config5_four_million_triangle_mesh(gpu_ctx, path_scene):
    """BASELINE configs[4] at its stated size: displaced grid of 2048 x 1024 x 2 = 4 194 304 connected triangles, 60 / 25 /
    15 % Lambertian / Mirror / Glass, depth 8 (maxDepth 37)."""
    scene, cfg = raygen.grid_scene(2048, 1024, path_scene.config)
    assert scene.n_prims == 4_194_304 + 2 and scene.tree_stats["max_depth_param"] == 37
    orc = Oracle(scene, cfg)
    gpu_ctx.upload_scene(scene)
    oc, dc = raygen.camera_rays(orc, cfg, 200_000, seed=31)
    prim, _ = _closest_both_schedules(gpu_ctx, orc, oc, dc, "4M mesh, camera")
    assert (prim >= 0).mean() > 0.3
    o, d = raygen.incoherent_rays(scene.bounds, 200_000, seed=32)
    _, _, _, cn = orc.trace_closest(o, d, counters=True)
    assert cn["max_stack"] > 12
    _closest_both_schedules(gpu_ctx, orc, o, d, "4M mesh, incoherent")
    of, df = raygen.mesh_feature_rays(scene.tri_p[:-2], 200_000, seed=33)
    prim, bary = _closest_both_schedules(gpu_ctx, orc, of, df, "4M mesh, shared vertices and edges")
    assert (bary[prim >= 0] == 1).any(axis=1).mean() > 0.25
    _any_both_schedules(gpu_ctx, orc, of, df, np.full(len(of), 450.0, np.float32), "4M mesh, any-hit")
    st = _full_size_frame_properties(gpu_ctx, cfg, depth=8)
    assert (st["rays_closest"] + st["rays_any"]) / st["camera_samples"] > 1.2
