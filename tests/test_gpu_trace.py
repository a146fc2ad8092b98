"""GPU parity of the traversal + intersection kernels against the CPU oracle, through the C ABI.
Bar: primitive ids bit-exact; t and barycentrics within 1e-5 relative (they are expected to be bit-equal,
which is asserted as well and reported separately)."""
import numpy as np
import pytest

import raygen
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["lockstep", "stream", "refill"])
def trace_schedule(request, gpu_ctx):
    """Every test of this module runs under both ray-to-lane schedules of the batch tracers."""
    gpu_ctx.set_trace_schedule(request.param)
    yield request.param
    gpu_ctx.set_trace_schedule("auto")


def _compare_closest(ctx, orc, o, d, tmax=None, label=""):
    gp, gt, gb = ctx.trace_closest(o, d, tmax)
    cp, ct, cb = orc.trace_closest(o, d, tmax)
    assert np.array_equal(gp, cp), f"{label}: {np.count_nonzero(gp != cp)} of {len(gp)} primitive ids differ"
    hit = cp >= 0
    np.testing.assert_allclose(gt[hit], ct[hit], rtol=1e-5, atol=0, err_msg=label)
    np.testing.assert_allclose(gb[hit], cb[hit], rtol=1e-5, atol=1e-7, err_msg=label)
    # stronger than the bar: identical bits
    assert np.array_equal(gt.view(np.uint32), ct.view(np.uint32)), f"{label}: t not bit-equal"
    assert np.array_equal(gb.view(np.uint32), cb.view(np.uint32)), f"{label}: barycentrics not bit-equal"
    return cp


def test_closest_camera_rays(gpu_ctx, path_scene):
    orc = Oracle(path_scene)
    gpu_ctx.upload_scene(path_scene)
    o, d = raygen.camera_rays(orc, path_scene.config, 500_000, seed=1)
    prim = _compare_closest(gpu_ctx, orc, o, d, label="camera")
    assert (prim >= 0).mean() > 0.9


def test_closest_incoherent_rays(gpu_ctx, path_scene):
    orc = Oracle(path_scene)
    gpu_ctx.upload_scene(path_scene)
    o, d = raygen.incoherent_rays(path_scene.bounds, 500_000, seed=2, slack=0.2)
    _compare_closest(gpu_ctx, orc, o, d, label="incoherent")


def test_closest_rays_from_surfaces(gpu_ctx, path_scene):
    """Rays that start exactly on triangles: the reference applies no origin offset, so the t <= deltaT guard
    and the zero-edge double fallback decide the result."""
    orc = Oracle(path_scene)
    gpu_ctx.upload_scene(path_scene)
    o, d = raygen.incoherent_rays(path_scene.bounds, 300_000, seed=3)
    o2, d2 = raygen.surface_rays(orc, o, d, seed=4)
    assert len(o2) > 100_000
    _compare_closest(gpu_ctx, orc, o2, d2, label="surface")


def test_closest_axis_parallel_and_tmax(gpu_ctx, path_scene):
    orc = Oracle(path_scene)
    gpu_ctx.upload_scene(path_scene)
    o, d = raygen.axis_rays(path_scene.bounds, 100_000, seed=5)
    _compare_closest(gpu_ctx, orc, o, d, label="axis")
    rs = np.random.RandomState(6)
    tmax = rs.uniform(0, 600, len(o)).astype(np.float32)
    _compare_closest(gpu_ctx, orc, o, d, tmax, label="axis+tmax")
    o, d = raygen.incoherent_rays(path_scene.bounds, 100_000, seed=7)
    tmax = rs.uniform(0, 400, len(o)).astype(np.float32)
    _compare_closest(gpu_ctx, orc, o, d, tmax, label="tmax")


def test_any_hit(gpu_ctx, path_scene):
    orc = Oracle(path_scene)
    gpu_ctx.upload_scene(path_scene)
    o, d = raygen.incoherent_rays(path_scene.bounds, 300_000, seed=8)
    rs = np.random.RandomState(9)
    for tmax in (None, rs.uniform(0, 300, len(o)).astype(np.float32), np.full(len(o), 1 - 1e-4, np.float32)):
        g = gpu_ctx.trace_any(o, d, tmax)
        c = orc.trace_any(o, d, tmax)
        assert np.array_equal(g, c)
    o2, d2 = raygen.surface_rays(orc, o, d, seed=10)
    assert np.array_equal(gpu_ctx.trace_any(o2, d2), orc.trace_any(o2, d2))


def test_empty_batch_and_all_miss(gpu_ctx, path_scene):
    gpu_ctx.upload_scene(path_scene)
    z = np.zeros((0, 3), np.float32)
    p, t, b = gpu_ctx.trace_closest(z, z)
    assert p.shape == (0,)
    o = np.full((1000, 3), 1e4, np.float32)
    d = np.tile(np.array([[0, 1, 0]], np.float32), (1000, 1))
    p, t, b = gpu_ctx.trace_closest(o, d)
    assert (p == -1).all() and np.isinf(t).all()
    assert (gpu_ctx.trace_any(o, d) == 0).all()


def test_soup_scene_global_memory_path(gpu_ctx):
    """A 200k-triangle soup does not fit in LDS: traversal tables are read from HBM and the todo stack
    spills past its LDS part."""
    from raytracing_amd.scene import build_scene, make_materials
    tri = raygen.triangle_soup(200_000, seed=11)
    scene = build_scene(tri, np.zeros(len(tri), np.int32), make_materials([(0, (0.73, 0.73, 0.73), 1.0)]))
    assert scene.n_nodes > 100_000
    orc = Oracle(scene)
    gpu_ctx.upload_scene(scene)
    o, d = raygen.incoherent_rays(scene.bounds, 200_000, seed=12)
    cp, ct, cb, cn = orc.trace_closest(o, d, counters=True)
    assert cn["max_stack"] > 12, "the batch must exercise the stack's overflow part (12 entries live in LDS)"
    _compare_closest(gpu_ctx, orc, o, d, label="soup")
    tmax = np.full(len(o), 30.0, np.float32)
    assert np.array_equal(gpu_ctx.trace_any(o, d, tmax), orc.trace_any(o, d, tmax))


def test_traversal_counters_match_oracle(gpu_ctx, path_scene):
    """The instrumented kernels count exactly the node visits and triangle tests the oracle performs."""
    orc = Oracle(path_scene)
    gpu_ctx.upload_scene(path_scene)
    o, d = raygen.camera_rays(orc, path_scene.config, 100_000, seed=13)
    gpu_ctx.reset_stats()
    gpu_ctx.trace_closest(o, d)
    st = gpu_ctx.stats()
    _, _, _, cn = orc.trace_closest(o, d, counters=True)
    assert st["counted_rays"] == len(o)
    assert st["n_interior"] == cn["n_interior"] and st["n_leaf"] == cn["n_leaf"]
    assert st["n_tri_tests"] == cn["n_tri_tests"] and st["n_leaf_idx"] == cn["n_leaf_idx"]


def test_sphere_scene_closest_and_any(gpu_ctx, sphere_scene):
    """cornell.json: 38 triangles + a mirror sphere + an emissive sphere (ASphereShape::hit, both flavours).
    Camera rays, rays from everywhere and rays that start ON surfaces (sphere surfaces included: a ray leaving the
    mirror sphere sees its own sphere at t ~ 0, which only the t0 <= 0 branch of the quadratic keeps out).
    For a sphere hit the batch interface reports (t, 0, 0) in place of barycentrics."""
    orc = Oracle(sphere_scene)
    gpu_ctx.upload_scene(sphere_scene)
    o, d = raygen.camera_rays(orc, sphere_scene.config, 300_000, seed=11)
    prim = _compare_closest(gpu_ctx, orc, o, d, label="sphere scene, camera")
    n_sphere = np.isin(prim, sphere_scene.spheres["prim_id"]).sum()
    assert n_sphere > 3000, "the camera must see the spheres"
    o, d = raygen.incoherent_rays(sphere_scene.bounds, 300_000, seed=12)
    prim = _compare_closest(gpu_ctx, orc, o, d, label="sphere scene, incoherent")
    assert np.isin(prim, sphere_scene.spheres["prim_id"]).sum() > 3000
    # second generation: start on whatever was hit (oracle's own hit points), new random directions
    _, t, _ = orc.trace_closest(o, d)
    keep = prim >= 0
    p = (o[keep] + d[keep] * t[keep, None]).astype(np.float32)
    _, d2 = raygen.incoherent_rays((0, 0, 0, 1, 1, 1), int(keep.sum()), seed=13)
    _compare_closest(gpu_ctx, orc, p, d2, label="sphere scene, from surfaces")
    tmax = np.random.RandomState(14).uniform(0, 800, len(p)).astype(np.float32)
    assert np.array_equal(gpu_ctx.trace_any(p, d2, tmax), orc.trace_any(p, d2, tmax))
    assert np.array_equal(gpu_ctx.trace_any(o, d), orc.trace_any(o, d))


def test_upload_rejects_a_tree_deeper_than_the_todo_stack(gpu_ctx, path_scene):
    """The traversal keeps 64 todo entries like the reference's todo[64] (Accelerators/KDTree.cpp:396-398); a caller's
    tree that could need more is refused at upload instead of overrunning the stack on the device."""
    import copy
    from raytracing_amd.device import RtError

    def chain(n_interior):
        # interior node i: split plane x = 1 + i, below child i + 1, above child = the shared empty leaf at the end
        nodes = np.zeros((n_interior + 2, 2), np.uint32)
        for i in range(n_interior):
            nodes[i, 0] = np.float32(1 + i).view(np.uint32)
            nodes[i, 1] = (n_interior + 1) << 2
        nodes[n_interior] = (0, 3)        # empty leaves
        nodes[n_interior + 1] = (0, 3)
        s = copy.copy(path_scene)
        s.nodes = nodes
        s.leaf_idx = np.zeros(0, np.int32)
        return s

    gpu_ctx.upload_scene(chain(64))     # deepest interior node at depth 63: at most 64 entries
    with pytest.raises(RtError, match="deeper"):
        gpu_ctx.upload_scene(chain(65))
    gpu_ctx.upload_scene(path_scene)
