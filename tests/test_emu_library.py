"""The whole device library - C ABI, host code and kernels of raytracing_amd/csrc/device/rt_device.hip, unchanged - compiled
for the CPU (tests/emu/emu_library.cpp) and called through the ABI under AddressSanitizer with leak detection. "Device"
memory is the heap at exact sizes, streams run in order, every launch runs its grid as fibres that meet at ballots and
barriers. The tests below are small versions of the GPU parity tests and compare with the oracle to the same bars: ids /
t / barycentrics / occlusion bit-equal, Path film bit-equal (correctly rounded trig on both sides) up to reordered sums
where several passes or members meet, Whitted within 1e-4 per pixel. One server process serves the whole module, so what
a call leaves behind (buffers, counters, options) meets the next call, as in a host program. The module runs twice: with
streams that run their work as it is queued, and with streams that defer it until something waits for it.
The library has no CPU path; this is a checker of its source (memory safety of host and device code, barrier discipline,
buffer lifetimes across calls), run where no GPU is needed."""
import copy
import os
from dataclasses import replace

import numpy as np
import pytest

import oracle_lib
import raygen
from emu_context import EmuContext, EmuError, build_server
from oracle_lib import Oracle
from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM
from raytracing_amd.scene import make_materials

W, H = 48, 36


def _mean(rgb, w):
    return rgb / np.maximum(w, 1)[..., None]


def _oracle(orc, **kw):
    """The oracle with cos / sin / acos rounded once from double, which is what the kernels compute."""
    oracle_lib.set_trig_mode(True)
    try:
        return orc.render(**kw)
    finally:
        oracle_lib.set_trig_mode(False)


def _oracle_path(orc, **kw):
    return _oracle(orc, integrator=RT_INTEGRATOR_PATH, **kw)


@pytest.fixture(scope="module", params=["eager", "lazy"])
def emu(request):
    """eager: queued work runs at once. lazy: it runs when something waits for it, which reorders the work of the context's
    streams as far as the library's events allow - a missing event or synchronize shows as stale data."""
    if build_server() is None:
        pytest.skip("no g++")
    ctx = EmuContext(lazy_streams=request.param == "lazy")
    yield ctx
    ctx.close()   # raises on a sanitizer report, leaks included


def _check_traces(ctx, orc, o, d, tmax_any, label):
    cp, ct, cb = orc.trace_closest(o, d)
    ca = orc.trace_any(o, d, tmax_any)
    try:
        for schedule in ("lockstep", "stream"):
            ctx.set_trace_schedule(schedule)
            gp, gt, gb = ctx.trace_closest(o, d)
            assert np.array_equal(gp, cp), f"{label}/{schedule}: {np.count_nonzero(gp != cp)} of {len(o)} primitive ids differ"
            assert np.array_equal(gt.view(np.uint32), ct.view(np.uint32)), f"{label}/{schedule}"
            assert np.array_equal(gb.view(np.uint32), cb.view(np.uint32)), f"{label}/{schedule}"
            assert np.array_equal(ctx.trace_any(o, d, tmax_any), ca), f"{label}/{schedule}"
    finally:
        ctx.set_trace_schedule("auto")
    return cp, ca


def _check_path(got, want, label, exact=True):
    (rgb, w), (crgb, cw) = got, want[:2]
    assert np.array_equal(w, cw), label
    assert np.abs(rgb - crgb).max() <= 4e-6 * np.abs(crgb).max(), label
    if exact:
        neq = (rgb != crgb).any(axis=2)
        assert neq.mean() <= 2e-3, f"{label}: {np.count_nonzero(neq)} pixels differ"


@pytest.mark.parametrize("scene_name", ["cornell", "mesh"])
def test_session_one_device(emu, path_scene, scene_name):
    """Upload, both schedules, Path, Whitted on tile streams (allocates its offset search), Path again, second upload,
    the entry point that allocates nothing, development - on the cornellBox (tables in LDS on a GPU) and on a 32 x 16 x 2
    mesh with Lambertian / mirror / glass stripes (tables in HBM), rays at shared vertices and edges included."""
    if scene_name == "cornell":
        scene, cfg = path_scene, path_scene.config.with_resolution(W, H)
    else:
        scene, cfg = raygen.grid_scene(32, 16, path_scene.config, width=W, height=H)
    orc = Oracle(scene, cfg)
    oc, dc = raygen.camera_rays(orc, cfg, 700, seed=1)
    oi, di = raygen.incoherent_rays(scene.bounds, 500, seed=3)
    o, d = np.concatenate([oc, oi]), np.concatenate([dc, di])
    if scene_name == "mesh":
        of, df = raygen.mesh_feature_rays(scene.tri_p[:-2], 600, seed=5)
        o, d = np.concatenate([o, of]), np.concatenate([d, df])
    tmax_any = np.random.RandomState(2).uniform(0, 400, len(o)).astype(np.float32)
    emu.upload_scene(scene)
    emu.set_camera(cfg)
    assert emu.stats()["scene_in_lds"] == (1 if scene_name == "cornell" else 0)
    prim, occ = _check_traces(emu, orc, o, d, tmax_any, scene_name)
    assert 0.3 < (prim >= 0).mean() and 0.05 < occ.mean() < 0.98
    kw = dict(rng_mode=RT_RNG_COUNTER, seed=3, max_depth=5, spp=4)
    want = _oracle_path(orc, **kw)
    assert _mean(*want[:2]).mean() > 0.005
    first = emu.render(integrator=RT_INTEGRATOR_PATH, **kw)
    _check_path(first, want, "path")
    wk = dict(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, max_depth=5, spp=2)
    wrgb, ww = emu.render(**wk)
    crgb, cw, _ = orc.render(**wk)
    assert np.array_equal(ww, cw) and np.abs(_mean(wrgb, ww) - _mean(crgb, cw)).max() <= 1e-4
    again = emu.render(integrator=RT_INTEGRATOR_PATH, **kw)
    assert np.array_equal(again[0], first[0]) and np.array_equal(again[1], first[1]), "the same job after a Whitted job"
    emu.upload_scene(scene)
    emu.set_camera(cfg)
    gp, gt, gb = emu.trace_closest(o, d)
    cp, ct, cb = orc.trace_closest(o, d)
    assert np.array_equal(gp, cp) and np.array_equal(gt.view(np.uint32), ct.view(np.uint32)) and np.array_equal(gb.view(np.uint32), cb.view(np.uint32))
    _check_path(emu.render_noalloc(integrator=RT_INTEGRATOR_PATH, **kw), want, "rt_render_noalloc_dev")
    # accumulation into the caller's film, in shards and in passes of 3 samples over the pipelines
    rgb, w = first[0].copy(), first[1].copy()
    for b, e in ((0, 1), (1, 4)):
        rgb, w = emu.render(integrator=RT_INTEGRATOR_PATH, spp_begin=b, spp_end=e, spp_per_pass=3, rgb_sum=rgb, weight_sum=w, **kw)
    assert np.array_equal(w, 2 * want[1]) and np.abs(rgb - 2 * want[0]).max() <= 1e-5 * np.abs(want[0]).max()
    rgb8 = emu.develop(*first)
    assert rgb8.shape == (H, W, 3) and rgb8.max() > 40


def test_a_device_list_that_repeats_some_devices_is_refused():
    """(1, 0, 1) names a device twice among three members: neither RCCL's case (distinct devices) nor the rehearsal's (one
    device throughout), where device-local sums would read another GPU's memory. rt_create_multi says so."""
    if build_server() is None:
        pytest.skip("no g++")
    with pytest.raises(EmuError, match="distinct"):
        EmuContext(device_ids=(1, 0, 1))


@pytest.mark.parametrize("device_ids", [(0, 0), (0, 1), (1, 1, 1)])
def test_group_of_members(device_ids):
    """rt_create_multi. One device listed two or three times: members' films summed by a kernel on that device. Distinct
    devices: the RCCL branch - communicator from ncclCommInitAll, one ncclReduce per member inside a group, on the member's
    stream - against tests/emu/rccl/rccl.h, which gives the six entry points the library binds the semantics of a sum-reduce
    to rank 0. A tile-stream Whitted job (member 0 alone, allocates) sits between two group jobs."""
    from raytracing_amd.scene import FlatScene
    if build_server() is None:
        pytest.skip("no g++")
    base = FlatScene.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornellBox_path.npz"))
    cfg = base.config.with_resolution(W, H)
    orc = Oracle(base, cfg)
    kw = dict(rng_mode=RT_RNG_COUNTER, seed=3, max_depth=5, spp=5)
    want = _oracle_path(orc, **kw)
    with EmuContext(device_ids=device_ids, lazy_streams=True) as grp:
        grp.upload_scene(base)
        grp.set_camera(cfg)
        grp.reset_stats()
        first = grp.render(integrator=RT_INTEGRATOR_PATH, **kw)
        assert grp.stats()["camera_samples"] == W * H * 5, "every sample once across the members"
        _check_path(first, want, "group", exact=False)   # the members' films are added: the same samples in another association
        wk = dict(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, max_depth=4, spp=2)
        wrgb, ww = grp.render(**wk)
        crgb, cw, _ = orc.render(**wk)
        assert np.array_equal(ww, cw) and np.abs(_mean(wrgb, ww) - _mean(crgb, cw)).max() <= 1e-4
        again = grp.render(integrator=RT_INTEGRATOR_PATH, **kw)
        assert np.array_equal(again[0], first[0]) and np.array_equal(again[1], first[1])
        o, d = raygen.camera_rays(orc, cfg, 300, seed=4)
        gp, _, _ = grp.trace_closest(o, d)
        assert np.array_equal(gp, orc.trace_closest(o, d)[0])


def test_spheres_and_specular_materials(emu, path_scene, sphere_scene):
    """The sphere flavours of every kernel (cornell.json: mirror sphere, emissive sphere) and mirror / glass boxes, both
    integrators, counter streams and the reference's tile streams."""
    mixed = copy.copy(path_scene)
    mats = mixed.materials.copy()
    mats[6] = make_materials([(2, (0.9, 0.9, 0.9), 1.5)])[0]    # short box: glass
    mats[7] = make_materials([(1, (0.8, 0.8, 0.8), 1.0)])[0]    # tall box: mirror
    mixed.materials = mats
    for label, scene in (("spheres", sphere_scene), ("mixed", mixed)):
        cfg = scene.config.with_resolution(40, 40, spp=3)
        orc = Oracle(scene, cfg)
        emu.upload_scene(scene)
        emu.set_camera(cfg)
        oc, dc = raygen.camera_rays(orc, cfg, 600, seed=11)
        os_, ds = raygen.surface_rays(orc, oc, dc, seed=12)
        o, d = np.concatenate([oc, os_]), np.concatenate([dc, ds])
        _check_traces(emu, orc, o, d, np.full(len(o), 250.0, np.float32), label)
        for rng_mode in (RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM):
            kw = dict(rng_mode=rng_mode, seed=5, max_depth=6)
            emu.reset_stats()
            got = emu.render(integrator=RT_INTEGRATOR_PATH, **kw)
            st = emu.stats()
            want = _oracle_path(orc, **kw)
            _check_path(got, want, f"{label} path {rng_mode}")
            if rng_mode == RT_RNG_REFERENCE_TILE_STREAM:
                assert st["rays_closest"] == want[2]["rays_closest"] and st["rays_any"] == want[2]["rays_any"]
            wrgb, ww = emu.render(integrator=RT_INTEGRATOR_WHITTED, **kw)
            crgb, cw, _ = _oracle(orc, integrator=RT_INTEGRATOR_WHITTED, **kw)
            assert np.array_equal(ww, cw) and np.abs(_mean(wrgb, ww) - _mean(crgb, cw)).max() <= 1e-4, f"{label} whitted {rng_mode}"


def test_film_window_wide_filter_and_clamp(emu, path_scene):
    """rt_set_film: cropped pixel bounds, a Box filter wider than a pixel (neighbours' samples arrive through float atomics
    and the spill buffers) and MaxLum, both integrators; then the full window, whose sample bounds start at negative
    film coordinates."""
    base = path_scene.config.with_resolution(40, 28, spp=3)
    cfg = replace(base, cropped_bounds=(4, 3, 35, 25)).with_filter((1.0, 1.5), max_sample_luminance=1.5)
    emu.upload_scene(path_scene)
    for c in (cfg, base.with_filter((1.0, 1.5))):
        orc = Oracle(path_scene, c)
        emu.set_camera(c)
        for integrator in (RT_INTEGRATOR_WHITTED, RT_INTEGRATOR_PATH):
            kw = dict(rng_mode=RT_RNG_COUNTER, seed=9, max_depth=4)
            grgb, gw = emu.render(integrator=integrator, spp_per_pass=2, **kw)
            oracle_lib.set_trig_mode(True)
            try:
                crgb, cw, _ = orc.render(integrator=integrator, **kw)
            finally:
                oracle_lib.set_trig_mode(False)
            assert np.array_equal(gw, cw)
            np.testing.assert_allclose(grgb, crgb, rtol=2e-5, atol=1e-6)
    assert not gw[:0].any() and gw.min() > 0
    with pytest.raises(EmuError, match="rt error"):
        emu._call(4, __import__("struct").pack("<2f4if", 0.5, 0.5, 0, 0, 41, 28, float("inf")))   # cropped bounds outside the film


def test_traversal_counters_and_ray_counts(emu, path_scene):
    """The instrumented kernel flavours (counters in every traversal) against the oracle's counts."""
    cfg = path_scene.config.with_resolution(W, H, spp=2)
    orc = Oracle(path_scene, cfg)
    emu.upload_scene(path_scene)
    emu.set_camera(cfg)
    o, d = raygen.camera_rays(orc, cfg, 900, seed=13)
    emu.reset_stats()
    emu.trace_closest(o, d)
    st = emu.stats()
    _, _, _, cn = orc.trace_closest(o, d, counters=True)
    assert st["counted_rays"] == len(o)
    assert st["n_interior"] == cn["n_interior"] and st["n_leaf"] == cn["n_leaf"]
    assert st["n_tri_tests"] == cn["n_tri_tests"] and st["n_leaf_idx"] == cn["n_leaf_idx"]
    emu.reset_stats()
    got = emu.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=3, max_depth=6, collect_stats=True)
    st = emu.stats()
    want = _oracle_path(orc, rng_mode=RT_RNG_COUNTER, seed=3, max_depth=6, collect_stats=True)
    _check_path(got, want, "path with counters")
    assert st["camera_samples"] == want[2]["camera_samples"] == W * H * 2
    assert st["rays_closest"] == want[2]["rays_closest"] and st["rays_any"] == want[2]["rays_any"]
    assert st["n_tri_tests"] == want[2]["n_tri_tests"] and st["n_interior"] == want[2]["n_interior"]


def test_todo_stack_at_its_limit_and_rejected_input(emu, path_scene):
    """A chain of 64 interior nodes whose planes a ray meets far to near fills the todo stack to its 64th entry
    (Accelerators/KDTree.cpp:396-398: todo[64]); one node more is refused at upload. Bad arguments come back as errors."""
    def chain(n_interior):
        # interior node i: plane x = 500 - i, below child i + 1, above child = the empty leaf at the end
        nodes = np.zeros((n_interior + 2, 2), np.uint32)
        for i in range(n_interior):
            nodes[i, 0] = np.float32(500 - i).view(np.uint32)
            nodes[i, 1] = (n_interior + 1) << 2
        nodes[n_interior] = (0, 3)        # empty leaves
        nodes[n_interior + 1] = (0, 3)
        s = copy.copy(path_scene)
        s.nodes = nodes
        s.leaf_idx = np.zeros(0, np.int32)
        return s

    deep = chain(64)
    emu.upload_scene(deep)
    orc = Oracle(deep, deep.config)
    n = 200
    rs = np.random.RandomState(3)
    o = np.stack([rs.uniform(1, 50, n), rs.uniform(10, 500, n), rs.uniform(10, 500, n)], axis=1).astype(np.float32)
    d = np.stack([np.ones(n), rs.uniform(-0.05, 0.05, n), rs.uniform(-0.05, 0.05, n)], axis=1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    _, _, _, cn = orc.trace_closest(o, d, counters=True)
    assert cn["max_stack"] == 64
    _check_traces(emu, orc, o, d, np.full(n, 1000.0, np.float32), "64-entry chain")
    with pytest.raises(EmuError, match="deeper"):
        emu.upload_scene(chain(65))
    emu.upload_scene(path_scene)
    with pytest.raises(EmuError, match="rt error"):
        emu.set_option(1, 7)
    with pytest.raises(EmuError, match="rt error"):
        emu.render(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_COUNTER, spp=2, max_depth=33)
    empty = np.zeros((0, 3), np.float32)
    gp, gt, gb = emu.trace_closest(empty, empty)
    assert len(gp) == 0 and len(emu.trace_any(empty, empty)) == 0


def test_texture_coordinates_and_edge_configurations(emu, path_scene):
    """Per-corner texture coordinates (they set dpdu, hence the frame BSDF samples are drawn in; one triangle with degenerate
    coordinates), a scene without lights, depth 0, one sample per pixel and pass sizes that do not divide the sample count."""
    rs = np.random.RandomState(11)
    uv = rs.uniform(0, 1, (path_scene.n_prims, 6)).astype(np.float32)
    uv[3] = [0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
    textured = replace(path_scene, tri_uv=uv)
    cfg = path_scene.config.with_resolution(40, 30, spp=5)
    emu.upload_scene(textured)
    emu.set_camera(cfg)
    kw = dict(rng_mode=RT_RNG_COUNTER, seed=4, max_depth=5)
    got = emu.render(integrator=RT_INTEGRATOR_PATH, **kw)
    _check_path(got, _oracle_path(Oracle(textured, cfg), **kw), "texture coordinates")
    dark = replace(path_scene, light_id=np.full(path_scene.n_prims, -1, np.int32), lights=path_scene.lights[:0])
    emu.upload_scene(dark)
    emu.set_camera(cfg)
    for integrator in (RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED):
        rgb, w = emu.render(integrator=integrator, rng_mode=RT_RNG_COUNTER, max_depth=5)
        assert not rgb.any() and (w == 5).all()
    orc = Oracle(path_scene, cfg)
    emu.upload_scene(path_scene)
    emu.set_camera(cfg)
    assert (got[0] != emu.render(integrator=RT_INTEGRATOR_PATH, **kw)[0]).any(axis=2).mean() > 0.3, "the coordinates matter"
    for depth, spp in ((0, 5), (1, 1), (7, 5)):
        want = _oracle_path(orc, rng_mode=RT_RNG_COUNTER, seed=3, max_depth=depth, spp=spp)
        for per_pass in (0, 2, 3):
            _check_path(emu.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=3, max_depth=depth, spp=spp, spp_per_pass=per_pass),
                        want, f"depth {depth}, {spp} spp, passes of {per_pass}")


def test_scene_tables_no_builder_would_produce(emu, path_scene):
    """rt_upload_scene promises to validate everything a kernel will index with. Random node arrays that pass it - children
    shared between parents, planes outside the bounds, infinite and NaN planes, empty and overlapping leaf ranges, leaves of
    dozens of primitives - are traced under both schedules and rendered by both integrators: whatever the images show, no
    access may leave a table or the stack and every launch must end. Arrays that break a rule must come back as errors."""
    rs = np.random.RandomState(20)
    n_prims = path_scene.n_prims
    cfg = path_scene.config.with_resolution(24, 18, spp=1)
    accepted = rejected = 0
    for trial in range(40):
        n_nodes = int(rs.randint(1, 120))
        n_leaf_idx = int(rs.randint(0, 200))
        leaf_idx = rs.randint(0, n_prims, n_leaf_idx).astype(np.int32)
        nodes = np.zeros((n_nodes, 2), np.uint32)
        breaks_a_rule = trial % 4 == 3
        for i in range(n_nodes):
            interior = i + 2 < n_nodes and rs.rand() < 0.6
            if interior:
                split = np.float32(rs.choice([rs.uniform(-100, 700), np.inf, -np.inf, np.nan, 0.0], p=[0.8, 0.05, 0.05, 0.05, 0.05]))
                above = int(rs.randint(i + 1, n_nodes))
                nodes[i] = (split.view(np.uint32), int(rs.randint(0, 3)) | (above << 2))
            else:
                count = int(rs.choice([0, 1, 1, 2, 5, 40]))
                if count == 1:
                    nodes[i] = (int(rs.randint(0, n_prims)), 3 | (1 << 2))
                elif count > 1 and n_leaf_idx >= count:
                    nodes[i] = (int(rs.randint(0, n_leaf_idx - count + 1)), 3 | (count << 2))
                else:
                    nodes[i] = (0, 3)
        if breaks_a_rule:
            kind = trial % 3
            if kind == 0:
                nodes[-1] = (n_prims + int(rs.randint(0, 5)), 3 | (1 << 2))               # primitive index past the list
            elif kind == 1:
                nodes[-1] = (max(0, n_leaf_idx - 1), 3 | (3 << 2))                         # index range past the list
            else:
                nodes[0] = (0, 0 | (0 << 2)) if n_nodes > 2 else (n_prims, 3 | (1 << 2))   # interior node naming itself
        s = copy.copy(path_scene)
        s.nodes, s.leaf_idx = nodes, leaf_idx
        try:
            emu.upload_scene(s)
        except EmuError as e:
            assert "rt error" in str(e), e
            rejected += 1
            continue
        assert not breaks_a_rule, f"trial {trial} should have been refused"
        accepted += 1
        emu.set_camera(cfg)
        o, d = raygen.incoherent_rays(path_scene.bounds, 96, seed=trial)
        for schedule in ("lockstep", "stream"):
            emu.set_trace_schedule(schedule)
            prim, t, _ = emu.trace_closest(o, d)
            assert ((prim >= -1) & (prim < n_prims)).all()
            emu.trace_any(o, d, np.full(len(o), 300.0, np.float32))
        emu.set_trace_schedule("auto")
        for integrator, rng_mode in ((RT_INTEGRATOR_PATH, RT_RNG_COUNTER), (RT_INTEGRATOR_WHITTED, RT_RNG_REFERENCE_TILE_STREAM)):
            rgb, w = emu.render(integrator=integrator, rng_mode=rng_mode, max_depth=3)
            assert (w == 1).all()
    assert accepted >= 25 and rejected >= 8
    emu.upload_scene(path_scene)


def test_calls_out_of_order_and_bad_arguments(path_scene):
    """A fresh context: work before a scene or a camera exists, sizes and ranges that make no sense, more members than the
    device list can name - errors, never a crash, and a clean teardown afterwards."""
    if build_server() is None:
        pytest.skip("no g++")
    cfg = path_scene.config.with_resolution(16, 12, spp=2)
    o, d = raygen.incoherent_rays(path_scene.bounds, 8, seed=1)
    with pytest.raises(EmuError, match="rt error"):
        EmuContext(device_ids=(0, 2))          # the stand-in runtime has two devices
    with pytest.raises(EmuError, match="rt error"):
        EmuContext(device_ids=())
    with EmuContext(lazy_streams=True) as ctx:
        with pytest.raises(EmuError, match="rt error"):
            ctx.trace_closest(o, d)            # no scene
        ctx.config = cfg
        with pytest.raises(EmuError):
            ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER)   # no scene, no camera
        ctx.upload_scene(replace(path_scene, config=None))
        with pytest.raises(EmuError, match="rt error"):
            ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, spp=1, max_depth=2)   # no camera
        ctx.set_camera(cfg)
        for bad in (dict(spp=0), dict(spp=2, spp_begin=2, spp_end=1), dict(spp=2, spp_end=3), dict(spp=2, max_depth=-1), dict(spp=2, rng_mode=7),
                    dict(spp=2, integrator=5)):
            kw = dict(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, max_depth=3)
            kw.update(bad)
            with pytest.raises(EmuError, match="rt error"):
                ctx.render(**kw)
        with pytest.raises(EmuError, match="rt error"):
            ctx.render_noalloc(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, spp=2, max_depth=3)   # counter streams only
        with pytest.raises(EmuError, match="rt error"):
            ctx.set_pipelines(9)
        rgb, w = ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, max_depth=3)   # and it still works
        assert (w == 2).all() and rgb.max() > 0
        ctx.set_camera(path_scene.config.with_resolution(600, 500, spp=1))   # 300 000 pixels: past the scratch array's 262 144
        with pytest.raises(EmuError, match="rt error"):
            ctx.render_noalloc(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, max_depth=3)
        ctx.set_camera(cfg)
        with pytest.raises(EmuError, match="rt error"):
            ctx.render_noalloc(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, max_depth=65)
        rgb2, w2 = ctx.render_noalloc(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, max_depth=3)
        assert np.array_equal(rgb2, rgb) and np.array_equal(w2, w)


@pytest.mark.slow
@pytest.mark.parametrize("t_min", [1, 12])
def test_merged_round_build_of_the_streaming_tracer(path_scene, t_min):
    """The library built with RT_STREAM_MERGED=1 (dstream.h: every round serves node steps, leaf indices and primitive tests
    behind one wait; off in the shipped build until it has been measured on a GPU): batch traces and the wavefront path
    tracer's extend / shadow stages on a scene whose tables live in HBM, against the oracle. RT_SLOW=1 (another build)."""
    if build_server() is None:
        pytest.skip("no g++")
    scene, cfg = raygen.grid_scene(32, 16, path_scene.config, width=W, height=H)
    orc = Oracle(scene, cfg)
    of, df = raygen.mesh_feature_rays(scene.tri_p[:-2], 900, seed=5)
    oi, di = raygen.incoherent_rays(scene.bounds, 500, seed=3)
    o, d = np.concatenate([of, oi]), np.concatenate([df, di])
    with EmuContext(lazy_streams=True, defines=("RT_STREAM_MERGED=1", f"RT_MERGED_T_MIN={t_min}"), suffix=f"_merged{t_min}") as ctx:
        ctx.upload_scene(scene)
        ctx.set_camera(cfg)
        assert ctx.stats()["scene_in_lds"] == 0
        _check_traces(ctx, orc, o, d, np.random.RandomState(2).uniform(0, 400, len(o)).astype(np.float32), "merged rounds")
        kw = dict(rng_mode=RT_RNG_COUNTER, seed=3, max_depth=6, spp=4)
        ctx.reset_stats()
        got = ctx.render(integrator=RT_INTEGRATOR_PATH, collect_stats=True, spp_per_pass=3, **kw)
        st = ctx.stats()
        want = _oracle_path(orc, collect_stats=True, **kw)
        _check_path(got, want, "path, merged rounds")
        assert st["rays_closest"] == want[2]["rays_closest"] and st["rays_any"] == want[2]["rays_any"]
        assert st["n_tri_tests"] == want[2]["n_tri_tests"] and st["n_interior"] == want[2]["n_interior"] and st["n_leaf_idx"] == want[2]["n_leaf_idx"]


def test_a_failing_collective_is_an_error_not_a_crash(path_scene):
    """ncclReduce returns an error (EMU_RCCL_REDUCE_FAILS makes the stand-in do so): rt_render reports it, the context can
    still be used for work that needs no collective, and teardown is clean."""
    if build_server() is None:
        pytest.skip("no g++")
    cfg = path_scene.config.with_resolution(24, 18, spp=4)
    with EmuContext(device_ids=(0, 1), extra_env={"EMU_RCCL_REDUCE_FAILS": "1"}) as grp:
        grp.upload_scene(path_scene)
        grp.set_camera(cfg)
        with pytest.raises(EmuError, match="ncclReduce of member 0"):
            grp.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, max_depth=3)
        rgb, w = grp.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, max_depth=3, spp=1)   # fewer samples than members: member 0 alone
        assert (w == 1).all()
