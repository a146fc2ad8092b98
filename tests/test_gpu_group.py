"""rt_create_multi on the one GPU a test box has: the same device listed two and three times is the library's one-GPU
rehearsal of its multi-GPU path (scene and camera mirrored to every member, the job split by sample index, the members'
packed films summed into member 0's; with distinct devices that sum is one ncclReduce, here a device-local kernel in
member order because RCCL refuses duplicate devices)."""
import numpy as np
import pytest

from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM
from raytracing_amd.distributed import shard_range

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("members", [2, 3])
def test_group_render_is_the_sum_of_its_shards(gpu_ctx, path_scene, members):
    from raytracing_amd.device import Context
    cfg = path_scene.config.with_resolution(160, 120, spp=11)
    gpu_ctx.upload_scene(path_scene)
    gpu_ctx.set_camera(cfg)
    kw = dict(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=21, spp=11, max_depth=15)
    full, fw = gpu_ctx.render(**kw)
    # what the group computes, member by member, on the single-GPU context: every shard into a zeroed film, summed in member order
    acc = np.zeros_like(full)
    accw = np.zeros_like(fw)
    for k in range(members):
        b, e = shard_range(11, k, members)
        rgb_k, w_k = gpu_ctx.render(spp_begin=b, spp_end=e, **kw)
        acc = (acc + rgb_k).astype(np.float32) if k else rgb_k.copy()
        accw = (accw + w_k).astype(np.float32) if k else w_k.copy()
    with Context(device_ids=[0] * members) as grp:
        assert grp.device_count() == members
        grp.upload_scene(path_scene)
        grp.set_camera(cfg)
        grp.reset_stats()
        rgb, w = grp.render(**kw)
        st = grp.stats()
        assert st["camera_samples"] == 160 * 120 * 11, "every sample rendered exactly once across the members"
        assert np.array_equal(w, fw) and np.array_equal(accw, fw)
        assert np.array_equal(rgb, acc), "the group's film is the member-order sum of the shard films"
        # against the unsharded render: the same samples, summed in another order
        assert np.abs(rgb - full).max() <= 4e-6 * np.abs(full).max()
        # accumulation into non-zero caller buffers, and a Whitted job (shards by sample index as well)
        rgb2, w2 = grp.render(rgb_sum=rgb.copy(), weight_sum=w.copy(), **kw)
        assert np.array_equal(w2, 2 * fw) and np.abs(rgb2 - 2 * full).max() <= 1e-5 * np.abs(full).max()
        wk = dict(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_COUNTER, seed=5, spp=6, max_depth=4)
        a, aw = grp.render(**wk)
        b_, bw = gpu_ctx.render(**wk)
        assert np.array_equal(aw, bw) and np.abs(a - b_).max() <= 4e-6 * max(1e-6, np.abs(b_).max())
        # a job that cannot be cut by sample index runs on member 0 alone and equals the single-GPU result bit for bit
        tk = dict(integrator=RT_INTEGRATOR_WHITTED, rng_mode=RT_RNG_REFERENCE_TILE_STREAM, spp=3, max_depth=3)
        a, aw = grp.render(**tk)
        b_, bw = gpu_ctx.render(**tk)
        assert np.array_equal(aw, bw) and np.array_equal(a, b_)
        # the group's own buffers outlive that job (its offset search allocates on member 0): the first job again
        rgb3, w3 = grp.render(**kw)
        assert np.array_equal(w3, fw) and np.array_equal(rgb3, rgb)
        # fewer samples than members: member 0 renders them all
        a, aw = grp.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=2, spp=1, max_depth=3)
        b_, bw = gpu_ctx.render(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=2, spp=1, max_depth=3)
        assert np.array_equal(aw, bw) and np.array_equal(a, b_)


def test_two_device_renders_back_to_back_on_a_group(gpu_ctx, path_scene):
    """rt_render_dev returns once the work is queued: a second call right behind the first clears and refills the members'
    packed films while the first call's sums may still be pending. Both films must come out as from the host entry point."""
    import torch
    from raytracing_amd.device import Context
    cfg = path_scene.config.with_resolution(320, 240, spp=8)
    kw = dict(integrator=RT_INTEGRATOR_PATH, rng_mode=RT_RNG_COUNTER, seed=4, spp=8, max_depth=15)
    dev = torch.device("cuda", 0)
    with Context(device_ids=[0, 0]) as grp:
        grp.upload_scene(path_scene)
        grp.set_camera(cfg)
        want, want_w = grp.render(**kw)
        films = [(torch.zeros((240, 320, 3), dtype=torch.float32, device=dev), torch.zeros((240, 320), dtype=torch.float32, device=dev)) for _ in range(3)]
        torch.cuda.synchronize()
        for rgb, w in films:
            grp.render_dev(rgb.data_ptr(), w.data_ptr(), **kw)
        grp.synchronize()
        for rgb, w in films:
            assert np.array_equal(w.cpu().numpy(), want_w)
            assert np.array_equal(rgb.cpu().numpy(), want)
