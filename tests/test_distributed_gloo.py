"""The multi-GPU split (sample-index shards + one sum-reduce of the film), exercised with two CPU processes over
gloo. The CPU oracle stands in for the per-rank renderer; the sharding and the collective are the code under test."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_lib import Oracle
    from raytracing_amd.distributed import render_sharded, shard_range
    from raytracing_amd.scene import FlatScene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = FlatScene.load_npz(os.path.join(GOLDEN, "cornellBox_path.npz"))
    cfg = scene.config.with_resolution(40, 40, spp=7)
    orc = Oracle(scene, cfg)

    def render_range(b, e):
        rgb, w, _ = orc.render(integrator=0, rng_mode=0, seed=11, spp_begin=b, spp_end=e, threads=1)
        return rgb, w

    rgb, w = render_sharded(render_range, cfg.spp)
    dist.barrier()
    if rank == 0:
        np.savez(os.path.join(outdir, "reduced.npz"), rgb=rgb.numpy(), w=w.numpy(), shard=np.array(shard_range(cfg.spp, rank, world)))
    dist.destroy_process_group()


def test_two_rank_sharded_render_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle_lib import Oracle
    from raytracing_amd.scene import FlatScene
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "reduced.npz"))
    scene = FlatScene.load_npz(os.path.join(GOLDEN, "cornellBox_path.npz"))
    cfg = scene.config.with_resolution(40, 40, spp=7)
    full, fw, _ = Oracle(scene, cfg).render(integrator=0, rng_mode=0, seed=11, threads=1)
    assert list(z["shard"]) == [0, 3]
    assert np.array_equal(z["w"], fw)
    np.testing.assert_allclose(z["rgb"], full, rtol=2e-6, atol=1e-7)


def test_shard_ranges_partition_the_samples():
    from raytracing_amd.distributed import shard_range
    for total in (1, 7, 64, 1024):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 8, 8)
