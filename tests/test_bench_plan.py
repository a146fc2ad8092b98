"""bench.py decides how it runs before anything touches a GPU: `--gpus N` without a launcher is ONE process that drives the
library's multi-GPU context (the reference's main is one process that calls render() once, reference
src/RayTracing.cpp:32-33); under torch.distributed.run it is one process per GPU. It never refuses a device count."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_every_device_count_has_a_plan_without_a_launcher():
    import bench
    assert bench.plan(1, {})["mode"] == "single"
    for n in (2, 4, 8):
        p = bench.plan(n, {})
        assert p == {"mode": "group", "world": n, "rank": 0, "local_rank": 0}
    p = bench.plan(8, {"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"})
    assert p == {"mode": "ranks", "world": 8, "rank": 5, "local_rank": 5}
    # a launcher's world size wins over the flag (the driver passes both, and they agree)
    assert bench.plan(1, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"})["mode"] == "ranks"


def test_without_a_gpu_the_only_refusal_is_the_missing_gpu(monkeypatch):
    """On a machine without a GPU `bench.py --gpus 2` stops because the device path has no CPU fallback - not because of
    how it was launched."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert "needs a GPU" in str(e.value)
