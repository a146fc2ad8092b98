"""The tracing kernels' device code, compiled for the CPU and run one emulated wave at a time (tests/emu/), against the
oracle: primitive ids, t, barycentrics and occlusion flags bit-equal, under AddressSanitizer. The harness includes
raytracing_amd/csrc/device/dstream.h and dtraverse.h unchanged; only <hip/hip_runtime.h> is replaced by a stand-in in which
a wave is 64 fibres that meet at every ballot and wave barrier. It covers both flavours (tables in HBM, tables
in LDS), both schedules, and the streaming tracer's merged-round build, with the LDS part of the todo stack
shortened so that a small scene uses the private part as well. What it adds to the GPU parity tests: an index that leaves a table or the stack, a wave-wide
operation that only part of a wave reaches, or a wave that never ends is reported here as such, on the CPU, instead of as a
GPU fault or hang."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import flatbin
import raygen
from oracle_lib import Oracle
from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_RNG_COUNTER

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "bin")
ANY_TMAX = 300.0
STACK_LDS = 4   # todo entries per lane in the LDS part for this build (12 in the library): a small tree then reaches the private part


def _build(merged, t_min=1):
    """g++ on tests/emu/emu_trace.cpp (a few seconds); rebuilt when a source it includes is newer."""
    exe = os.path.join(BIN, f"emu_trace_m{merged}" + (f"t{t_min}" if t_min > 1 else ""))
    sources = [os.path.join(ROOT, "tests", "emu", n) for n in ("emu_trace.cpp", "simt.h", os.path.join("hip", "hip_runtime.h"))]
    sources += [os.path.join(ROOT, "raytracing_amd", "csrc", "device", n) for n in ("dstream.h", "drefill.h", "dtraverse.h", "dmath.h")]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(s) for s in sources):
        return exe
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    os.makedirs(BIN, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan", "-fno-omit-frame-pointer", "-ffp-contract=off",
           "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-unused-variable", f"-DRT_STREAM_MERGED={merged}", f"-DRT_MERGED_T_MIN={t_min}",
           f"-DRT_STACK_LDS_LARGE={STACK_LDS}",
           "-I" + os.path.join(ROOT, "tests", "emu"), "-o", exe, sources[0]]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def _write_batch(tmp, scene, cfg, orc, o, d):
    prim, t, bary, cn = orc.trace_closest(o, d, counters=True)
    occ = orc.trace_any(o, d, np.full(len(o), ANY_TMAX, np.float32))
    flatbin.write_scene(str(tmp / "scene.bin"), scene, cfg, RT_INTEGRATOR_PATH, 1, 5, RT_RNG_COUNTER, 1)
    flatbin.write_rays(str(tmp / "rays.bin"), o, d)
    assert cn["max_stack"] > STACK_LDS + 1, "the batch must reach the private part of the todo stack"
    assert 0.5 < (prim >= 0).mean() <= 1.0 and 0.05 < occ.mean() < 0.95
    return tmp, len(o), prim, t, bary, occ


@pytest.fixture(scope="module")
def batch(path_scene, tmp_path_factory):
    """A 64 x 32 x 2 = 4096-triangle connected mesh; 3000 rays aimed at shared vertices and edges (equal-t ties: the last
    primitive tested wins, so the order of tests inside a leaf shows) and 1133 incoherent rays."""
    scene, cfg = raygen.grid_scene(64, 32, path_scene.config, width=64, height=48)
    orc = Oracle(scene, cfg)
    of, df = raygen.mesh_feature_rays(scene.tri_p[:-2], 3000, seed=5)
    oi, di = raygen.incoherent_rays(scene.bounds, 1133, seed=6)   # 4133 rays: the last chunk of 64 is ragged
    out = _write_batch(tmp_path_factory.mktemp("emu"), scene, cfg, orc, np.concatenate([of, oi]), np.concatenate([df, di]))
    prim, bary = out[2], out[4]
    assert (bary[prim >= 0] == 1).any(axis=1).mean() > 0.1, "vertex hits (ties between up to six triangles) must be exercised"
    return out


@pytest.fixture(scope="module")
def batch_lds(path_scene, tmp_path_factory):
    """cornellBox (tables in LDS on the GPU): camera rays, rays that start on the surfaces they hit, incoherent rays."""
    cfg = path_scene.config.with_resolution(64, 48)
    orc = Oracle(path_scene, cfg)
    oc, dc = raygen.camera_rays(orc, cfg, 1500, seed=1)
    os_, ds = raygen.surface_rays(orc, oc, dc, seed=2)
    oi, di = raygen.incoherent_rays(path_scene.bounds, 1100, seed=3)
    return _write_batch(tmp_path_factory.mktemp("emu_lds"), path_scene, cfg, orc, np.concatenate([oc, os_, oi]), np.concatenate([dc, ds, di]))


def _run(exe, tmp, n, mode, blocks):
    out = tmp / f"out_{os.path.basename(exe)}_{mode}.bin"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    r = subprocess.run([exe, str(tmp / "scene.bin"), str(tmp / "rays.bin"), str(out), mode, str(blocks), str(ANY_TMAX)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    raw = np.fromfile(out, np.uint8)
    assert raw.size == 21 * n
    return (raw[:4 * n].view(np.int32), raw[4 * n:8 * n].view(np.float32), raw[8 * n:20 * n].view(np.float32).reshape(n, 3), raw[20 * n:], r.stdout)


@pytest.mark.parametrize("merged,t_min,mode,blocks", [(0, 1, "lockstep", 2), (0, 1, "stream", 1), (0, 1, "stream", 3), (0, 1, "refill", 1), (0, 1, "refill", 3), (1, 1, "stream", 1),
                                                      (1, 1, "stream", 3), (1, 12, "stream", 2)])
def test_device_tracers_on_the_cpu_match_the_oracle(batch, merged, t_min, mode, blocks):
    tmp, n, prim, t, bary, occ = batch
    gp, gt, gb, ga, log = _run(_build(merged, t_min), tmp, n, mode, blocks)
    assert ("merged rounds" in log) == bool(merged and mode == "stream")
    assert np.array_equal(gp, prim), f"{np.count_nonzero(gp != prim)} of {n} primitive ids differ"
    assert np.array_equal(gt.view(np.uint32), t.view(np.uint32))
    assert np.array_equal(gb.view(np.uint32), bary.view(np.uint32))
    assert np.array_equal(ga, occ)


@pytest.mark.parametrize("mode,blocks", [("lockstep-lds", 2), ("stream-lds", 1), ("stream-lds", 3), ("refill-lds", 1), ("refill-lds", 3)])
def test_device_tracers_for_scenes_in_lds_on_the_cpu_match_the_oracle(batch_lds, mode, blocks):
    tmp, n, prim, t, bary, occ = batch_lds
    gp, gt, gb, ga, _ = _run(_build(0), tmp, n, mode, blocks)
    assert np.array_equal(gp, prim), f"{np.count_nonzero(gp != prim)} of {n} primitive ids differ"
    assert np.array_equal(gt.view(np.uint32), t.view(np.uint32))
    assert np.array_equal(gb.view(np.uint32), bary.view(np.uint32))
    assert np.array_equal(ga, occ)
