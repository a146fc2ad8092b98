"""The C-ABI shared libraries load on a machine without a GPU and export every symbol their headers declare.
No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_device_library_exports_every_declared_symbol():
    from raytracing_amd.device import EXPORTED_SYMBOLS, device_lib
    names = _declared("raytracing_amd.h")
    assert sorted(EXPORTED_SYMBOLS) == names
    lib = device_lib()
    for n in names:
        assert getattr(lib, n) is not None
    assert b"gfx950" in lib.rt_version()


def test_host_library_exports_every_declared_symbol():
    from raytracing_amd.scene import host_lib
    names = _declared("rt_host.h")
    assert "rt_host_load_json" in names and len(names) >= 10
    lib = host_lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_layouts_match_the_header():
    """Sizes that the C compiler and the ctypes mirrors must agree on (checked against a compiled probe)."""
    import subprocess
    import tempfile
    from raytracing_amd import _ctypes_defs as d
    src = r'''
#include <stdio.h>
#include "raytracing_amd.h"
#include "rt_host.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_material), sizeof(rt_light), sizeof(rt_scene_desc),
         sizeof(rt_render_params), sizeof(rt_stats), sizeof(rt_host_config), sizeof(rt_host_tree_stats), sizeof(rt_sphere));
  return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "p")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mirrors = [d.RtMaterial, d.RtLight, d.RtSceneDesc, d.RtRenderParams, d.RtStats, d.RtHostConfig, d.RtHostTreeStats, d.RtSphere]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    """On a machine without a GPU the device path refuses to start; nothing renders on the CPU instead."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from raytracing_amd.device import Context, RtError
    with pytest.raises(RtError, match="no CPU fallback"):
        Context(0)


def test_product_code_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), f"{os.path.join(dirpath, f)} mentions the oracle"
    mk = open(os.path.join(ROOT, "Makefile")).read()
    dev_rule = mk[mk.index("raytracing_amd/lib/libraytracing_amd.so:"):mk.index("oracle/liboracle.so:")]
    assert "oracle" not in dev_rule


def test_product_code_has_no_cpu_path_and_no_stand_ins():
    """tests/emu compiles the library's sources for the CPU as a checker. Nothing of it may leak the other way: the package
    never names the stand-in headers or their namespaces, the shipped library is built by hipcc for gfx950 alone, and its
    only concessions are two macros with a device definition (a register pin, the dynamic LDS declaration)."""
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for word in ("simt::", "emu_rt", "emu_rccl", "tests/emu", "EMU_"):
                    assert word not in text, f"{os.path.join(dirpath, f)} mentions {word}"
    mk = open(os.path.join(ROOT, "Makefile")).read()
    dev_rule = mk[mk.index("raytracing_amd/lib/libraytracing_amd.so:"):mk.index("oracle/liboracle.so:")]
    assert "$(HIPCC)" in dev_rule and "emu" not in dev_rule
    assert "--offload-arch=gfx950" in mk[mk.index("HIPFLAGS ="):mk.index("HOST_SRC")]


def test_device_trig_helpers_are_correctly_rounded():
    """The kernels evaluate cos / sin in double and round once (the reference calls libm's cosf / sinf, whose last
    bit is implementation-defined). The helpers are host-callable: compare them with numpy's double results rounded
    to float, over the ranges the path uses (disk mapping: [-pi/4, 3pi/4]; sphere azimuth / polar angle: [0, 2pi])."""
    import ctypes as C
    from raytracing_amd.device import device_lib
    lib = device_lib()
    rs = np.random.RandomState(5)
    for name, lo, hi in (("rt_debug_sincos_disk", -np.pi / 4, 3 * np.pi / 4), ("rt_debug_sincos_cr", -2 * np.pi, 2 * np.pi)):
        fn = getattr(lib, name)
        fn.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        fn.restype = None
        th = np.concatenate([rs.uniform(lo, hi, 20000), [v for v in (0.0, lo, hi, np.pi / 2, np.pi, np.pi / 4) if lo <= v <= hi]]).astype(np.float32)
        s, c = C.c_float(), C.c_float()
        bad = 0
        for t in th:
            fn(float(t), C.byref(s), C.byref(c))
            bad += (np.float32(np.sin(np.float64(t))) != np.float32(s.value)) + (np.float32(np.cos(np.float64(t))) != np.float32(c.value))
        assert bad == 0, f"{name}: {bad} results differ from the correctly rounded value"


def test_shard_rule_covers_every_job_exactly_once():
    """rt_shard_range (the split rt_render applies inside a multi-GPU context, host arithmetic only): contiguous shards
    in member order, union = the job, sizes balanced to within one sample, also when there are fewer samples than
    members or the job does not start at sample 0."""
    from raytracing_amd.device import device_lib
    from raytracing_amd.distributed import shard_range
    lib = device_lib()
    b, e = C.c_int(), C.c_int()
    for begin, end in ((0, 1024), (0, 7), (3, 3), (5, 1000), (0, 1), (17, 20)):
        for g in (1, 2, 3, 4, 8, 64):
            at = begin
            sizes = []
            for k in range(g):
                lib.rt_shard_range(begin, end, k, g, C.byref(b), C.byref(e))
                assert b.value == at and e.value >= b.value
                sizes.append(e.value - b.value)
                at = e.value
            assert at == end and max(sizes) - min(sizes) <= 1
    assert [shard_range(1024, k, 8) for k in range(8)] == [(128 * k, 128 * (k + 1)) for k in range(8)]
    assert [shard_range(10, k, 4) for k in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    # the Python rule (plain integers, no library needed) is the library's rule
    for total in (0, 1, 7, 1000, 1024):
        for g in (1, 3, 8, 64):
            for k in range(g):
                lib.rt_shard_range(0, total, k, g, C.byref(b), C.byref(e))
                assert shard_range(total, k, g) == (b.value, e.value)


def test_shard_range_with_bad_arguments_gives_the_empty_range():
    """No members, a member outside the group or an inverted job: the empty range at spp_begin, no division by zero; a NULL
    output pointer is skipped."""
    from raytracing_amd.device import device_lib
    lib = device_lib()
    b, e = C.c_int(-7), C.c_int(-7)
    for begin, end, k, g in ((5, 100, 0, 0), (5, 100, 0, -3), (5, 100, 4, 4), (5, 100, -1, 4), (100, 5, 0, 2)):
        lib.rt_shard_range(begin, end, k, g, C.byref(b), C.byref(e))
        assert (b.value, e.value) == (begin, begin)
    lib.rt_shard_range(0, 10, 1, 2, None, C.byref(e))
    assert e.value == 10
    lib.rt_shard_range(0, 10, 1, 2, C.byref(b), None)
    assert b.value == 5


def test_a_device_list_that_repeats_only_some_devices_is_refused():
    """rt_create_multi takes distinct devices (RCCL) or one device repeated (rehearsal); {0, 1, 0} is neither. The check comes
    before any device is touched, so it runs without a GPU."""
    from raytracing_amd.device import device_lib
    lib = device_lib()
    ctx = C.c_void_p()
    for ids in ([0, 1, 0], [0, 1, 1], [2, 0, 2, 1]):
        arr = (C.c_int * len(ids))(*ids)
        rc = lib.rt_create_multi(arr, len(ids), C.byref(ctx))
        assert rc == -1 and not ctx.value, ids   # RT_ERR_INVALID_ARG
        assert b"distinct" in lib.rt_last_error(None)


def test_rccl_offers_what_the_multi_gpu_context_binds():
    """rt_create_multi opens librccl at run time (the device library does not link it): the entry points it looks up must
    exist in the RCCL of this image, and the device library must not depend on RCCL at load time."""
    import subprocess
    from raytracing_amd.device import device_lib
    device_lib()
    path = os.path.join(ROOT, "raytracing_amd", "lib", "libraytracing_amd.so")
    needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed.lower()
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = C.CDLL(name)
            break
        except OSError:
            continue
    assert rccl is not None, "no librccl on this machine"
    for sym in ("ncclCommInitAll", "ncclCommDestroy", "ncclReduce", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert getattr(rccl, sym) is not None
