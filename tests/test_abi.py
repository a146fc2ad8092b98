"""The C-ABI shared libraries load on a machine without a GPU and export every symbol their headers declare.
No compute entry point is called here."""
import ctypes as C
import os
import re

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
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_material), sizeof(rt_light), sizeof(rt_scene_desc),
         sizeof(rt_render_params), sizeof(rt_stats), sizeof(rt_host_config), sizeof(rt_host_tree_stats));
  return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "p")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mirrors = [d.RtMaterial, d.RtLight, d.RtSceneDesc, d.RtRenderParams, d.RtStats, d.RtHostConfig, d.RtHostTreeStats]
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
