#!/usr/bin/env python3
"""Wave-wide operations of the streaming tracer's two flavours on the CPU emulation (tests/emu), on incoherent rays in a
triangle soup: a count of loop rounds, i.e. of memory round trips of a latency-bound wave. usage: tools/emu_rounds.py [n_triangles] [n_rays]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import flatbin  # noqa: E402
import raygen  # noqa: E402
from raytracing_amd._ctypes_defs import RT_INTEGRATOR_PATH, RT_RNG_COUNTER  # noqa: E402
from raytracing_amd.scene import FlatScene  # noqa: E402


def main():
    n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    base = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz")).config
    scene, cfg = raygen.soup_scene(n_tris, base)
    o, d = raygen.incoherent_rays(scene.bounds, n, seed=22)
    tmp = tempfile.mkdtemp()
    flatbin.write_scene(tmp + "/s.bin", scene, cfg, RT_INTEGRATOR_PATH, 1, 5, RT_RNG_COUNTER, 1)
    flatbin.write_rays(tmp + "/r.bin", o, d)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", EMU_OPS_BY_LINE="1")
    for exe in ("emu_trace_m0", "emu_trace_m1", "emu_trace_m1t12"):
        if not os.path.exists(os.path.join(ROOT, "tests", "bin", exe)):
            continue
        r = subprocess.run([os.path.join(ROOT, "tests", "bin", exe), tmp + "/s.bin", tmp + "/r.bin", tmp + "/o.bin", "stream", "1", "40"],
                           capture_output=True, text=True, env=env)
        print(exe, r.returncode, r.stdout.strip(), r.stderr.strip()[-500:])


if __name__ == "__main__":
    main()
