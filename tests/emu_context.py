"""Client of tests/emu/emu_library.cpp: the device library compiled for the CPU as a checker (AddressSanitizer and
LeakSanitizer, work-items as fibres), served over a pipe. EmuContext has the methods of raytracing_amd.device.Context that
take host arrays, so a CPU test reads like its GPU counterpart. A sanitizer report ends the server; the next call then
raises with the report, and close() raises when the process leaves one behind at exit (leaks)."""
import ctypes as C
import os
import shutil
import struct
import subprocess

import numpy as np

from raytracing_amd._ctypes_defs import (RT_INTEGRATOR_PATH, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM, RtRenderParams, RtStats)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "bin")


def build_server(defines=(), suffix=""):
    """g++ on tests/emu/emu_library.cpp (about a minute); rebuilt when a source it includes is newer. None without g++.
    defines / suffix: a build of the library with other compile-time constants, kept beside the default one."""
    exe = os.path.join(BIN, "emu_library" + suffix)
    emu = os.path.join(ROOT, "tests", "emu")
    dev = os.path.join(ROOT, "raytracing_amd", "csrc", "device")
    sources = [os.path.join(emu, n) for n in ("emu_library.cpp", "simt.h", os.path.join("hip", "hip_runtime.h"), os.path.join("rccl", "rccl.h"))]
    sources += [os.path.join(dev, n) for n in os.listdir(dev)] + [os.path.join(ROOT, "include", "raytracing_amd.h")]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(s) for s in sources):
        return exe
    if shutil.which("g++") is None:
        return None
    os.makedirs(BIN, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan", "-fno-omit-frame-pointer", "-ffp-contract=off", "-Wall",
           "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-unused-variable", *["-D" + d for d in defines], "-I" + emu, "-o", exe, sources[0], "-ldl",
           "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the emulated library failed:\n" + r.stderr[-3000:])
    return exe


class EmuError(RuntimeError):
    pass


def _f32(a):
    return np.ascontiguousarray(a, np.float32).tobytes()


class EmuContext:
    def __init__(self, device_ids=(0,), lazy_streams=False, defines=(), suffix="", extra_env=None):
        """lazy_streams: queued work runs only when something waits for it (stream synchronize, an event another stream waits
        for, a release), so work on different streams is reordered as far as the library's events allow."""
        exe = build_server(defines, suffix)
        if exe is None:
            raise EmuError("no g++ to build the emulated library with")
        self._err = open(os.path.join(BIN, f"emu_library_{os.getpid()}_{id(self)}.stderr"), "w+")
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1")
        env.pop("EMU_LAZY_STREAMS", None)
        env.pop("EMU_LAST_WORKGROUP_FIRST", None)
        if lazy_streams:   # the second schedule also runs the workgroups of every grid from the last to the first
            env["EMU_LAZY_STREAMS"] = "1"
            env["EMU_LAST_WORKGROUP_FIRST"] = "1"
        env.update(extra_env or {})
        self._p = subprocess.Popen([exe], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=self._err, env=env)
        self.config = None
        try:
            self._call(1, struct.pack("<i", len(device_ids)) + np.asarray(device_ids, np.int32).tobytes())
        except EmuError:
            self._p.kill()
            self._p.wait()
            self._p = None
            self._err.close()
            os.unlink(self._err.name)
            raise

    # -- transport -----------------------------------------------------------------------------------------------------
    def _stderr(self):
        self._err.flush()
        self._err.seek(0)
        return self._err.read()[-6000:]

    def _call(self, op, payload=b""):
        try:
            self._p.stdin.write(struct.pack("<IQ", op, len(payload)) + payload)
            self._p.stdin.flush()
            head = self._p.stdout.read(12)
        except (BrokenPipeError, OSError):
            head = b""
        if len(head) != 12:
            self._p.wait()
            raise EmuError(f"the emulated library ended (exit status {self._p.returncode}):\n{self._stderr()}")
        rc, n = struct.unpack("<iQ", head)
        body = self._p.stdout.read(n) if n else b""
        if rc != 0:
            raise EmuError(f"rt error {rc}: {body.decode(errors='replace')}")
        return body

    def close(self):
        """rt_destroy, then the end of the process: raises when a sanitizer had something to say."""
        if getattr(self, "_p", None) is None:
            return
        p, self._p = self._p, None
        try:
            if p.poll() is None:
                p.stdin.write(struct.pack("<IQ", 12, 0))
                p.stdin.flush()
                p.stdout.read(12)
                p.stdin.close()
            p.wait(timeout=120)
        finally:
            report = self._stderr()
            self._err.close()
            os.unlink(self._err.name)
        if p.returncode != 0 or "ERROR: AddressSanitizer" in report or "LeakSanitizer" in report:
            raise EmuError(f"exit status {p.returncode}:\n{report}")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- the methods of raytracing_amd.device.Context (host-array flavours) -------------------------------------------
    def upload_scene(self, scene):
        has_n, has_uv = scene.tri_n is not None, scene.tri_uv is not None
        spheres = scene.spheres if scene.spheres is not None and len(scene.spheres) else None
        flags = (1 if has_n else 0) | (2 if has_uv else 0) | (4 if spheres is not None else 0)
        parts = [struct.pack("<7I", scene.n_nodes, len(scene.leaf_idx), scene.n_prims, len(scene.materials), len(scene.lights),
                             0 if spheres is None else len(spheres), flags),
                 _f32(scene.bounds), np.ascontiguousarray(scene.nodes, np.uint32).tobytes(), np.ascontiguousarray(scene.leaf_idx, np.int32).tobytes(),
                 _f32(scene.tri_p)]
        if has_n:
            parts.append(_f32(scene.tri_n))
        if has_uv:
            parts.append(_f32(scene.tri_uv))
        parts += [np.ascontiguousarray(scene.material_id, np.int32).tobytes(), np.ascontiguousarray(scene.light_id, np.int32).tobytes(),
                  np.ascontiguousarray(scene.materials).tobytes(), np.ascontiguousarray(scene.lights).tobytes()]
        if spheres is not None:
            parts += [np.ascontiguousarray(scene.shape_type, np.uint8).tobytes(), np.ascontiguousarray(spheres).tobytes()]
        self._call(2, b"".join(parts))
        if scene.config is not None:
            self.set_camera(scene.config)

    def set_camera(self, cfg):
        self._call(3, _f32(cfg.raster_to_camera) + _f32(cfg.camera_to_world) + struct.pack("<6i", cfg.width, cfg.height, *cfg.sample_bounds))
        self._call(4, struct.pack("<2f4if", cfg.filter_radius[0], cfg.filter_radius[1], *cfg.cropped_bounds, cfg.max_sample_luminance))
        self.config = cfg

    def set_option(self, option, value):
        self._call(5, struct.pack("<2i", option, value))

    def set_trace_schedule(self, schedule):
        self.set_option(1, {"auto": 0, "lockstep": 1, "stream": 2}[schedule])

    def set_pipelines(self, n):
        self.set_option(2, int(n))

    def _trace(self, op, o, d, tmax):
        n = len(o)
        body = struct.pack("<QI", n, 0 if tmax is None else 1) + _f32(o) + _f32(d) + (b"" if tmax is None else _f32(tmax))
        return n, np.frombuffer(self._call(op, body), np.uint8)

    def trace_closest(self, o, d, tmax=None):
        n, raw = self._trace(6, o, d, tmax)
        return raw[:4 * n].view(np.int32).copy(), raw[4 * n:8 * n].view(np.float32).copy(), raw[8 * n:].view(np.float32).reshape(n, 3).copy()

    def trace_any(self, o, d, tmax=None):
        return self._trace(7, o, d, tmax)[1].copy()

    def _render(self, entry, integrator, spp, max_depth, rng_mode, seed, spp_begin, spp_end, spp_per_pass, collect_stats, rr_threshold, rgb_sum,
                weight_sum):
        cfg = self.config
        p = RtRenderParams()
        p.integrator = cfg.integrator if integrator is None else integrator
        p.spp_total = cfg.spp if spp is None else spp
        p.spp_begin = spp_begin
        p.spp_end = p.spp_total if spp_end is None else spp_end
        p.max_depth = cfg.max_depth if max_depth is None else max_depth
        p.rr_threshold = rr_threshold
        if rng_mode is None:
            rng_mode = RT_RNG_COUNTER if p.integrator == RT_INTEGRATOR_PATH else RT_RNG_REFERENCE_TILE_STREAM
        p.rng_mode = rng_mode
        p.seed = seed
        p.spp_per_pass = spp_per_pass
        p.collect_stats = 1 if collect_stats else 0
        h, w = cfg.height, cfg.width
        rgb = np.zeros((h, w, 3), np.float32) if rgb_sum is None else rgb_sum
        ws = np.zeros((h, w), np.float32) if weight_sum is None else weight_sum
        raw = np.frombuffer(self._call(8, bytes(p) + struct.pack("<I", entry) + _f32(rgb) + _f32(ws)), np.float32)
        return raw[:3 * h * w].reshape(h, w, 3).copy(), raw[3 * h * w:].reshape(h, w).copy()

    def render(self, integrator=None, spp=None, max_depth=None, rng_mode=None, seed=0, spp_begin=0, spp_end=None, spp_per_pass=0,
               collect_stats=False, rr_threshold=1.0, rgb_sum=None, weight_sum=None):
        return self._render(0, integrator, spp, max_depth, rng_mode, seed, spp_begin, spp_end, spp_per_pass, collect_stats, rr_threshold, rgb_sum,
                            weight_sum)

    def render_noalloc(self, integrator=None, spp=None, max_depth=None, rng_mode=None, seed=0, spp_begin=0, spp_end=None, spp_per_pass=0,
                       rr_threshold=1.0, rgb_sum=None, weight_sum=None):
        """rt_render_noalloc_dev on buffers of the server (there, device pointers are heap pointers)."""
        return self._render(1, integrator, spp, max_depth, rng_mode, seed, spp_begin, spp_end, spp_per_pass, False, rr_threshold, rgb_sum, weight_sum)

    def develop(self, rgb_sum, weight_sum, film_scale=1.0):
        raw = self._call(9, struct.pack("<f", film_scale) + _f32(rgb_sum) + _f32(weight_sum))
        return np.frombuffer(raw, np.uint8).reshape(np.shape(rgb_sum)).copy()

    def stats(self):
        return RtStats.from_buffer_copy(self._call(10)).as_dict()

    def reset_stats(self):
        self._call(11)
