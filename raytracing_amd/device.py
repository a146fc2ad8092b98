"""ctypes binding of the device library (include/raytracing_amd.h). There is no CPU fallback: creating a
Context without the built HIP library or without a visible GPU raises."""
import ctypes as C
import os

import numpy as np

from ._ctypes_defs import (RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM,
                           RtRenderParams, RtSceneDesc, RtStats)

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_dev = None

# every symbol include/raytracing_amd.h declares
EXPORTED_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_version", "rt_upload_scene", "rt_set_camera", "rt_trace_closest",
    "rt_trace_any", "rt_trace_closest_dev", "rt_trace_any_dev", "rt_render", "rt_render_dev", "rt_synchronize", "rt_stream",
    "rt_get_stats", "rt_reset_stats", "rt_set_option", "rt_develop", "rt_develop_dev", "rt_set_film",
    "rt_debug_sincos_disk", "rt_debug_sincos_cr", "rt_debug_whitted_passes", "rt_render_noalloc_dev", "rt_create_multi", "rt_device_count", "rt_shard_range",
]


def _share_hip_runtime_with_torch():
    """PyTorch wheels bring their own HIP runtime. A process must run on one runtime: if this library pulled in the
    system's first, a later `import torch` would find no GPU. When torch is installed and not yet imported, its runtime
    is therefore loaded first (without importing torch); the dynamic loader then resolves this library against it, exactly
    as when the caller imports torch first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def device_lib():
    """Loads libraytracing_amd.so. Raises if it has not been built (run `make device`)."""
    global _dev
    if _dev is None:
        _share_hip_runtime_with_torch()
        path = os.environ.get("RT_AMD_DEVICE_LIB") or os.path.join(_LIB_DIR, "libraytracing_amd.so")   # override: experiments
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make device` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the device path")
        lib = C.CDLL(path)
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        lib.rt_create.argtypes = [i32, C.POINTER(vp)]
        lib.rt_create_multi.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
        lib.rt_device_count.argtypes = [vp]
        lib.rt_shard_range.argtypes = [i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        lib.rt_shard_range.restype = None
        lib.rt_destroy.argtypes = [vp]
        lib.rt_destroy.restype = None
        lib.rt_last_error.argtypes = [vp]
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_version.restype = C.c_char_p
        lib.rt_upload_scene.argtypes = [vp, C.POINTER(RtSceneDesc)]
        lib.rt_set_camera.argtypes = [vp, vp, vp, i32, i32, vp]
        lib.rt_set_film.argtypes = [vp, C.c_float, C.c_float, vp, C.c_float]
        lib.rt_trace_closest.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
        lib.rt_trace_any.argtypes = [vp, u64, vp, vp, vp, vp]
        lib.rt_trace_closest_dev.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
        lib.rt_trace_any_dev.argtypes = [vp, u64, vp, vp, vp, vp]
        lib.rt_render.argtypes = [vp, C.POINTER(RtRenderParams), vp, vp]
        lib.rt_render_dev.argtypes = [vp, C.POINTER(RtRenderParams), vp, vp]
        lib.rt_render_noalloc_dev.argtypes = [vp, C.POINTER(RtRenderParams), vp, vp]
        lib.rt_synchronize.argtypes = [vp]
        lib.rt_stream.argtypes = [vp]
        lib.rt_stream.restype = vp
        lib.rt_get_stats.argtypes = [vp, C.POINTER(RtStats)]
        lib.rt_reset_stats.argtypes = [vp]
        lib.rt_set_option.argtypes = [vp, i32, i32]
        lib.rt_develop.argtypes = [vp, vp, vp, C.c_float, vp]
        lib.rt_develop_dev.argtypes = [vp, vp, vp, C.c_float, vp]
        _dev = lib
    return _dev


class RtError(RuntimeError):
    pass


class Context:
    """One rt_context: on one GPU (device_id), or over several GPUs of the node (device_ids: rt_create_multi - renders are
    split by sample index and summed over RCCL inside the library; device pointers belong to device_ids[0])."""

    def __init__(self, device_id=0, device_ids=None):
        self._lib = device_lib()
        h = C.c_void_p()
        if device_ids is None:
            rc = self._lib.rt_create(int(device_id), C.byref(h))
        else:
            ids = (C.c_int * len(device_ids))(*[int(v) for v in device_ids])
            rc = self._lib.rt_create_multi(ids, len(device_ids), C.byref(h))
            device_id = int(device_ids[0])
        if rc != 0:
            raise RtError(f"rt_create failed ({rc}): {self._lib.rt_last_error(None).decode()}")
        self._h = h
        self.device_id = device_id
        self.config = None
        self.scene = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RtError(f"rt error {rc}: {self._lib.rt_last_error(self._h).decode()}")

    # -- scene / camera --------------------------------------------------------------------------
    def upload_scene(self, scene):
        d, keep = scene.desc()
        self._check(self._lib.rt_upload_scene(self._h, C.byref(d)))
        del keep
        self.scene = scene
        if scene.config is not None:
            self.set_camera(scene.config)

    def set_camera(self, cfg):
        r2c = np.ascontiguousarray(cfg.raster_to_camera, np.float32)
        c2w = np.ascontiguousarray(cfg.camera_to_world, np.float32)
        sb = np.array(cfg.sample_bounds, np.int32)
        self._check(self._lib.rt_set_camera(self._h, r2c.ctypes.data, c2w.ctypes.data, cfg.width, cfg.height, sb.ctypes.data))
        cb = np.array(cfg.cropped_bounds, np.int32)
        self._check(self._lib.rt_set_film(self._h, C.c_float(cfg.filter_radius[0]), C.c_float(cfg.filter_radius[1]), cb.ctypes.data,
                                          C.c_float(cfg.max_sample_luminance)))
        self.config = cfg

    # -- batch tracing (host arrays) ----------------------------------------------------------------
    def trace_closest(self, o, d, tmax=None):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        prim = np.full(n, -2, np.int32)
        t = np.zeros(n, np.float32)
        bary = np.zeros((n, 3), np.float32)
        self._check(self._lib.rt_trace_closest(self._h, n, o.ctypes.data, d.ctypes.data, None if tm is None else tm.ctypes.data,
                                               prim.ctypes.data, t.ctypes.data, bary.ctypes.data))
        return prim, t, bary

    def trace_any(self, o, d, tmax=None):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        occ = np.full(n, 255, np.uint8)
        self._check(self._lib.rt_trace_any(self._h, n, o.ctypes.data, d.ctypes.data, None if tm is None else tm.ctypes.data,
                                           occ.ctypes.data))
        return occ

    # -- batch tracing (device pointers, e.g. torch tensors' data_ptr()) ------------------------------
    def trace_closest_dev(self, n, o_ptr, d_ptr, tmax_ptr, prim_ptr, t_ptr, bary_ptr):
        self._check(self._lib.rt_trace_closest_dev(self._h, int(n), o_ptr, d_ptr, tmax_ptr, prim_ptr, t_ptr, bary_ptr))

    def trace_any_dev(self, n, o_ptr, d_ptr, tmax_ptr, occ_ptr):
        self._check(self._lib.rt_trace_any_dev(self._h, int(n), o_ptr, d_ptr, tmax_ptr, occ_ptr))

    # -- rendering ---------------------------------------------------------------------------------------
    def _params(self, integrator, spp, max_depth, rng_mode, seed, spp_begin, spp_end, spp_per_pass, collect_stats, rr_threshold):
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
        return p

    def render(self, integrator=None, spp=None, max_depth=None, rng_mode=None, seed=0, spp_begin=0, spp_end=None,
               spp_per_pass=0, collect_stats=False, rr_threshold=1.0, rgb_sum=None, weight_sum=None):
        """Blocking render into host arrays; returns (rgb_sum (H,W,3), weight_sum (H,W))."""
        cfg = self.config
        p = self._params(integrator, spp, max_depth, rng_mode, seed, spp_begin, spp_end, spp_per_pass, collect_stats, rr_threshold)
        rgb = np.zeros((cfg.height, cfg.width, 3), np.float32) if rgb_sum is None else rgb_sum
        w = np.zeros((cfg.height, cfg.width), np.float32) if weight_sum is None else weight_sum
        self._check(self._lib.rt_render(self._h, C.byref(p), rgb.ctypes.data, w.ctypes.data))
        return rgb, w

    def render_dev(self, rgb_ptr, weight_ptr, **kw):
        """Queues a render that accumulates into device buffers (call synchronize() to wait)."""
        p = self._params(kw.get("integrator"), kw.get("spp"), kw.get("max_depth"), kw.get("rng_mode"), kw.get("seed", 0),
                         kw.get("spp_begin", 0), kw.get("spp_end"), kw.get("spp_per_pass", 0), kw.get("collect_stats", False),
                         kw.get("rr_threshold", 1.0))
        self._check(self._lib.rt_render_dev(self._h, C.byref(p), rgb_ptr, weight_ptr))

    def render_noalloc_dev(self, rgb_ptr, weight_ptr, **kw):
        """render_dev through the entry point that allocates nothing (queues in the library's static scratch array)."""
        p = self._params(kw.get("integrator"), kw.get("spp"), kw.get("max_depth"), kw.get("rng_mode"), kw.get("seed", 0),
                         kw.get("spp_begin", 0), kw.get("spp_end"), kw.get("spp_per_pass", 0), False, kw.get("rr_threshold", 1.0))
        self._check(self._lib.rt_render_noalloc_dev(self._h, C.byref(p), rgb_ptr, weight_ptr))

    def set_trace_schedule(self, schedule):
        """Ray-to-lane schedule: "auto" (default), "lockstep" (one ray per lane) or "stream" (lanes take new rays as they finish)."""
        value = {"auto": 0, "lockstep": 1, "stream": 2, "refill": 3}[schedule]
        self._check(self._lib.rt_set_option(self._h, 1, value))

    def set_pipelines(self, n):
        """How many passes of the wavefront path tracer overlap on separate streams (1..3)."""
        self._check(self._lib.rt_set_option(self._h, 2, int(n)))

    def set_kernel_times(self, on):
        """HIP events around the extend and shade launches of Path renders (stats: extend_ms, shade_ms); off by default."""
        self._check(self._lib.rt_set_option(self._h, 3, 1 if on else 0))

    def develop(self, rgb_sum, weight_sum, film_scale=1.0):
        """Film sums (host arrays of the camera's size) -> (H, W, 3) uint8 sRGB image, computed on the device."""
        rgb_sum = np.ascontiguousarray(rgb_sum, np.float32)
        weight_sum = np.ascontiguousarray(weight_sum, np.float32)
        out = np.empty(rgb_sum.shape, np.uint8)
        self._check(self._lib.rt_develop(self._h, rgb_sum.ctypes.data, weight_sum.ctypes.data, float(film_scale), out.ctypes.data))
        return out

    def develop_dev(self, rgb_ptr, weight_ptr, out_ptr, film_scale=1.0):
        self._check(self._lib.rt_develop_dev(self._h, rgb_ptr, weight_ptr, float(film_scale), out_ptr))

    def device_count(self):
        return self._lib.rt_device_count(self._h)

    def synchronize(self):
        self._check(self._lib.rt_synchronize(self._h))

    def stream(self):
        return self._lib.rt_stream(self._h)

    def stats(self):
        s = RtStats()
        self._check(self._lib.rt_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(self._lib.rt_reset_stats(self._h))


__all__ = ["Context", "RtError", "device_lib", "EXPORTED_SYMBOLS", "RT_INTEGRATOR_PATH", "RT_INTEGRATOR_WHITTED",
           "RT_RNG_COUNTER", "RT_RNG_REFERENCE_TILE_STREAM"]
