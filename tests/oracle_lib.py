"""ctypes access to oracle/liboracle.so — the CPU checker. Only tests, smoke() and bench.py's
cpu_baseline leg may import this module."""
import ctypes as C
import os

import numpy as np

from raytracing_amd._ctypes_defs import RtRenderParams, RtSceneDesc

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class OracleCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_interior", "n_leaf", "n_leaf_idx", "n_tri_tests", "n_rays", "max_stack")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OracleRenderStats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("camera_samples", C.c_uint64),
                ("rng_draws", C.c_uint64), ("seconds", C.c_double), ("threads", C.c_int32),
                ("n_interior", C.c_uint64), ("n_leaf", C.c_uint64), ("n_leaf_idx", C.c_uint64),
                ("n_tri_tests", C.c_uint64), ("counted_rays", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make oracle`")
        l = C.CDLL(path)
        l.oracle_create.argtypes = [C.POINTER(RtSceneDesc)]
        l.oracle_create.restype = C.c_void_p
        l.oracle_destroy.argtypes = [C.c_void_p]
        l.oracle_set_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        l.oracle_set_film.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float]
        l.oracle_trace_closest.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 7
        l.oracle_trace_any.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 5
        l.oracle_surface.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        l.oracle_camera_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        l.oracle_render.argtypes = [C.c_void_p, C.POINTER(RtRenderParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        l.oracle_rng_floats.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
        l.oracle_set_trig_mode.argtypes = [C.c_int]
        _lib = l
    return _lib


class Oracle:
    def __init__(self, scene, config=None):
        self._l = lib()
        d, self._keep = scene.desc()
        self._h = self._l.oracle_create(C.byref(d))
        if not self._h:
            raise RuntimeError("oracle_create failed")
        self.scene = scene
        self.config = None
        cfg = config if config is not None else scene.config
        if cfg is not None:
            self.set_camera(cfg)

    def close(self):
        if self._h:
            self._l.oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_camera(self, cfg):
        sb = np.array(cfg.sample_bounds, np.int32)
        r2c = np.ascontiguousarray(cfg.raster_to_camera, np.float32)
        c2w = np.ascontiguousarray(cfg.camera_to_world, np.float32)
        rc = self._l.oracle_set_camera(self._h, r2c.ctypes.data, c2w.ctypes.data, cfg.width, cfg.height, sb.ctypes.data)
        assert rc == 0
        cb = np.array(cfg.cropped_bounds, np.int32)
        rc = self._l.oracle_set_film(self._h, C.c_float(cfg.filter_radius[0]), C.c_float(cfg.filter_radius[1]), cb.ctypes.data,
                                     C.c_float(cfg.max_sample_luminance))
        assert rc == 0
        self.config = cfg

    def trace_closest(self, o, d, tmax=None, counters=False):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        bary = np.zeros((n, 3), np.float32)
        cn = OracleCounters()
        rc = self._l.oracle_trace_closest(self._h, n, o.ctypes.data, d.ctypes.data, None if tm is None else tm.ctypes.data,
                                          prim.ctypes.data, t.ctypes.data, bary.ctypes.data,
                                          C.addressof(cn) if counters else None)
        assert rc == 0
        return (prim, t, bary, cn.as_dict()) if counters else (prim, t, bary)

    def trace_any(self, o, d, tmax=None, counters=False):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        occ = np.zeros(n, np.uint8)
        cn = OracleCounters()
        rc = self._l.oracle_trace_any(self._h, n, o.ctypes.data, d.ctypes.data, None if tm is None else tm.ctypes.data,
                                      occ.ctypes.data, C.addressof(cn) if counters else None)
        assert rc == 0
        return (occ, cn.as_dict()) if counters else occ

    def camera_rays(self, film_xy):
        xy = np.ascontiguousarray(film_xy, np.float32)
        n = xy.shape[0]
        o = np.zeros((n, 3), np.float32)
        d = np.zeros((n, 3), np.float32)
        rc = self._l.oracle_camera_rays(self._h, n, xy.ctypes.data, o.ctypes.data, d.ctypes.data)
        assert rc == 0
        return o, d

    def surface(self, prim, bary, d):
        b = np.ascontiguousarray(bary, np.float32)
        dd = np.ascontiguousarray(d, np.float32)
        p = np.zeros(3, np.float32)
        n = np.zeros(3, np.float32)
        dpdu = np.zeros(3, np.float32)
        rc = self._l.oracle_surface(self._h, int(prim), b.ctypes.data, dd.ctypes.data, p.ctypes.data, n.ctypes.data, dpdu.ctypes.data)
        return rc, p, n, dpdu

    def render(self, integrator=None, spp=None, max_depth=None, rng_mode=1, seed=0, spp_begin=0, spp_end=None,
               threads=0, collect_stats=False, rr_threshold=1.0):
        cfg = self.config
        p = RtRenderParams()
        p.integrator = cfg.integrator if integrator is None else integrator
        p.spp_total = cfg.spp if spp is None else spp
        p.spp_begin = spp_begin
        p.spp_end = p.spp_total if spp_end is None else spp_end
        p.max_depth = cfg.max_depth if max_depth is None else max_depth
        p.rr_threshold = rr_threshold
        p.rng_mode = rng_mode
        p.seed = seed
        p.collect_stats = 1 if collect_stats else 0
        rgb = np.zeros((cfg.height, cfg.width, 3), np.float32)
        w = np.zeros((cfg.height, cfg.width), np.float32)
        st = OracleRenderStats()
        rc = self._l.oracle_render(self._h, C.byref(p), rgb.ctypes.data, w.ctypes.data, threads, C.addressof(st))
        assert rc == 0
        return rgb, w, st.as_dict()


def set_trig_mode(correctly_rounded):
    """False: libm cosf/sinf as the reference calls them. True: correctly rounded, bit-equal to the HIP kernels."""
    lib().oracle_set_trig_mode(1 if correctly_rounded else 0)


def rng_floats(seq, n):
    out = np.zeros(n, np.float32)
    lib().oracle_rng_floats(seq, n, out.ctypes.data)
    return out
