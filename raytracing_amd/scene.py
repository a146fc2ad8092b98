"""Flat scenes: numpy mirrors of rt_scene_desc, produced by the host front-end (librt_host.so).

A FlatScene is what crosses the C ABI (include/raytracing_amd.h): the reference's KD-tree node array,
leaf index list, world-space triangles, per-primitive material / light ids, material and light tables,
plus (when it came from a scene file) the render configuration the reference's Renderer holds.
"""
import ctypes as C
import os
from dataclasses import dataclass, field, replace
from typing import Optional

import numpy as np

from ._ctypes_defs import (RtHostConfig, RtHostTreeStats, RtLight, RtMaterial, RtSceneDesc, RtSphere)

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_host = None


def host_lib():
    """Loads librt_host.so (CPU only). Raises if it has not been built (run `make host`)."""
    global _host
    if _host is None:
        path = os.path.join(_LIB_DIR, "librt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make host` (or __graft_entry__.build())")
        lib = C.CDLL(path)
        lib.rt_host_last_error.restype = C.c_char_p
        lib.rt_host_load_json.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rt_host_build.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.rt_host_free.argtypes = [C.c_void_p]
        lib.rt_host_set_build_threads.argtypes = [C.c_int]
        lib.rt_host_set_build_threads.restype = None
        lib.rt_host_scene_desc.argtypes = [C.c_void_p]
        lib.rt_host_scene_desc.restype = C.POINTER(RtSceneDesc)
        lib.rt_host_get_config.argtypes = [C.c_void_p, C.POINTER(RtHostConfig)]
        lib.rt_host_tree_stats_get.argtypes = [C.c_void_p, C.POINTER(RtHostTreeStats)]
        lib.rt_host_camera.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.rt_host_camera_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        lib.rt_host_develop.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        _host = lib
    return _host


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"rt_host error {rc}: {host_lib().rt_host_last_error().decode()}")


@dataclass
class RenderConfig:
    integrator: int = 0
    max_depth: int = 2
    spp: int = 1
    width: int = 800
    height: int = 600
    cropped_bounds: tuple = (0, 0, 800, 600)
    sample_bounds: tuple = (0, 0, 800, 600)
    film_scale: float = 1.0
    raster_to_camera: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32).reshape(16))
    camera_to_world: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32).reshape(16))
    fov: float = 39.0
    eye: tuple = (0.0, 0.0, 0.0)
    focus: tuple = (0.0, 0.0, 1.0)
    up: tuple = (0.0, 1.0, 0.0)
    filename: str = "rendered.png"
    filter_radius: tuple = (0.5, 0.5)
    max_sample_luminance: float = float("inf")

    def with_resolution(self, width, height, spp=None):
        """Same camera and filter at another film resolution (full crop window)."""
        r2c, c2w = camera_matrices(self.fov, self.eye, self.focus, self.up, width, height)
        crop = (0, 0, width, height)
        return RenderConfig(self.integrator, self.max_depth, self.spp if spp is None else spp, width, height,
                            crop, sample_bounds(crop, self.filter_radius), self.film_scale, r2c, c2w,
                            self.fov, self.eye, self.focus, self.up, self.filename, self.filter_radius,
                            self.max_sample_luminance)

    def with_filter(self, radius, max_sample_luminance=None):
        """Same film with another Box filter radius (and optionally another MaxLum)."""
        radius = (float(radius[0]), float(radius[1]))
        return replace(self, filter_radius=radius, sample_bounds=sample_bounds(self.cropped_bounds, radius),
                       max_sample_luminance=self.max_sample_luminance if max_sample_luminance is None else float(max_sample_luminance))


def sample_bounds(cropped_bounds, radius):
    """Film::getSampleBounds (Object/Film.cpp:79-85) in float32, as the reference computes it."""
    f = np.float32
    x0 = np.floor(f(cropped_bounds[0]) + f(0.5) - f(radius[0]))
    y0 = np.floor(f(cropped_bounds[1]) + f(0.5) - f(radius[1]))
    x1 = np.ceil(f(cropped_bounds[2]) - f(0.5) + f(radius[0]))
    y1 = np.ceil(f(cropped_bounds[3]) - f(0.5) + f(radius[1]))
    return (int(min(x0, x1)), int(min(y0, y1)), int(max(x0, x1)), int(max(y0, y1)))


def camera_matrices(fov, eye, focus, up, width, height):
    r2c = np.zeros(16, np.float32)
    c2w = np.zeros(16, np.float32)
    e = np.asarray(eye, np.float32)
    f = np.asarray(focus, np.float32)
    u = np.asarray(up, np.float32)
    _check(host_lib().rt_host_camera(float(fov), e.ctypes.data, f.ctypes.data, u.ctypes.data, int(width), int(height),
                                     r2c.ctypes.data, c2w.ctypes.data))
    return r2c, c2w


def camera_ray(cfg: RenderConfig, film_x, film_y):
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    _check(host_lib().rt_host_camera_ray(cfg.raster_to_camera.ctypes.data, cfg.camera_to_world.ctypes.data,
                                         float(film_x), float(film_y), o.ctypes.data, d.ctypes.data))
    return o, d


def develop(rgb_sum, weight_sum, film_scale=1.0):
    """rgb_sum (H,W,3) and weight_sum (H,W) float32 -> uint8 (H,W,3), the reference's image output arithmetic."""
    rgb_sum = np.ascontiguousarray(rgb_sum, np.float32)
    weight_sum = np.ascontiguousarray(weight_sum, np.float32)
    h, w = weight_sum.shape
    out = np.zeros((h, w, 3), np.uint8)
    _check(host_lib().rt_host_develop(rgb_sum.ctypes.data, weight_sum.ctypes.data, w, h, float(film_scale), out.ctypes.data))
    return out


_MAT_DTYPE = np.dtype([("type", np.int32), ("rgb", np.float32, 3), ("eta", np.float32)])
_LIGHT_DTYPE = np.dtype([("prim_id", np.int32), ("emit", np.float32, 3), ("two_sided", np.int32)])
_SPHERE_DTYPE = np.dtype([("prim_id", np.int32), ("radius", np.float32), ("object_to_world", np.float32, 16),
                          ("world_to_object", np.float32, 16)])
assert _MAT_DTYPE.itemsize == C.sizeof(RtMaterial) and _LIGHT_DTYPE.itemsize == C.sizeof(RtLight)
assert _SPHERE_DTYPE.itemsize == C.sizeof(RtSphere)


@dataclass
class FlatScene:
    nodes: np.ndarray         # (n_nodes, 2) uint32
    leaf_idx: np.ndarray      # (n_leaf_idx,) int32
    bounds: np.ndarray        # (6,) float32
    tri_p: np.ndarray         # (n_prims, 9) float32
    tri_n: Optional[np.ndarray]   # (n_prims, 9) float32 or None
    material_id: np.ndarray   # (n_prims,) int32
    light_id: np.ndarray      # (n_prims,) int32
    materials: np.ndarray     # structured (_MAT_DTYPE)
    lights: np.ndarray        # structured (_LIGHT_DTYPE)
    config: Optional[RenderConfig] = None
    tree_stats: Optional[dict] = None
    spheres: Optional[np.ndarray] = None   # structured (_SPHERE_DTYPE); primitives listed here are spheres
    tri_uv: Optional[np.ndarray] = None    # (n_prims, 6) float32 or None: texture coordinates of the three corners

    @property
    def shape_type(self):
        st = np.zeros(self.n_prims, np.uint8)
        if self.spheres is not None and len(self.spheres):
            st[self.spheres["prim_id"]] = 1
        return st

    @property
    def n_prims(self):
        return int(self.tri_p.shape[0])

    @property
    def n_nodes(self):
        return int(self.nodes.shape[0])

    def desc(self):
        """Returns (RtSceneDesc, keepalive). Pointers reference this object's arrays."""
        d = RtSceneDesc()
        keep = []

        def ptr(a, ty):
            a = np.ascontiguousarray(a)
            keep.append(a)
            return C.cast(a.ctypes.data, C.POINTER(ty))

        d.n_nodes = self.n_nodes
        d.nodes = ptr(self.nodes.astype(np.uint32, copy=False), C.c_uint32)
        d.n_leaf_idx = int(self.leaf_idx.shape[0])
        if d.n_leaf_idx:
            d.leaf_idx = ptr(self.leaf_idx.astype(np.int32, copy=False), C.c_int32)
        for i in range(6):
            d.bounds[i] = float(self.bounds[i])
        d.n_prims = self.n_prims
        d.tri_p = ptr(self.tri_p.astype(np.float32, copy=False), C.c_float)
        if self.tri_n is not None:
            d.tri_n = ptr(self.tri_n.astype(np.float32, copy=False), C.c_float)
        if self.tri_uv is not None:
            d.tri_uv = ptr(self.tri_uv.astype(np.float32, copy=False), C.c_float)
        d.material_id = ptr(self.material_id.astype(np.int32, copy=False), C.c_int32)
        d.light_id = ptr(self.light_id.astype(np.int32, copy=False), C.c_int32)
        d.n_materials = int(self.materials.shape[0])
        d.materials = ptr(self.materials, RtMaterial)
        d.n_lights = int(self.lights.shape[0])
        if d.n_lights:
            d.lights = ptr(self.lights, RtLight)
        if self.spheres is not None and len(self.spheres):
            d.shape_type = ptr(self.shape_type, C.c_uint8)
            d.n_spheres = int(self.spheres.shape[0])
            d.spheres = ptr(self.spheres, RtSphere)
        return d, keep

    def save_npz(self, path):
        cfg = self.config
        extra = {}
        if cfg is not None:
            extra = dict(cfg_ints=np.array([cfg.integrator, cfg.max_depth, cfg.spp, cfg.width, cfg.height,
                                            *cfg.cropped_bounds, *cfg.sample_bounds], np.int32),
                         cfg_floats=np.array([cfg.film_scale, cfg.fov, *cfg.eye, *cfg.focus, *cfg.up, *cfg.filter_radius,
                                              cfg.max_sample_luminance], np.float32),
                         raster_to_camera=cfg.raster_to_camera, camera_to_world=cfg.camera_to_world,
                         filename=np.array(cfg.filename))
        np.savez_compressed(path, nodes=self.nodes, leaf_idx=self.leaf_idx, bounds=self.bounds, tri_p=self.tri_p,
                            tri_n=self.tri_n if self.tri_n is not None else np.zeros((0, 9), np.float32),
                            material_id=self.material_id, light_id=self.light_id,
                            mat_type=self.materials["type"], mat_rgb=self.materials["rgb"], mat_eta=self.materials["eta"],
                            light_prim=self.lights["prim_id"], light_emit=self.lights["emit"],
                            light_two_sided=self.lights["two_sided"],
                            spheres=(self.spheres if self.spheres is not None else np.zeros(0, _SPHERE_DTYPE)).view(np.uint8),
                            tri_uv=self.tri_uv if self.tri_uv is not None else np.zeros((0, 6), np.float32), **extra)

    @staticmethod
    def load_npz(path):
        z = np.load(path, allow_pickle=False)
        mats = np.zeros(z["mat_type"].shape[0], _MAT_DTYPE)
        mats["type"], mats["rgb"], mats["eta"] = z["mat_type"], z["mat_rgb"], z["mat_eta"]
        lights = np.zeros(z["light_prim"].shape[0], _LIGHT_DTYPE)
        lights["prim_id"], lights["emit"], lights["two_sided"] = z["light_prim"], z["light_emit"], z["light_two_sided"]
        cfg = None
        if "cfg_ints" in z.files:
            ci = [int(v) for v in z["cfg_ints"]]
            cf = [float(v) for v in z["cfg_floats"]]
            cfg = RenderConfig(ci[0], ci[1], ci[2], ci[3], ci[4], tuple(ci[5:9]), tuple(ci[9:13]), cf[0],
                               z["raster_to_camera"].astype(np.float32), z["camera_to_world"].astype(np.float32),
                               cf[1], tuple(cf[2:5]), tuple(cf[5:8]), tuple(cf[8:11]), str(z["filename"]))
            if len(cf) >= 14:   # files written before the filter parameters were stored hold the defaults
                cfg = replace(cfg, filter_radius=(cf[11], cf[12]), max_sample_luminance=cf[13])
        tri_n = z["tri_n"] if z["tri_n"].shape[0] else None
        spheres = None
        if "spheres" in z.files and z["spheres"].size:
            spheres = np.ascontiguousarray(z["spheres"]).view(_SPHERE_DTYPE).copy()
        tri_uv = z["tri_uv"] if "tri_uv" in z.files and z["tri_uv"].shape[0] else None
        return FlatScene(z["nodes"], z["leaf_idx"], z["bounds"], z["tri_p"], tri_n, z["material_id"], z["light_id"],
                         mats, lights, cfg, None, spheres, tri_uv)


def _from_handle(h, with_config):
    lib = host_lib()
    d = lib.rt_host_scene_desc(h).contents
    n = d.n_prims

    def arr(p, count, dtype):
        if not p or count == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(p, shape=(count,)).astype(dtype, copy=True)

    nodes = arr(d.nodes, 2 * d.n_nodes, np.uint32).reshape(-1, 2)
    leaf_idx = arr(d.leaf_idx, d.n_leaf_idx, np.int32)
    bounds = np.array(list(d.bounds), np.float32)
    tri_p = arr(d.tri_p, 9 * n, np.float32).reshape(-1, 9)
    tri_n = arr(d.tri_n, 9 * n, np.float32).reshape(-1, 9) if d.tri_n else None
    tri_uv = arr(d.tri_uv, 6 * n, np.float32).reshape(-1, 6) if d.tri_uv else None
    material_id = arr(d.material_id, n, np.int32)
    light_id = arr(d.light_id, n, np.int32)
    mats = np.zeros(d.n_materials, _MAT_DTYPE)
    for i in range(d.n_materials):
        m = d.materials[i]
        mats[i] = (m.type, tuple(m.rgb), m.eta)
    lights = np.zeros(d.n_lights, _LIGHT_DTYPE)
    for i in range(d.n_lights):
        l = d.lights[i]
        lights[i] = (l.prim_id, tuple(l.emit), l.two_sided)
    cfg = None
    if with_config:
        c = RtHostConfig()
        _check(lib.rt_host_get_config(h, C.byref(c)))
        cfg = RenderConfig(c.integrator, c.max_depth, c.spp, c.width, c.height, tuple(c.cropped_bounds),
                           tuple(c.sample_bounds), c.film_scale, np.array(list(c.raster_to_camera), np.float32),
                           np.array(list(c.camera_to_world), np.float32), c.fov, tuple(c.eye), tuple(c.focus),
                           tuple(c.up), c.filename.decode(), tuple(c.filter_radius), float(c.max_sample_luminance))
    ts = RtHostTreeStats()
    _check(lib.rt_host_tree_stats_get(h, C.byref(ts)))
    spheres = None
    if d.n_spheres:
        spheres = np.zeros(d.n_spheres, _SPHERE_DTYPE)
        for i in range(d.n_spheres):
            sp = d.spheres[i]
            spheres[i] = (sp.prim_id, sp.radius, tuple(sp.object_to_world), tuple(sp.world_to_object))
    return FlatScene(nodes, leaf_idx, bounds, tri_p, tri_n, material_id, light_id, mats, lights, cfg, ts.as_dict(), spheres, tri_uv)


def load_scene_json(path) -> FlatScene:
    """Parses a reference-format scene JSON (+ OBJ meshes) and builds the KD-tree on the host."""
    lib = host_lib()
    h = C.c_void_p()
    _check(lib.rt_host_load_json(os.fsencode(path), C.byref(h)))
    try:
        return _from_handle(h, True)
    finally:
        lib.rt_host_free(h)


def build_scene(tri_p, material_id, materials, tri_n=None, light_id=None, lights=None, config=None, tri_uv=None) -> FlatScene:
    """Builds the KD-tree over caller-supplied world-space triangles (synthetic scenes)."""
    lib = host_lib()
    tri_p = np.ascontiguousarray(tri_p, np.float32).reshape(-1, 9)
    n = tri_p.shape[0]
    material_id = np.ascontiguousarray(material_id, np.int32)
    mats = np.ascontiguousarray(materials, _MAT_DTYPE)
    tn = None if tri_n is None else np.ascontiguousarray(tri_n, np.float32)
    lid = None if light_id is None else np.ascontiguousarray(light_id, np.int32)
    lts = np.zeros(0, _LIGHT_DTYPE) if lights is None else np.ascontiguousarray(lights, _LIGHT_DTYPE)
    h = C.c_void_p()
    tuv = None if tri_uv is None else np.ascontiguousarray(tri_uv, np.float32).reshape(-1, 6)
    _check(lib.rt_host_build(n, tri_p.ctypes.data, None if tn is None else tn.ctypes.data, None if tuv is None else tuv.ctypes.data,
                             material_id.ctypes.data, None if lid is None else lid.ctypes.data,
                             mats.shape[0], mats.ctypes.data, lts.shape[0], lts.ctypes.data if lts.shape[0] else None,
                             C.byref(h)))
    try:
        s = _from_handle(h, False)
    finally:
        lib.rt_host_free(h)
    s.config = config
    return s


def make_materials(rows):
    """rows: iterable of (type, (r,g,b), eta)."""
    a = np.zeros(len(rows), _MAT_DTYPE)
    for i, (t, rgb, eta) in enumerate(rows):
        a[i] = (t, tuple(rgb), eta)
    return a


def make_lights(rows):
    """rows: iterable of (prim_id, (r,g,b), two_sided)."""
    a = np.zeros(len(rows), _LIGHT_DTYPE)
    for i, (p, rgb, ts) in enumerate(rows):
        a[i] = (p, tuple(rgb), ts)
    return a
