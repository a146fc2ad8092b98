"""ctypes mirrors of the structs in include/raytracing_amd.h and include/rt_host.h."""
import ctypes as C

RT_INTEGRATOR_PATH = 0
RT_INTEGRATOR_WHITTED = 1
RT_RNG_COUNTER = 0
RT_RNG_REFERENCE_TILE_STREAM = 1
RT_MAT_LAMBERTIAN = 0
RT_MAT_MIRROR = 1
RT_MAT_GLASS = 2


class RtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("rgb", C.c_float * 3), ("eta", C.c_float)]


class RtLight(C.Structure):
    _fields_ = [("prim_id", C.c_int32), ("emit", C.c_float * 3), ("two_sided", C.c_int32)]


class RtSphere(C.Structure):
    _fields_ = [("prim_id", C.c_int32), ("radius", C.c_float), ("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32),
        ("nodes", C.POINTER(C.c_uint32)),
        ("n_leaf_idx", C.c_uint32),
        ("leaf_idx", C.POINTER(C.c_int32)),
        ("bounds", C.c_float * 6),
        ("n_prims", C.c_uint32),
        ("shape_type", C.POINTER(C.c_uint8)),
        ("tri_p", C.POINTER(C.c_float)),
        ("tri_n", C.POINTER(C.c_float)),
        ("tri_has_n", C.POINTER(C.c_uint8)),
        ("tri_uv", C.POINTER(C.c_float)),
        ("material_id", C.POINTER(C.c_int32)),
        ("light_id", C.POINTER(C.c_int32)),
        ("n_materials", C.c_uint32),
        ("materials", C.POINTER(RtMaterial)),
        ("n_lights", C.c_uint32),
        ("lights", C.POINTER(RtLight)),
        ("n_spheres", C.c_uint32),
        ("spheres", C.POINTER(RtSphere)),
    ]


class RtRenderParams(C.Structure):
    _fields_ = [
        ("integrator", C.c_int32),
        ("spp_total", C.c_int32),
        ("spp_begin", C.c_int32),
        ("spp_end", C.c_int32),
        ("max_depth", C.c_int32),
        ("rr_threshold", C.c_float),
        ("rng_mode", C.c_int32),
        ("seed", C.c_uint64),
        ("spp_per_pass", C.c_int32),
        ("collect_stats", C.c_int32),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("rays_closest", C.c_uint64),
        ("rays_any", C.c_uint64),
        ("camera_samples", C.c_uint64),
        ("n_interior", C.c_uint64),
        ("n_leaf", C.c_uint64),
        ("n_leaf_idx", C.c_uint64),
        ("n_tri_tests", C.c_uint64),
        ("counted_rays", C.c_uint64),
        ("extend_launches", C.c_uint64),
        ("extend_ms", C.c_double),
        ("extend_rays", C.c_uint64),
        ("render_ms", C.c_double),
        ("shade_launches", C.c_uint64),
        ("shade_ms", C.c_double),
        ("queue_bytes_per_pipeline", C.c_uint64),
        ("queue_paths", C.c_uint64),
        ("pipelines", C.c_int32),
        ("scene_in_lds", C.c_int32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class RtHostConfig(C.Structure):
    _fields_ = [
        ("integrator", C.c_int32),
        ("max_depth", C.c_int32),
        ("spp", C.c_int32),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("cropped_bounds", C.c_int32 * 4),
        ("sample_bounds", C.c_int32 * 4),
        ("filter_radius", C.c_float * 2),
        ("film_scale", C.c_float),
        ("raster_to_camera", C.c_float * 16),
        ("camera_to_world", C.c_float * 16),
        ("fov", C.c_float),
        ("eye", C.c_float * 3),
        ("focus", C.c_float * 3),
        ("up", C.c_float * 3),
        ("filename", C.c_char * 256),
        ("max_sample_luminance", C.c_float),
    ]


class RtHostTreeStats(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32),
        ("n_leaves", C.c_uint32),
        ("n_empty_leaves", C.c_uint32),
        ("max_leaf_prims", C.c_uint32),
        ("n_prim_refs", C.c_uint32),
        ("n_leaf_idx", C.c_uint32),
        ("max_depth_param", C.c_int32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}
