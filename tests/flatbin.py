"""Flat binary files for tests/abi_harness.c (the C caller of the ABI): a FlatScene + render job, and a ray batch."""
import struct

import numpy as np


def write_scene(path, scene, cfg, integrator, spp, max_depth, rng_mode, seed):
    has_n = scene.tri_n is not None
    with open(path, "wb") as f:
        f.write(struct.pack("<7I", 0x53465452, scene.n_nodes, len(scene.leaf_idx), scene.n_prims, len(scene.materials), len(scene.lights), int(has_n)))
        f.write(struct.pack("<10i", cfg.width, cfg.height, *cfg.sample_bounds, integrator, spp, max_depth, rng_mode))
        f.write(struct.pack("<Q", seed))
        f.write(np.asarray(scene.bounds, np.float32).tobytes())
        f.write(np.asarray(cfg.raster_to_camera, np.float32).tobytes())
        f.write(np.asarray(cfg.camera_to_world, np.float32).tobytes())
        f.write(np.ascontiguousarray(scene.nodes, np.uint32).tobytes())
        f.write(np.ascontiguousarray(scene.leaf_idx, np.int32).tobytes())
        f.write(np.ascontiguousarray(scene.tri_p, np.float32).tobytes())
        if has_n:
            f.write(np.ascontiguousarray(scene.tri_n, np.float32).tobytes())
        f.write(np.ascontiguousarray(scene.material_id, np.int32).tobytes())
        f.write(np.ascontiguousarray(scene.light_id, np.int32).tobytes())
        f.write(np.ascontiguousarray(scene.materials).tobytes())
        f.write(np.ascontiguousarray(scene.lights).tobytes())


def write_rays(path, o, d):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(o)))
        f.write(np.ascontiguousarray(o, np.float32).tobytes())
        f.write(np.ascontiguousarray(d, np.float32).tobytes())


def read_outputs(path, n, width, height):
    raw = np.fromfile(path, np.uint8)
    npix = width * height
    at = 0

    def take(count, dtype):
        nonlocal at
        size = count * np.dtype(dtype).itemsize
        a = raw[at:at + size].view(dtype)
        at += size
        return a

    prim, t, bary, occ = take(n, np.int32), take(n, np.float32), take(3 * n, np.float32).reshape(n, 3), take(n, np.uint8)
    rgb, w, rgb8 = take(3 * npix, np.float32).reshape(height, width, 3), take(npix, np.float32).reshape(height, width), take(3 * npix, np.uint8).reshape(height, width, 3)
    assert at == raw.size
    return prim, t, bary, occ, rgb, w, rgb8
