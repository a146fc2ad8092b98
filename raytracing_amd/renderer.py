"""Host-side mirror of the reference's renderer interface for the device path.

The reference's `main` builds a scene and a Renderer from the scene file and calls
`renderer->preprocess(*scene); renderer->render(*scene);` (reference src/RayTracing.cpp:26-33;
Renderer interface src/core/Render/Render.h:20-21; PathRenderer / WhittedRenderer constructors
Render.cpp:337-340, 457-471). The classes below keep those names, arguments and the same division of work:
preprocess() prepares light sampling (here: uploads the flat scene, which builds the light distribution),
render() runs the sample loop on the GPU and writes Film "Filename".
"""
import os

import numpy as np

from ._ctypes_defs import RT_INTEGRATOR_PATH, RT_INTEGRATOR_WHITTED, RT_RNG_COUNTER, RT_RNG_REFERENCE_TILE_STREAM
from .device import Context
from .pngio import write_png
from .scene import FlatScene, develop, load_scene_json


class Film:
    """rgb_sum / weight_sum as FilmTile::addSample accumulates them (reference src/core/Object/Film.h:111-155)."""

    def __init__(self, rgb_sum, weight_sum, scale=1.0, filename="rendered.png", context=None):
        self.rgb_sum, self.weight_sum, self.scale, self.filename = rgb_sum, weight_sum, scale, filename
        self._ctx = context

    def to_rgb8(self):
        """8-bit sRGB image: on the device when the film came from a renderer, else through the host restatement."""
        if self._ctx is not None:
            return self._ctx.develop(self.rgb_sum, self.weight_sum, self.scale)
        return develop(self.rgb_sum, self.weight_sum, self.scale)

    def write_image_to_file(self, path=None):   # Film::writeImageToFile (reference src/core/Object/Film.cpp:116-167)
        path = path or self.filename
        write_png(path, self.to_rgb8())
        return path


class SamplerRenderer:
    integrator = None

    def __init__(self, max_depth=None, config=None, spp=None, device_id=0, rng_mode=None, seed=0, context=None):
        self.max_depth, self.config, self.spp = max_depth, config, spp
        self.rng_mode, self.seed = rng_mode, seed
        self.ctx = context if context is not None else Context(device_id)
        self._scene = None

    def preprocess(self, scene: FlatScene):
        self.ctx.upload_scene(scene)
        cfg = self.config if self.config is not None else scene.config
        if cfg is None:
            raise ValueError("no render configuration: pass config= or load the scene from a scene file")
        self.ctx.set_camera(cfg)
        self.config = cfg
        self._scene = scene

    def render(self, scene: FlatScene, spp_begin=0, spp_end=None, write=False):
        if self._scene is not scene:
            self.preprocess(scene)
        cfg = self.config
        rgb, w = self.ctx.render(integrator=self.integrator, spp=self.spp if self.spp is not None else cfg.spp,
                                 max_depth=self.max_depth if self.max_depth is not None else cfg.max_depth,
                                 rng_mode=self.rng_mode, seed=self.seed, spp_begin=spp_begin, spp_end=spp_end)
        film = Film(rgb, w, cfg.film_scale, cfg.filename, self.ctx)
        if write:
            film.write_image_to_file()
        return film


class PathRenderer(SamplerRenderer):
    """PathRenderer (reference src/core/Render/Render.cpp:320-449): rrThreshold and the light strategy are fixed
    to the reference's values (1, uniform)."""
    integrator = RT_INTEGRATOR_PATH


class WhittedRenderer(SamplerRenderer):
    """WhittedRenderer (reference src/core/Render/Render.cpp:457-532). Defaults to the reference's per-tile random
    stream so images are comparable pixel by pixel."""
    integrator = RT_INTEGRATOR_WHITTED


def renderer_for(scene: FlatScene, **kw):
    """ObjectFactory::createInstance for "Renderer": {"Type": "Path" | "Whitted"}."""
    cls = PathRenderer if scene.config.integrator == RT_INTEGRATOR_PATH else WhittedRenderer
    return cls(**kw)


def main(argv=None):
    """Like the reference's main() (src/RayTracing.cpp:13-37): scene file in, Film "Filename" out; the scene path comes
    from stdin when it is not given as an argument."""
    import argparse
    import time
    ap = argparse.ArgumentParser(prog="python -m raytracing_amd", description=main.__doc__)
    ap.add_argument("scene", nargs="?", help="scene JSON of the reference's format")
    ap.add_argument("-o", "--output", help="PNG path (default: the film's Filename in the current directory)")
    ap.add_argument("--spp", type=int, help="samples per pixel (default: the sampler's SPP)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0, help="seed of the counter streams")
    ap.add_argument("--reference-stream", action="store_true",
                    help="draw from the CPU renderer's per-tile PCG32 streams (reproduces its image; serial per tile, so "
                         "seconds instead of milliseconds for the Path integrator). Whitted scenes use it by default.")
    a = ap.parse_args(argv)
    path = a.scene if a.scene else input("Input the scene file: ").strip()
    scene = load_scene_json(path)
    r = renderer_for(scene, spp=a.spp, device_id=a.device, seed=a.seed,
                     rng_mode=RT_RNG_REFERENCE_TILE_STREAM if a.reference_stream else None)
    r.preprocess(scene)
    t0 = time.perf_counter()
    film = r.render(scene)
    dt = time.perf_counter() - t0
    out = film.write_image_to_file(a.output or os.path.join(os.getcwd(), scene.config.filename))
    print(f"rendered in {dt:.3f} s, wrote {out}")


__all__ = ["Film", "SamplerRenderer", "PathRenderer", "WhittedRenderer", "renderer_for", "main",
           "RT_RNG_COUNTER", "RT_RNG_REFERENCE_TILE_STREAM"]
