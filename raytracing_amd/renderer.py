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
    """Like the reference's main(): the scene path comes from stdin when it is not given as an argument."""
    import sys
    argv = sys.argv[1:] if argv is None else argv
    path = argv[0] if argv else input("Input the scene file: ").strip()
    scene = load_scene_json(path)
    r = renderer_for(scene)
    r.preprocess(scene)
    film = r.render(scene)
    out = film.write_image_to_file(os.path.join(os.getcwd(), scene.config.filename))
    print("wrote", out)


__all__ = ["Film", "SamplerRenderer", "PathRenderer", "WhittedRenderer", "renderer_for", "main",
           "RT_RNG_COUNTER", "RT_RNG_REFERENCE_TILE_STREAM"]
