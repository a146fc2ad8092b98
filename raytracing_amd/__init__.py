"""MI355X-native hot path of the Neuroglial/RayTracing renderer: KD-tree traversal, ray/triangle
intersection and the Path / Whitted bounce loop as HIP kernels behind a C ABI."""
from .scene import FlatScene, RenderConfig, build_scene, camera_matrices, camera_ray, develop, load_scene_json  # noqa: F401
