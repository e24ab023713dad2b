"""path_tracer_amd -- MI355X (gfx950) replacement for the radiance hot path of
addrianyy/path-tracer: ParallelRenderer::render -> Raytracer::render_fragment ->
trace_pixel -> trace_ray -> BVH / sphere / material / texture / rng.

The product is ``libpt_amd.so`` (C ABI: include/pt_amd.h, kernels:
path_tracer_amd/csrc).  This package is the thin host mirror of the reference's
builder API used by the tests and the benchmark.  There is no CPU fallback.
"""
from ._abi import LIB_PATH, PtError, lib
from .renderer import ParallelRenderer, Raytracer, RendererGroup, group_shard, shard_range
from .scene import (CORNELL_CAMERA, REFERENCE_BOUNDS, RTOW_BOUNDS, Camera, Dielectric, DiffuseLight, Integrator, Lambertian,
                    Metal, PictureTexture, Rect, Scene, SolidTexture, Sphere, Transform, box_rects, cornell_box,
                    random_scene, simple_scene)

__all__ = [
    "LIB_PATH", "PtError", "lib", "ParallelRenderer", "Raytracer", "RendererGroup", "group_shard", "shard_range", "REFERENCE_BOUNDS", "RTOW_BOUNDS",
    "Camera", "Dielectric", "Lambertian", "Metal", "PictureTexture", "Scene", "SolidTexture", "Sphere",
    "random_scene", "simple_scene",
    # extension (not in the reference): Cornell-box primitives, lights, integrator options
    "CORNELL_CAMERA", "DiffuseLight", "Integrator", "Rect", "Transform", "box_rects", "cornell_box",
]
