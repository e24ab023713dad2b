import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The reference's own camera (src/main.rs:80-87)
MAIN_RS_CAMERA = dict(eyes=(12.0, 2.0, 3.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=20.0)

SCENE_SEED = 0x5EED5CE7E      # seed of the seeded random_scene used by tests and bench
RENDER_SEED = 20260711        # seed of the per-(pixel,sample) streams


def make_raytracer(kind, width, height, samples, scene_seed=SCENE_SEED):
    """kind: 'simple' (generators.rs:8-23), 'rtow' (random_scene, RTOW bounds),
    'reference' (random_scene with the reference's bounds, ~10.7k spheres)."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    if kind == "simple":
        pt.simple_scene(scene)
    elif kind == "rtow":
        pt.random_scene(scene, scene_seed, pt.RTOW_BOUNDS)
    elif kind == "reference":
        pt.random_scene(scene, scene_seed, pt.REFERENCE_BOUNDS)
    else:
        raise ValueError(kind)
    camera = pt.Camera.new(width=width, height=height, **MAIN_RS_CAMERA)
    return pt.Raytracer.new(camera, scene, samples)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def renderer():
    """One GPU context for the whole session (gpu tests only)."""
    import path_tracer_amd as pt
    r = pt.ParallelRenderer.new(0)
    yield r
    r.close()
