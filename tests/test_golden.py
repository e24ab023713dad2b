"""Committed golden frames (tests/golden/frames.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them bit for bit and the seeded scene / camera inputs have
not drifted.  GPU: the library reproduces them through the C ABI without an oracle run."""
import os

import numpy as np
import pytest

from conftest import make_raytracer

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz"))
CASES = ["simple", "rtow", "reference"]


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def case(name):
    w, h, s, seed = (int(v) for v in GOLD[name + "_meta"])
    return make_raytracer(name, w, h, s), s, seed


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    rt, s, seed = case(name)
    assert np.array_equal(rt.packed.spheres_numpy(), GOLD[name + "_spheres"])          # generator inputs
    assert bits_equal(np.frombuffer(bytes(rt.camera.pod), np.float32, 12), GOLD[name + "_camera"])
    osc = oracle.OracleScene(rt.packed)
    r = osc.render(rt.camera.pod, s, seed=seed, hit_ids=True, hit_t=True)
    assert np.array_equal(r["hit_id"], GOLD[name + "_hit_id"])
    assert bits_equal(r["hit_t"], GOLD[name + "_hit_t"])
    assert bits_equal(r["radiance"], GOLD[name + "_radiance"])
    assert np.array_equal(osc.resolve(r["radiance"], s), GOLD[name + "_rgb8"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_golden(renderer, name):
    rt, s, seed = case(name)
    renderer.set_lanes_per_pixel(1)
    try:
        img, rad, ids = renderer.render(rt, seed=seed, radiance=True, hit_ids=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert np.array_equal(ids, GOLD[name + "_hit_id"])                                  # bit-exact ids
    a, b = img.astype(int), GOLD[name + "_rgb8"].astype(int)
    assert np.mean(np.all(np.abs(a - b) <= 1, axis=-1)) >= 0.999                        # stated tolerance
    assert bits_equal(rad, GOLD[name + "_radiance"])                                    # and in fact bit-exact
    assert np.array_equal(img, GOLD[name + "_rgb8"])
