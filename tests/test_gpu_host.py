"""The compiled, non-Python host: host/main.cpp plays the reference's src/main.rs (:73-122) on top of the C ABI
-- builder API, Raytracer::new (BVH), ParallelRenderer, progress lines, output.png -- as a fresh process of its
own.  Its image must be the oracle's, byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from conftest import RENDER_SEED, make_raytracer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h,s", [(160, 90, 3), (1024, 576, 8)], ids=["small-frame-one-pass", "one-lane-per-pixel-8-passes"])
def test_cpp_host_renders_the_oracles_image(renderer, oracle, tmp_path, w, h, s):
    from PIL import Image
    subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True, capture_output=True)   # no-op when built (build())
    exe = os.path.join(ROOT, "host", "pt_main")
    run = subprocess.run([exe, "rtow", str(w), str(h), str(s)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    assert "Constructing BVH for" in run.stdout and "Progress: 100.0%" in run.stdout       # scene/mod.rs:51-59, main.rs:41-71
    img = np.asarray(Image.open(tmp_path / "output.png").convert("RGB"))
    assert img.shape == (h, w, 3)
    # the same scene, camera and seed through the oracle (host/main.cpp: random_scene with the RTOW bounds, main.rs:80-87 camera)
    rt = make_raytracer("rtow", w, h, s)
    osc = oracle.OracleScene(rt.packed)
    renderer.bind(rt)
    lanes = renderer.lanes_per_pixel(s, 0, s * s)
    if s * s >= 64:
        assert lanes == 1           # pt_render_progress continues sequential sums across its passes only then
    ref = osc.render(rt.camera.pod, s, seed=RENDER_SEED, chunk=-(-s * s // lanes))["radiance"]
    assert np.array_equal(img, osc.resolve(ref, s))
