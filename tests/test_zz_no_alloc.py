"""pt_render_device_static: the entry point that allocates nothing (include/pt_amd.h) against
pt_render_device, through the C ABI, bit for bit, on two calls in a row -- the second with a smaller
input (smaller frame, fewer samples), so that what the first call left in the library's scratch
array must not leak into the second."""
import numpy as np
import pytest

import path_tracer_amd as pt
from path_tracer_amd import _abi

pytestmark = pytest.mark.gpu

CAMERA = dict(eyes=(12.0, 2.0, 3.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=20.0)


def _render_both(renderer, rt, torch, seed, sample_range=None):
    dev = torch.device("cuda", 0)
    n = rt.width() * rt.height() * 3
    stream = torch.cuda.Stream(dev)
    out = []
    with torch.cuda.stream(stream):
        for static in (False, True):
            buf = torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
            renderer.render_device(rt, buf.data_ptr(), seed=seed, sample_range=sample_range, stream=stream.cuda_stream,
                                   static_scratch=static)
            assert renderer.last_path_kernel_ms() >= 0.0      # synchronises; raises on a kernel-side error
            out.append(buf.cpu().numpy())
    stream.synchronize()
    return out


def test_static_scratch_entry_point_matches_render_device():
    torch = pytest.importorskip("torch")
    scene = pt.Scene.new()
    pt.random_scene(scene, 0x5EED5CE7E, pt.RTOW_BOUNDS)
    renderer = pt.ParallelRenderer.new(0)
    renderer.set_lanes_per_pixel(1)          # one lane per pixel: the launch is cut into pieces, which is what uses the scratch
    try:
        # first call: 640 x 360 @ 144 spp -> several pieces per tile (per-tile progress words in use)
        big = pt.Raytracer.new(pt.Camera.new(width=640, height=360, **CAMERA), scene, 12)
        ref, got = _render_both(renderer, big, torch, seed=11)
        assert np.isfinite(ref).all()
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
        # second call, smaller input: 200 x 120, a sub-range of a 8 x 8 grid
        small = pt.Raytracer.new(pt.Camera.new(width=200, height=120, **CAMERA), scene, 8)
        ref2, got2 = _render_both(renderer, small, torch, seed=12, sample_range=(3, 61))
        assert np.isfinite(ref2).all()
        assert np.array_equal(ref2.view(np.uint32), got2.view(np.uint32))
        # and the static entry point twice in a row agrees with itself
        _, again = _render_both(renderer, small, torch, seed=12, sample_range=(3, 61))
        assert np.array_equal(again.view(np.uint32), got2.view(np.uint32))
    finally:
        renderer.close()


def test_static_scratch_rejects_frames_beyond_its_array():
    renderer = pt.ParallelRenderer.new(0)
    try:
        scene = pt.Scene.new()
        pt.simple_scene(scene)
        # 16384 x 16384 = 2^28 pixels = 4.2 M tiles: beyond the 262,144-tile array (and nothing is rendered)
        rt = pt.Raytracer.new(pt.Camera.new(width=16384, height=16384, **CAMERA), scene, 2)
        renderer.bind(rt)
        with pytest.raises(_abi.PtError) as err:
            renderer.render_device(rt, 0x1000, static_scratch=True)
        assert err.value.code == -5          # PT_ERR_LIMIT
    finally:
        renderer.close()
