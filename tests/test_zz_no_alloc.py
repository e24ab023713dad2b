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


def test_static_scratch_is_guarded_per_gpu_not_per_context():
    """All contexts of a GPU share the library's scratch array: a static launch on another stream -- from ANOTHER
    context, or from the same context with a non-static launch in between -- while the array's previous launch is
    still running is refused (PT_ERR_INVALID), as the same misuse of pt_render_device is; once the first launch has
    finished the second goes through, and the first frame is untouched."""
    torch = pytest.importorskip("torch")
    scene = pt.Scene.new()
    pt.random_scene(scene, 0x5EED5CE7E, pt.RTOW_BOUNDS)
    rt = pt.Raytracer.new(pt.Camera.new(width=1280, height=720, **CAMERA), scene, 16)    # tens of milliseconds of kernel time
    small = pt.Raytracer.new(pt.Camera.new(width=64, height=48, **CAMERA), scene, 2)
    first, second = pt.ParallelRenderer.new(0), pt.ParallelRenderer.new(0)
    try:
        dev = torch.device("cuda", 0)
        n = rt.width() * rt.height() * 3
        s1, s2, s3 = (torch.cuda.Stream(dev) for _ in range(3))
        buf_a = torch.zeros(n, dtype=torch.float32, device=dev)
        buf_b = torch.zeros(n, dtype=torch.float32, device=dev)
        buf_c = torch.zeros(small.width() * small.height() * 3, dtype=torch.float32, device=dev)
        first.bind(rt)
        second.bind(rt)
        ref = first.render(rt, seed=5, radiance=True)[1]
        torch.cuda.synchronize()
        # another context, another stream
        first.render_device(rt, buf_a.data_ptr(), seed=5, stream=s1.cuda_stream, static_scratch=True)
        with pytest.raises(_abi.PtError) as err:
            second.render_device(rt, buf_b.data_ptr(), seed=5, stream=s2.cuda_stream, static_scratch=True)
        assert err.value.code == -1
        # the same context: static on s1 (running), its own scratch on s2 (allowed: another scratch), static on s3 (refused)
        first.render_device(rt, buf_b.data_ptr(), seed=5, stream=s2.cuda_stream)
        with pytest.raises(_abi.PtError) as err:
            first.render_device(rt, buf_b.data_ptr(), seed=5, stream=s3.cuda_stream, static_scratch=True)
        assert err.value.code == -1
        s1.synchronize()
        s2.synchronize()
        second.render_device(rt, buf_b.data_ptr(), seed=5, stream=s2.cuda_stream, static_scratch=True)   # now it goes through
        s2.synchronize()
        a, b = buf_a.cpu().numpy(), buf_b.cpu().numpy()
        assert np.array_equal(a.view(np.uint32), ref.reshape(-1).view(np.uint32))
        assert np.array_equal(b.view(np.uint32), ref.reshape(-1).view(np.uint32))
        del buf_c, small
    finally:
        first.close()
        second.close()


def test_static_entry_point_under_hip_graph_capture():
    """The callers the entry point is for: pt_render_device_static + pt_resolve_device captured into a HIP graph (the
    capturing stream records a memset node and two kernel nodes; the library allocates nothing, records no event and
    queries nothing while capturing), the graph replayed twice -- each replay's radiance and image equal pt_render's
    bit for bit."""
    torch = pytest.importorskip("torch")
    scene = pt.Scene.new()
    pt.random_scene(scene, 0x5EED5CE7E, pt.RTOW_BOUNDS)
    rt = pt.Raytracer.new(pt.Camera.new(width=320, height=184, **CAMERA), scene, 9)
    renderer = pt.ParallelRenderer.new(0)
    try:
        renderer.set_lanes_per_pixel(1)
        dev = torch.device("cuda", 0)
        n = rt.width() * rt.height() * 3
        img_ref, rad_ref, _ = renderer.render(rt, seed=21, radiance=True)
        accum = torch.zeros(n, dtype=torch.float32, device=dev)
        rgb8 = torch.zeros(n, dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream(dev)
        # one un-captured launch first: it fills the launch's occupancy / attribute cache
        renderer.render_device(rt, accum.data_ptr(), seed=21, stream=side.cuda_stream, static_scratch=True)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            cap = torch.cuda.current_stream(dev).cuda_stream
            renderer.render_device(rt, accum.data_ptr(), seed=21, stream=cap, static_scratch=True)
            renderer.resolve_device(rt, accum.data_ptr(), rgb8.data_ptr(), stream=cap)
        assert renderer.last_path_kernel_ms() == -1.0           # a captured launch is not timed
        for _ in range(2):
            accum.fill_(float("nan"))
            rgb8.zero_()
            torch.cuda.synchronize()        # the fills are done before the graph starts (the graph's nodes were captured on a non-blocking stream)
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(accum.cpu().numpy().view(np.uint32), rad_ref.reshape(-1).view(np.uint32))
            assert np.array_equal(rgb8.cpu().numpy().reshape(img_ref.shape), img_ref)
        del graph
    finally:
        renderer.close()
