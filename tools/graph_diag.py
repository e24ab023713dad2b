"""Diagnostic: what a HIP graph captured from pt_render_device(_static) + pt_resolve_device replays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import path_tracer_amd as pt

CAMERA = dict(eyes=(12.0, 2.0, 3.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=20.0)
scene = pt.Scene.new()
pt.random_scene(scene, 0x5EED5CE7E, pt.RTOW_BOUNDS)
rt = pt.Raytracer.new(pt.Camera.new(width=320, height=184, **CAMERA), scene, 9)
r = pt.ParallelRenderer.new(0)
r.set_lanes_per_pixel(1)
dev = torch.device("cuda", 0)
n = rt.width() * rt.height() * 3
img_ref, rad_ref, _ = r.render(rt, seed=21, radiance=True)
accum = torch.zeros(n, dtype=torch.float32, device=dev)
rgb8 = torch.zeros(n, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(dev)
for static in (True, False):
    r.render_device(rt, accum.data_ptr(), seed=21, stream=side.cuda_stream, static_scratch=static)
    side.synchronize()
    print("static", static, "eager ok", np.array_equal(accum.cpu().numpy().view(np.uint32), rad_ref.reshape(-1).view(np.uint32)), flush=True)
    # 1. resolve alone
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        r.resolve_device(rt, accum.data_ptr(), rgb8.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    rgb8.zero_(); g.replay(); torch.cuda.synchronize()
    print("  resolve-only graph ok", np.array_equal(rgb8.cpu().numpy().reshape(img_ref.shape), img_ref), flush=True)
    # 2. path kernel alone
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side, capture_error_mode="thread_local"):
        cap = torch.cuda.current_stream(dev).cuda_stream
        print("  capture stream", hex(cap), "side", hex(side.cuda_stream), "capturing", torch.cuda.is_current_stream_capturing(), flush=True)
        r.render_device(rt, accum.data_ptr(), seed=21, stream=cap, static_scratch=static)
    print("  after capture: accum untouched by capture?", flush=True)
    accum.fill_(float("nan")); torch.cuda.synchronize()
    g2.replay(); torch.cuda.synchronize()
    a = accum.cpu().numpy()
    print("  path graph: nan words", int(np.isnan(a).sum()), "of", a.size, "equal", np.array_equal(a.view(np.uint32), rad_ref.reshape(-1).view(np.uint32)), flush=True)
    g2.replay(); torch.cuda.synchronize()
    a = accum.cpu().numpy()
    print("  path graph replay 2: nan words", int(np.isnan(a).sum()), "equal", np.array_equal(a.view(np.uint32), rad_ref.reshape(-1).view(np.uint32)), flush=True)
    # 3. path + resolve in one graph (the test's shape)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=side, capture_error_mode="thread_local"):
        cap = torch.cuda.current_stream(dev).cuda_stream
        r.render_device(rt, accum.data_ptr(), seed=21, stream=cap, static_scratch=static)
        r.resolve_device(rt, accum.data_ptr(), rgb8.data_ptr(), stream=cap)
    for k in range(2):
        accum.fill_(float("nan")); rgb8.zero_(); torch.cuda.synchronize()
        g3.replay(); torch.cuda.synchronize()
        a = accum.cpu().numpy()
        print("  path+resolve graph replay", k, ": nan words", int(np.isnan(a).sum()), "equal", np.array_equal(a.view(np.uint32), rad_ref.reshape(-1).view(np.uint32)),
              "image ok", np.array_equal(rgb8.cpu().numpy().reshape(img_ref.shape), img_ref), flush=True)
    del g, g2, g3
r.close()
