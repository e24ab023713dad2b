"""Host-side cost of one pt_render_device call (GPU box): time to enqueue against time to finish.
    python3 tools/host_latency.py      (10 us per call after the first, which loads the code object: 3.3 ms)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import path_tracer_amd as pt
from conftest import make_raytracer
rt = make_raytracer("rtow", 1920, 1080, 8)
r = pt.ParallelRenderer.new(0)
dev = torch.device("cuda", 0)
buf = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device=dev)
s = torch.cuda.Stream(dev)
r.bind(rt); torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter()
    r.render_device(rt, buf.data_ptr(), seed=5, stream=s.cuda_stream)
    t1 = time.perf_counter()
    s.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.3f ms, total %.3f ms, kernel %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, r.last_path_kernel_ms()))
