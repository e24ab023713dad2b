"""PT_FAST against PT_EXACT on the GPU: how many pixels differ after tonemapping, by how much, at several spp.
usage (GPU box): python tools/fast_diag.py [kind w h]"""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import path_tracer_amd as pt
from conftest import make_raytracer, RENDER_SEED

kind = sys.argv[1] if len(sys.argv) > 1 else "rtow"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 320
h = int(sys.argv[3]) if len(sys.argv) > 3 else 180
r = pt.ParallelRenderer.new(0)
for s in (3, 8, 10, 16, 32, 64):
    rt = make_raytracer(kind, w, h, s)
    r.set_mode(False)
    img0, rad0, _ = r.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
    r.set_mode(True)
    img1, rad1, _ = r.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
    ms = r.last_path_kernel_ms()
    r.set_mode(False)
    d = np.abs(img0.astype(int) - img1.astype(int)).max(axis=2)
    t0, t1 = np.sqrt(rad0 / (s * s)), np.sqrt(rad1 / (s * s))
    print("%s %dx%d spp %5d: within1 %.5f within2 %.5f max %3d  rmse %.2e  identical words %.4f  finite %s  fast kernel %.2f ms" % (
        kind, w, h, s * s, (d <= 1).mean(), (d <= 2).mean(), d.max(), float(np.sqrt(np.mean((t0 - t1) ** 2))),
        (rad0.view(np.uint32) == rad1.view(np.uint32)).mean(), np.isfinite(rad1).all(), ms), flush=True)
r.close()
