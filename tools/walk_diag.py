"""PT_WALK_VIEW against PT_WALK_REFERENCE on the GPU: pixels whose radiance sums differ.
usage (GPU box): python tools/walk_diag.py [kind w h s]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import path_tracer_amd as pt
from conftest import make_raytracer, RENDER_SEED

kind = sys.argv[1] if len(sys.argv) > 1 else "rtow"
w, h, s = (int(sys.argv[i]) for i in (2, 3, 4)) if len(sys.argv) > 4 else (1920, 1080, 6)
r = pt.ParallelRenderer.new(0)
seed = RENDER_SEED
if kind.startswith("scaled:"):
    from test_gpu_parity import _scaled_case
    rt, s = _scaled_case(2.0 ** int(kind.split(":")[1]))
    seed = 5
elif kind.startswith("fuzz:"):
    from test_gpu_parity import _fuzz_case
    rt, s = _fuzz_case(int(kind.split(":")[1]))
    seed = int(kind.split(":")[1])
else:
    rt = make_raytracer(kind, w, h, s)
r.set_lanes_per_pixel(1)
out = {}
for view in (False, True):
    r.set_walk_order(view)
    img, rad, ids = r.render(rt, seed=seed, radiance=True, hit_ids=True)
    out[view] = (rad.copy(), ids.copy(), r.last_path_kernel_ms())
    print("view" if view else "reference", "order in effect", r.walk_order(), "kernel ms %.3f" % out[view][2], flush=True)
d = (out[False][0].view(np.uint32) != out[True][0].view(np.uint32)).any(axis=2)
print("pixels differing:", int(d.sum()), "of", d.size, " hit ids differing:", int((out[False][1] != out[True][1]).sum()))
ys, xs = np.nonzero(d)
for x, y in list(zip(xs, ys))[:40]:
    print("x %d y %d  reference %s  view %s" % (x, y, out[False][0][y, x], out[True][0][y, x]))
r.close()
