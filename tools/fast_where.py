"""Where PT_FAST differs from PT_EXACT by more than 1/255: primary-hit material and silhouette statistics."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import path_tracer_amd as pt
from conftest import make_raytracer, RENDER_SEED
w, h, s = 640, 360, 32
r = pt.ParallelRenderer.new(0)
rt = make_raytracer("rtow", w, h, s)
img0, rad0, ids = r.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
r.set_mode(True)
img1, rad1, _ = r.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
d = np.abs(img0.astype(int) - img1.astype(int)).max(axis=2)
bad = d >= 2
kinds = np.array([m.kind for m in rt.packed.materials])
mat = np.array([sp.material for sp in rt.packed.spheres])
prim_kind = np.where(ids >= 0, kinds[mat[np.clip(ids, 0, None)]], -1)
edge = np.zeros_like(bad)
edge[1:-1, 1:-1] = (ids[1:-1, 1:-1] != ids[:-2, 1:-1]) | (ids[1:-1, 1:-1] != ids[2:, 1:-1]) | (ids[1:-1, 1:-1] != ids[1:-1, :-2]) | (ids[1:-1, 1:-1] != ids[1:-1, 2:])
print("bad pixels", bad.sum(), "of", bad.size, " on silhouettes:", (bad & edge).sum(), " silhouette share of frame: %.3f" % edge.mean())
for k, name in ((-1, "sky"), (0, "lambertian"), (1, "metal"), (2, "dielectric")):
    sel = prim_kind == k
    print("%-10s pixels %7d  bad %5d  (%.4f)" % (name, sel.sum(), (bad & sel).sum(), (bad & sel).sum() / max(1, sel.sum())))
rel = (rad1 - rad0) / np.maximum(rad0, 1e-6)
print("mean relative radiance difference (fast - exact): all %.2e  bad pixels %.2e" % (rel.mean(), rel[bad].mean()))
for k, name in ((0, "lambertian"), (1, "metal"), (2, "dielectric")):
    sel = prim_kind == k
    print("  %-10s mean rel diff %.3e  mean |rel diff| %.3e" % (name, rel[sel].mean(), np.abs(rel[sel]).mean()))
ys, xs = np.nonzero(bad)
print("sample of bad pixels (x, y, id, diff):", [(int(x), int(y), int(ids[y, x]), int(d[y, x])) for x, y in list(zip(xs, ys))[:12]])
