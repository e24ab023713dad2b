"""Diagnostic: primary-ray candidate lists against the plain walk on the GPU (pt_set_primary_lists on / off), full frame:
which pixels differ, and in which sample."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import path_tracer_amd as pt

workload = sys.argv[1] if len(sys.argv) > 1 else "largest"
scene_kind, width, height, samples, _ = bench.WORKLOADS[workload]
if len(sys.argv) > 2:
    samples = int(sys.argv[2])
rt = bench.build_raytracer(pt, scene_kind, width, height, samples)
r = pt.ParallelRenderer.new(0)
r.set_lanes_per_pixel(1)
out = {}
for on in (True, False):
    r.set_primary_lists(on)
    _, rad, ids = r.render(rt, seed=bench.RENDER_SEED, radiance=True, hit_ids=True)
    out[on] = rad.copy()
    print("lists", on, "in effect", r.primary_lists(), "kernel ms", r.last_path_kernel_ms(), flush=True)
diff = np.any(out[True].view(np.uint32) != out[False].view(np.uint32), axis=-1)
ys, xs = np.nonzero(diff)
print("pixels that differ between lists on and off:", len(ys), flush=True)
for y, x in list(zip(ys, xs))[:12]:
    print(" pixel", x, y, "block", x // 8, y // 8, "on", out[True][y, x], "off", out[False][y, x], "hit id", ids[y, x], flush=True)
# which sample: single-sample renders of the whole frame, compared at the differing pixels
if len(ys):
    found = 0
    for s in range(samples * samples):
        a = {}
        for on in (True, False):
            r.set_primary_lists(on)
            a[on] = r.render(rt, seed=bench.RENDER_SEED, sample_range=(s, s + 1), radiance=True)[1]
        d = np.any(a[True].view(np.uint32) != a[False].view(np.uint32), axis=-1)
        yy, xx = np.nonzero(d)
        for y, x in zip(yy, xx):
            print(" sample", s, "(sx %d, sy %d)" % (s // samples, s % samples), "pixel", x, y, "on", a[True][y, x], "off", a[False][y, x], flush=True)
            found += 1
        if found >= 12:
            break
r.set_primary_lists(True)
r.close()
