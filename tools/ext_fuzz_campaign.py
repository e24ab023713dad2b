"""One-off fuzz campaign of the EXTENSION (GPU box): random scenes of rects on all three axes (plain, flipped),
boxes behind Translate(RotateY(..)) instances, spheres of every material, lights, random integrator options
(bounce cap, sky or constant background, Russian roulette) and cameras inside and outside the geometry --
library against the C oracle (this repository's specification of the extension), hit ids
and fp32 radiance bit for bit, in both lane modes.  Prints a heartbeat and a JSON summary (also written to
gpurun_out/ext_fuzz_campaign.json).
    python3 tools/ext_fuzz_campaign.py [first [last]]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle_lib  # noqa: E402   (the checker)
import path_tracer_amd as pt  # noqa: E402


def material(rnd):
    m = int(rnd.integers(0, 5))
    if m <= 1:
        return pt.Lambertian.new_solid(tuple(float(x) for x in rnd.uniform(0.05, 0.95, 3)))
    if m == 2:
        return pt.Metal.new(tuple(float(x) for x in rnd.uniform(0.3, 1.0, 3)), float(rnd.choice([0.0, 0.3, 1.0])))
    if m == 3:
        return pt.Dielectric.new(float(rnd.choice([1.5, 0.7, 2.4])))
    return pt.DiffuseLight.new(tuple(float(x) for x in rnd.uniform(0.5, 8.0, 3)))


def ext_case(k):
    rnd = np.random.default_rng(500000 + k)
    sc = pt.Scene.new()
    n_rects = int(rnd.choice([1, 2, 4, 8, 14]))
    for _ in range(n_rects):
        axis = int(rnd.integers(0, 3))
        a0, b0 = (float(x) for x in rnd.uniform(-3.0, 2.0, 2))
        a1, b1 = a0 + float(rnd.uniform(0.2, 4.0)), b0 + float(rnd.uniform(0.2, 4.0))
        sc.add(pt.Rect(axis, a0, a1, b0, b1, float(rnd.uniform(-3.0, 3.0)), material(rnd), flip=bool(rnd.integers(0, 2))))
    for _ in range(int(rnd.choice([0, 1, 2, 3]))):
        tr = pt.Transform.new(tuple(float(x) for x in rnd.uniform(-2.5, 2.5, 3)), float(rnd.choice([0.0, 15.0, -18.0, 45.0, 90.0, 133.0, -270.0])))
        size = tuple(float(x) for x in rnd.uniform(0.3, 2.0, 3))
        for r in pt.box_rects((0.0, 0.0, 0.0), size, material(rnd), tr if rnd.integers(0, 4) else None):
            sc.add(r)
    for _ in range(int(rnd.choice([0, 1, 3, 9]))):
        sc.add(pt.Sphere.new(tuple(float(x) for x in rnd.uniform(-3.0, 3.0, 3)), float(rnd.choice([0.1, 0.5, 1.5, 30.0], p=[0.3, 0.4, 0.25, 0.05])),
                             material(rnd)))
    eye = rnd.uniform(-5.0, 5.0, 3)
    target = rnd.uniform(-1.0, 1.0, 3)
    if k % 7 == 0:                                          # axis-aligned view: zero direction components
        eye, target = np.array([0.0, 0.0, 6.0]), np.array([0.0, 0.0, 0.0])
    fov = float(rnd.choice([10.0, 40.0, 90.0, 150.0]))
    w, h = [(33, 17), (64, 36), (48, 48), (24, 64)][k % 4]
    cam = pt.Camera.new(tuple(map(float, eye)), tuple(map(float, target)), (0.0, 1.0, 0.0), fov, w, h)
    kind = k % 4
    if kind == 0:
        integ = None
    elif kind == 1:
        integ = pt.Integrator(max_bounces=int(rnd.choice([1, 3, 12, 50])), background=tuple(float(x) for x in rnd.uniform(0.0, 0.5, 3)))
    elif kind == 2:
        integ = pt.Integrator(max_bounces=int(rnd.choice([4, 20])), rr_start=int(rnd.choice([0, 2, 5])))
    else:
        integ = pt.Integrator(max_bounces=50, background=(0.0, 0.0, 0.0), rr_start=int(rnd.choice([-1, 3])))
    samples = int(rnd.choice([2, 3, 5]))
    return pt.Raytracer.new(cam, sc, samples, integrator=integ), samples


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    oracle_lib.load()
    r = pt.ParallelRenderer.new(0)
    bad, paths, t0 = [], 0, time.time()
    for k in range(first, last):
        rt, s = ext_case(k)
        osc = oracle_lib.OracleScene(rt.packed)
        if rt.integrator is not None:
            osc.set_integrator(rt.integrator.pod)
        for lanes in (1, 0):
            r.set_lanes_per_pixel(lanes)
            used = r.lanes_per_pixel(s, 0, s * s) if lanes != 1 else 1
            _img, rad, ids = r.render(rt, seed=k, radiance=True, hit_ids=True)
            chunk = 0 if lanes == 1 else -(-(s * s) // used)
            ref = osc.render(rt.camera.pod, s, seed=k, hit_ids=True, chunk=chunk)
            ok = np.array_equal(ids, ref["hit_id"]) and np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32))
            if not ok:
                bad.append([k, lanes, int(np.sum(ids != ref["hit_id"])), int(np.sum(rad.view(np.uint32) != ref["radiance"].view(np.uint32)))])
            paths += rt.width() * rt.height() * s * s
        if (k - first) % 200 == 199:
            print("case %d, %d mismatching so far, %.0f s" % (k + 1, len(bad), time.time() - t0), flush=True)
    r.set_lanes_per_pixel(0)
    r.close()
    out = {"cases": 2 * (last - first), "first": first, "last": last, "paths": paths, "mismatching_cases": bad[:50], "n_mismatching": len(bad)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ext_fuzz_campaign.json"), "w"))
    print(json.dumps(out))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
