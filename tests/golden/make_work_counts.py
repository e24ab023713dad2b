"""Generates path_tracer_amd/data/work_counts.json: the algorithmic work per path of every
bench workload, from the CPU oracle's instrumented counters.

The counts are properties of the REFERENCE algorithm (its BVH, its fixed left-first
traversal): roofline.achieved in bench.py = flops_per_path x paths / kernel seconds,
independent of how many nodes the GPU's own traversal visits.  Full frames at reduced
samples per pixel (stated per entry): per-path work does not depend on spp, the grid is
deterministic.  Run from the repository root (CPU only).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402  (workload table + scene construction only)
import oracle_lib  # noqa: E402
import path_tracer_amd as pt  # noqa: E402

# usage: make_work_counts.py [workload ...]   (no argument: all; named ones are merged into the existing file)
PATH = os.path.join(ROOT, "path_tracer_amd", "data", "work_counts.json")
only = sys.argv[1:]
out = json.load(open(PATH)) if (only and os.path.exists(PATH)) else {}
for name, (scene_kind, w, h, samples, desc) in bench.WORKLOADS.items():
    if only and name not in only:
        continue
    s = 2
    while (s + 1) ** 2 * w * h <= 20e6 and s + 1 <= samples:
        s += 1
    rt = bench.build_raytracer(pt, scene_kind, w, h, s)
    osc = oracle_lib.OracleScene(rt.packed)
    if getattr(rt, "integrator", None) is not None:
        osc.set_integrator(rt.integrator.pod)
    c = osc.render(rt.camera.pod, s, seed=bench.RENDER_SEED, counters=True)["counters"]
    flops = oracle_lib.algorithmic_flops(c)
    out[name] = {
        "description": desc, "measured_at": "%dx%d, samples=%d (%d spp)" % (w, h, s, s * s),
        "spheres": rt.packed.n_spheres, "flops_per_path": flops / c["paths"],
        "segments_per_path": c["segments"] / c["paths"], "aabb_tests_per_segment": c["aabb_tests"] / c["segments"],
        "sphere_tests_per_segment": c["sphere_tests"] / c["segments"],
        "rng_draws_per_path": c["rng_draws"] / c["paths"], "counters": c, "flop_model": oracle_lib.FLOPS,
    }
    print(name, out[name]["measured_at"], "flops/path = %.1f" % out[name]["flops_per_path"], flush=True)
with open(PATH, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
