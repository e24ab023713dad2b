"""One-off fuzz campaign (GPU box): the random scenes of tests/test_gpu_parity.py::_fuzz_case for a
range of seeds far beyond the 24 the suite runs, library against oracle, bit for bit, in both lane
modes.  Prints a heartbeat and a JSON summary (also written to gpurun_out/fuzz_campaign.json).
    python3 tools/fuzz_campaign.py [first [last]]
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
from test_gpu_parity import _fuzz_case  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 24
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
r = pt.ParallelRenderer.new(0)
bad = []
paths = 0
view_order = 0        # cases the library walked nearer-child-first (PT_WALK_VIEW in effect)
with_lists = 0        # cases whose camera rays tested candidate lists instead of walking (pt_get_primary_lists)
t0 = time.time()
for k in range(first, last):
    rt, samples = _fuzz_case(k)
    osc = oracle_lib.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, samples, seed=k, hit_ids=True)
    r.set_lanes_per_pixel(1)
    img, rad, ids = r.render(rt, seed=k, radiance=True, hit_ids=True)
    r.set_lanes_per_pixel(0)
    view_order += r.walk_order()
    with_lists += int(r.primary_lists())
    ok = np.array_equal(ids, ref["hit_id"]) and np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32))
    r.bind(rt)
    lanes = r.lanes_per_pixel(samples, 0, samples * samples)
    chunk = -(-samples * samples // lanes)
    rad2 = r.render(rt, seed=k, radiance=True)[1]
    ref2 = osc.render(rt.camera.pod, samples, seed=k, chunk=chunk)["radiance"]
    ok = ok and np.array_equal(rad2.view(np.uint32), ref2.view(np.uint32))
    paths += 2 * rt.pixel_count() * samples * samples
    if not ok:
        bad.append(k)
    osc.close()
    if k % 50 == 0:
        print("case %d, %d mismatching so far, %.0f s" % (k, len(bad), time.time() - t0), flush=True)
out = {"cases": last - first, "first": first, "last": last, "paths": int(paths), "mismatching_cases": bad,
       "cases_walked_in_view_order": int(view_order), "cases_with_primary_lists": int(with_lists)}
print(json.dumps(out), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fuzz_campaign.json"), "w"))
sys.exit(1 if bad else 0)
