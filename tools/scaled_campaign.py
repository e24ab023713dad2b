"""One-off campaign (GPU box): tests/test_gpu_parity.py::_scaled_case at every power-of-two scene size from
2^-42 to 2^32 x 12 scene seeds, library against oracle bit for bit (the edges of the REGULAR class of
pt_device.hpp).  Last run: 900 cases, none mismatching.
    python3 tools/scaled_campaign.py
"""
import os, sys, json
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, path_tracer_amd as pt
from test_gpu_parity import _scaled_case
r = pt.ParallelRenderer.new(0)
bad = []; n = 0
for l in range(-42, 33):
    for k in range(12):
        rt, s = _scaled_case(2.0 ** l, k)
        osc = oracle_lib.OracleScene(rt.packed)
        ref = osc.render(rt.camera.pod, s, seed=5, hit_ids=True)
        r.set_lanes_per_pixel(1)
        img, rad, ids = r.render(rt, seed=5, radiance=True, hit_ids=True)
        r.set_lanes_per_pixel(0)
        same = np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32)) or \
            (np.isnan(rad) == np.isnan(ref["radiance"])).all() and np.array_equal(rad[~np.isnan(rad)].view(np.uint32), ref["radiance"][~np.isnan(rad)].view(np.uint32))
        if not (same and np.array_equal(ids, ref["hit_id"])): bad.append((l, k))
        osc.close(); n += 1
print(json.dumps({"cases": n, "mismatching": bad}))
