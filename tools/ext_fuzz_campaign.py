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


from test_gpu_ext import _ext_fuzz_case as ext_case  # noqa: E402


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
