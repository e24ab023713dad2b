"""Generates tests/golden/frames.npz: small frames rendered by the CPU oracle
(oracle/pt_oracle.c, parity build) -- primary hit ids, fp32 radiance sums, RGB8 image --
plus the scene dump each was rendered from.  The reference holds no fixtures of its own
 and cannot be run here, so these vectors pin the ORACLE against
regressions and give the GPU tests data that needs no oracle run; they are not outputs
of the Rust binary.  Run from the repository root (CPU only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from conftest import RENDER_SEED, make_raytracer  # noqa: E402

CASES = {"simple": ("simple", 64, 36, 4), "rtow": ("rtow", 64, 36, 4), "reference": ("reference", 48, 27, 2)}
out = {}
for name, (kind, w, h, s) in CASES.items():
    rt = make_raytracer(kind, w, h, s)
    osc = oracle_lib.OracleScene(rt.packed)
    r = osc.render(rt.camera.pod, s, seed=RENDER_SEED, hit_ids=True, hit_t=True)
    out[name + "_meta"] = np.array([w, h, s, RENDER_SEED], np.int64)
    out[name + "_hit_id"] = r["hit_id"]
    out[name + "_hit_t"] = r["hit_t"]
    out[name + "_radiance"] = r["radiance"]
    out[name + "_rgb8"] = osc.resolve(r["radiance"], s)
    out[name + "_spheres"] = rt.packed.spheres_numpy()
    out[name + "_camera"] = np.frombuffer(bytes(rt.camera.pod), np.float32, 12).copy()
    print(name, w, h, s, "hit fraction %.3f" % (r["hit_id"] >= 0).mean())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames.npz"), **out)
