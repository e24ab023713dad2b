"""One-off check (GPU box, ~7 minutes of host CPU): the FULL headline frame -- BASELINE config 3,
RTOW final scene 1920x1080 @ 4096 spp, 8.49 G paths -- rendered by the library against the CPU
oracle's sequential render of the same frame: every fp32 radiance sum, bit for bit.
Too long for the test suite (tests/test_gpu_parity.py::test_full_frames_bit_exact runs the same
frame at 36 spp); the result of the last run is recorded in DESIGN.md.
    python3 tools/full_frame_check.py [workload [samples]]      -> gpurun_out/full_frame_check_<workload>.json
(workload: a key of bench.WORKLOADS; "largest" = config 5, 3840x2160 @ 1024 spp, takes ~9 minutes)
"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import bench  # noqa: E402
import oracle_lib  # noqa: E402   (the checker)
import path_tracer_amd as pt  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "final"
scene_kind, width, height, samples, _desc = bench.WORKLOADS[workload]
if len(sys.argv) > 2:
    samples = int(sys.argv[2])
rt = bench.build_raytracer(pt, scene_kind, width, height, samples)
r = pt.ParallelRenderer.new(0)
t0 = time.time()
img, rad, ids = r.render(rt, seed=bench.RENDER_SEED, radiance=True, hit_ids=True)
t_gpu = time.time() - t0
print("GPU: %.2f s (kernel %.1f ms)" % (t_gpu, r.last_path_kernel_ms()), flush=True)

stop = threading.Event()


def heartbeat():
    while not stop.wait(30.0):
        print("oracle running, %.0f s" % (time.time() - t1), flush=True)


osc = oracle_lib.OracleScene(rt.packed)
if getattr(rt, "integrator", None) is not None:      # extension scenes (cfg 4): bounce cap, background, roulette
    osc.set_integrator(rt.integrator.pod)
t1 = time.time()
threading.Thread(target=heartbeat, daemon=True).start()
r.bind(rt)
lanes = r.lanes_per_pixel(samples, 0, samples * samples)   # small frames: several lanes per pixel, chunk sums added in order
chunk = 0 if lanes == 1 else -(-(samples * samples) // lanes)
ref = osc.render(rt.camera.pod, samples, seed=bench.RENDER_SEED, hit_ids=True, threads=16, chunk=chunk)
stop.set()
t_cpu = time.time() - t1
same = bool(np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32)))
ids_same = bool(np.array_equal(ids, ref["hit_id"]))
img_same = bool(np.array_equal(img, osc.resolve(ref["radiance"], samples)))
diff = int((rad.view(np.uint32) != ref["radiance"].view(np.uint32)).sum())
out = {"workload": workload, "frame": "%dx%d" % (width, height), "samples": samples, "spp": samples * samples,
       "paths": width * height * samples * samples, "lanes_per_pixel": lanes,
       "radiance_bit_exact": same, "differing_words": diff, "hit_ids_equal": ids_same, "image_equal": img_same,
       "gpu_seconds": round(t_gpu, 3), "oracle_seconds_16_threads": round(t_cpu, 1)}
print(json.dumps(out), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_frame_check_%s.json" % workload), "w"))
sys.exit(0 if same and ids_same and img_same else 1)
