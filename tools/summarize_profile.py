"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile.sh) into profiles/<name>_*:
the rocprofv3 kernel-stats CSV as is, and one JSON with the per-launch averages of every
PMC counter of the path kernel plus the derived figures DESIGN.md quotes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, note = sys.argv[1], sys.argv[2], sys.argv[3]
kernel = sys.argv[4] if len(sys.argv) > 4 else "path_kernel"   # substring of the kernel's name (path_kernel_fast: the PT_FAST build)
# gpurun merges every call's files into the same local directory: the newest file of each pass is this run's
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
stats = newest(src + "/trace/*/*_kernel_stats.csv")
shutil.copy(stats, dst + "_kernel_stats.csv")
agg = collections.defaultdict(list)
for d in glob.glob(src + "/pmc_*/"):
    f = newest(d + "*/*_counter_collection.csv")
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
avg_ns = None
for r in csv.DictReader(open(stats)):
    if kernel in r["Name"]:
        avg_ns = float(r["AverageNs"])
d = {}
if "SQ_THREAD_CYCLES_VALU" in m:
    d["lane_active_fraction"] = m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64)
if "SQ_WAIT_ANY" in m:
    d["wave_wait_fraction"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
if "TCC_HIT_sum" in m:
    d["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) reads half the bytes of a wide coalesced read on
    # gfx950 -> doubled; WRITE_SIZE (KiB) is exact for 16-B-per-lane stores.
    d["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
if "GRBM_GUI_ACTIVE" in m and avg_ns:
    d["effective_clock_ghz"] = m["GRBM_GUI_ACTIVE"] / 8 / avg_ns
json.dump({"note": note, "path_kernel_avg_ns": avg_ns, "pmc_avg_per_launch": m, "derived": d},
          open(dst + "_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(d, indent=1), avg_ns)
