#!/bin/bash
# Memory-pipe counter pass over bench.py: tools/pmc_mem.sh <tag> [bench args...]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcm_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -- python3 $ARGS > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'path_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k,v in sorted(m.items()): print("%-36s %.4g"%(k,v))
cyc = m.get('GRBM_GUI_ACTIVE',0)/8
print("kernel cycles ~ %.4g; per-CU busy fractions (sum / 256 CUs / cycles):" % cyc)
for k in ('TA_TA_BUSY_sum','TD_TD_BUSY_sum','TCP_GATE_EN1_sum','TCP_TCP_TA_DATA_STALL_CYCLES_sum','TCP_PENDING_STALL_CYCLES_sum','TCP_TOTAL_CACHE_ACCESSES_sum'):
    if k in m: print("  %-36s %.3f" % (k, m[k]/256/cyc))
PY
grep -o 'kernel_ms": [0-9.]*' "$OUT/pmc.log"
