#!/bin/bash
# Quick SQ counter pass over bench.py: tools/pmc_quick.sh <tag> [bench args...]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d "$OUT/pmc" -- python3 $ARGS > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg=collections.defaultdict(list); dur=[]
for f in glob.glob(sys.argv[1]+'/pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'path_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k,v in sorted(m.items()): print("%-26s %.4g"%(k,v))
if 'SQ_THREAD_CYCLES_VALU' in m:
    print("lane-active fraction (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)): %.3f" % (m['SQ_THREAD_CYCLES_VALU']/(m['SQ_ACTIVE_INST_VALU']*64)))
    print("wait fraction of wave cycles (WAIT_ANY/WAVE_CYCLES): %.3f" % (m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']))
PY
grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*' "$OUT/pmc.log"
