#!/bin/bash
# LDS counter pass over bench.py: tools/pmc_lds.sh <tag> [bench args...]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcl_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc" -- python3 $ARGS > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'path_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k,v in sorted(m.items()): print("%-26s %.4g"%(k,v))
if m.get('SQ_LDS_IDX_ACTIVE'):
    print("bank-conflict share of LDS active cycles: %.3f" % (m['SQ_LDS_BANK_CONFLICT']/m['SQ_LDS_IDX_ACTIVE']))
    print("LDS active cycles per LDS instruction: %.2f" % (m['SQ_LDS_IDX_ACTIVE']/m['SQ_INSTS_LDS']))
PY
grep -o 'kernel_ms": [0-9.]*' "$OUT/pmc.log"
