#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (one run each).
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq1" -- python3 $ARGS > "$OUT/pmc_sq1.log" 2>&1 || { echo "pmc_sq1 failed"; tail -5 "$OUT/pmc_sq1.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d "$OUT/pmc_sq2" -- python3 $ARGS > "$OUT/pmc_sq2.log" 2>&1 || { echo "pmc_sq2 failed"; tail -5 "$OUT/pmc_sq2.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc_fetch failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || { echo "pmc_write failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/pmc_tcp" -- python3 $ARGS > "$OUT/pmc_tcp.log" 2>&1 || { echo "pmc_tcp failed"; tail -5 "$OUT/pmc_tcp.log"; }
find "$OUT" -name "*.csv" | head -40
