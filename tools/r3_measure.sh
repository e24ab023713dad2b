#!/bin/bash
# Round 3's measurements of the final build, one GPU call: the driver's own command, every workload with its CPU leg,
# then rocprofv3 trace + counter passes of four workloads (tools/profile.sh).  Results under gpurun_out/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_driver.json 2> gpurun_out/r3_driver.err || { echo "driver command failed"; exit 1; }
for w in final earth cpu largest cornell; do
  timeout -k 10 300 python3 bench.py --workload $w > gpurun_out/r3_final_$w.json 2> gpurun_out/r3_final_$w.err || { echo "bench $w failed"; exit 1; }
  echo "bench $w done"
done
for w in final largest cornell earth; do
  bash tools/profile.sh r3_$w --workload $w --no-fast --steps 2 --warmup 1 > gpurun_out/r3_profile_$w.log 2>&1 || { echo "profile $w failed"; tail -3 gpurun_out/r3_profile_$w.log; exit 1; }
  echo "profile $w done"
done
bash tools/profile.sh r3_fast --mode fast --no-fast --steps 2 --warmup 1 > gpurun_out/r3_profile_fast.log 2>&1 || echo "profile fast failed"
if [ -f build/variants/stats.so ]; then
  PT_AMD_LIB=$REPO/build/variants/stats.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-fast --steps 1 --warmup 0 > gpurun_out/r3_stats.json 2> gpurun_out/r3_stats.err || echo "stats run failed"
fi
echo all done
