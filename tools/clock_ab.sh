#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in persist base persist base; do
  OUT=$REPO/gpurun_out/clk_$v
  rm -rf $OUT; mkdir -p $OUT
  PT_AMD_LIB=$REPO/build/variants/$v.so timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --no-cpu-baseline --workload final --steps 2 --warmup 1 > $OUT/log 2>&1
  python3 - $OUT $v <<'PY'
import csv,glob,sys
g=[];d={}
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'path_kernel' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE': g.append((r['Dispatch_Id'],float(r['Counter_Value'])))
for f in glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'path_kernel' in r['Kernel_Name']: d[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in g: print(sys.argv[2], 'cycles %.4g'%v, 'ns', d.get(k), 'GHz %.3f'%(v/d[k]/ (8 if v/d[k]>10 else 1)) if k in d else '')
PY
done
