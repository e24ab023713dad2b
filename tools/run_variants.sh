#!/bin/bash
# Runs bench.py per variant in build/variants (GPU box), two passes in opposite order, and
# prints the best kernel time of each (the first run on a box is ~3 % slow whatever it is).
# usage: tools/run_variants.sh [bench args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
names=$(ls $REPO/build/variants/*.so | xargs -n1 basename | sed 's/\.so$//')
rev=$(echo "$names" | tac)
for n in $names $rev; do
  r=$(PT_AMD_LIB=$REPO/build/variants/$n.so timeout -k 10 120 python3 $REPO/bench.py --no-cpu-baseline "$@" 2>/dev/null | grep -o 'kernel_ms": [0-9.]*' | grep -o '[0-9.]*$')
  echo "$n ${r:-failed}"
  [ -z "$r" ] && break   # a failed run may have been a GPU fault: nothing more runs in this call
done | python3 -c '
import sys
best = {}
for line in sys.stdin:
    n, r = line.split()
    if r != "failed": best[n] = min(best.get(n, 1e30), float(r))
    else: best.setdefault(n, 1e30)
for n in sorted(best): print("%s kernel_ms %.3f" % (n, best[n]))
'
