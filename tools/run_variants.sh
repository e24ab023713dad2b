#!/bin/bash
# Runs bench.py once per variant in build/variants (GPU box). usage: tools/run_variants.sh [bench args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for so in $REPO/build/variants/*.so; do
  n=$(basename $so .so)
  r=$(PT_AMD_LIB=$so timeout -k 10 120 python3 $REPO/bench.py --no-cpu-baseline "$@" 2>/dev/null | grep -o 'kernel_ms": [0-9.]*')
  echo "$n $r"
done
