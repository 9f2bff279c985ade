#!/bin/bash
# What would smaller segment records buy the scoring kernel?  Timing experiments (results wrong): only the first 2 / 1 of the
# three 16-byte planes stored (32 / 16 bytes per record instead of 48), no record stores at all -- against the product,
# alternating, K copies of the counts x K engines per process (scripts/placement_grid.py).  usage: bash scripts/ab_record_bytes.sh [rounds]
cd ${GRAFT_REPO_ROOT:-.}
for r in $(seq 1 ${1:-2}); do
for V in ${VARIANTS:-product planes2 planes1 nostore}; do
  if [ $V = product ]; then unset RIBOPHASE_LIB; else export RIBOPHASE_LIB=$PWD/variants/$V.so; fi
  timeout 600 python3 scripts/placement_grid.py 11000000 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-8s kernel min %.3f median %.3f max %.3f   finish %s' % ('$V', d['min'], d['median'], d['max'], d['finish_ms_min_median_max']))"
done
done
