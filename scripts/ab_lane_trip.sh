#!/bin/bash
# A/B of the lane-per-segment limit (csrc/variants/lt*.so built with -DRP_LANE_TRIP=22/30/44/50): shapes + a short-heavy set
for round in 1 2; do for v in lt22 lt30 lt44 lt50; do
echo "== $v round $round"
RIBOPHASE_LIB=$PWD/ribotricer_amd/csrc/variants/$v.so timeout 300 python scripts/bench_shapes.py 2>/dev/null | head -4 | cut -c1-110
RIBOPHASE_LIB=$PWD/ribotricer_amd/csrc/variants/$v.so timeout 300 python bench.py --cfg gencode_short --cpu-sample 0 --no-fused --no-pipelined --no-tune-workspace --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('gencode_short kernel %.4f finish %.4f verify %s' % (r['kernel_ms'], r['aux_kernels_ms']['orf_finish'], d['verify']['ok']))"
done; done
