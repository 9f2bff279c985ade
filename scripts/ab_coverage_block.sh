#!/bin/bash
# The bench's fused sections over compact coverages of 64-, 8- and 1-position blocks (and the dense layout), alternating,
# one box: kernel ms, fraction of peak, coverage positions, chunks per tile.  -> profiles/archive/r04_ab_coverage_block.txt
for rep in 1 2; do
  for flag in "--coverage-block 64" "--coverage-block 1" "--coverage-block 8" "--dense-coverage"; do
    timeout 400 python bench.py --cpu-sample 0 --no-pipelined $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('RES [$flag]', 'plain %.4f' % r['kernel_ms'], {k:(round(d[k]['kernel_ms'],4), round(d[k]['frac'],3), d[k]['coverage_positions'], d[k]['verify']['ok'], d[k]['coverage_map_build_ms'] and round(d[k]['coverage_map_build_ms'],1), round(d[k]['gather_plan_build_ms'],1), round(d[k]['gather_plan']['chunks_per_tile_mean'],1), d[k]['gather_plan']['slow_tiles']) for k in ('fused','fused_nested')})"
  done
done
