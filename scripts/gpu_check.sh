#!/bin/bash
# Full GPU parity suite + a short bench line per algo (run on the GPU box).
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for A in ${ALGOS:-tile pipe}; do
  for C in ${CFGS:-cfg2}; do
  timeout 300 python bench.py --algo $A --cfg $C --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$A $C kernel_ms=%.4f GB/s=%.0f step_ms=%.4f value=%.3e' % (r['kernel_ms'], r['achieved'], d['ms_per_step'], d['value']), r.get('aux_kernels_ms'), d.get('quality'))"
  done
done
