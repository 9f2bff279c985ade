#!/bin/bash
# Parity tests + timing of the persistent (register-prefetch) kernel next to the tile kernel.
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipe" 2>&1 | tail -5
for A in tile pipe; do
  for C in cfg2 ${EXTRA_CFGS}; do
  timeout 300 python bench.py --algo $A --cfg $C --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$A $C kernel_ms=%.4f GB/s=%.0f step_ms=%.4f' % (r['kernel_ms'], r['achieved'], d['ms_per_step']), r.get('aux_kernels_ms'))"
  done
done
