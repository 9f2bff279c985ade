#!/bin/bash
# streaming (nt) vs ordinary stores for records / results on THIS box (boxes differ: some show no memory classes at all)
for round in 1 2; do for v in nt1 nt0; do
RIBOPHASE_LIB=$PWD/ribotricer_amd/csrc/variants/$v.so timeout 300 python bench.py --cpu-sample 0 --no-fused --no-pipelined --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; wp=d['config']['workspace_placement']
print('$v kernel %.4f finish %.4f ms/step %.4f first_alloc_ms %.4f search %s penalty %s' % (r['kernel_ms'], r['aux_kernels_ms']['orf_finish'], d['ms_per_step'], d['first_allocation']['ms_per_step'], wp['step_ms'], [round(x,3) for x in r['stream_read']['record_write_penalty']['penalty']]))"
done; done
