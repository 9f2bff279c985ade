#!/bin/bash
# k_orf_finish as persistent waves (RP_FINISH_PERSIST = workgroups per resident slot; 0 = the one-shot grid of rounds 1-4):
# finish times of the CSR path, the fused path on the exon layout, and the 1/8 slice, alternating.  usage: bash scripts/ab_finish_persist.sh [rounds]
cd ${GRAFT_REPO_ROOT:-.}
for r in $(seq 1 ${1:-2}); do
for P in 0 1 2 4; do
  RP_FINISH_PERSIST=$P timeout 600 python3 bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-pipelined --no-fused-nested --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f=d['fused']; s=d['slice_projection']['slices']
print('persist=$P cfg3 kernel %.4f finish %.4f step %.4f | fused kernel %.4f finish %.4f | slice8 step %.4f finish %.4f eff %.3f | slice2 finish %.4f' % (
  r['kernel_ms'], r['aux_kernels_ms']['orf_finish'], d['ms_per_step'], f['kernel_ms'], f['finish_ms'], s[2]['step_ms'], s[2]['finish_ms'], s[2]['projected_efficiency'], s[0]['finish_ms']))"
done
done
