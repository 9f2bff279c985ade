# k_orf_finish with its too-close-to-call ORFs re-walked on the spot (ship: rounds 2-3), handed to k_orf_rewalk through the
# re-walk lists (queue: round 4), and skipped (norewalk: RP_EXPERIMENT_NO_REWALK, results wrong: the floor), cfg3 (11 M ORFs) and cfg2 (1 M);
# orf_finish = everything between the scoring kernel and the end of the step.
for r in 1 2; do for v in ${VARIANTS:-ship queue norewalk}; do for cfg in cfg3 cfg2 gencode_short; do RIBOPHASE_LIB=$PWD/ribotricer_amd/csrc/variants/$v.so timeout 300 python bench.py --cfg $cfg --cpu-sample 0 --no-pipelined --no-fused-nested --no-verify --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; f=d.get('fused') or {}
print('$v $cfg', 'kernel %.4f' % r['kernel_ms'], 'finish %.4f' % r['aux_kernels_ms']['orf_finish'], 'step %.4f' % d['ms_per_step'], 'fused kernel %.4f finish %.4f' % (f.get('kernel_ms', 0), f.get('finish_ms', 0)))"; done; done; done
