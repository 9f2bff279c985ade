#!/bin/bash
# A/B of the finish kernel's workgroup size (variants built with -DRP_FINISH_BLOCK=64/128/256/512 into csrc/variants/fb*.so):
# one process per variant and round, finish time from the library's HIP events, verify on.
for round in 1 2; do for v in fb64 fb128 fb256 fb512; do for spec in "cfg3:11000000" "gencode_short:11000000"; do cfg=${spec%%:*}; n=${spec#*:}
RIBOPHASE_LIB=$PWD/ribotricer_amd/csrc/variants/$v.so timeout 300 python bench.py --cfg $cfg --orfs $n --cpu-sample 0 --no-fused --no-pipelined --no-tune-workspace --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v $cfg', 'ms/step %.4f kernel %.4f finish+rewalk %.4f verify %s' % (d['ms_per_step'], r['kernel_ms'], r['aux_kernels_ms']['orf_finish'], d['verify']['ok']))"
done; done; done
