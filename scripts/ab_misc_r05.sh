#!/bin/bash
# round 5, one call: (1) the strided-write probe, (2) k_rewalk_long with 64 instead of 512 workgroups in the two-stream mode,
# (3) the fused finish pass forced to 5 waves per SIMD
cd ${GRAFT_REPO_ROOT:-.}
bash scripts/box_kind.sh 2>/dev/null | tail -2
echo "== strided writes"; timeout 600 python3 scripts/probe_rw_stride.py
echo "== k_rewalk_long grid (two-stream mode)"
for r in 1 2; do for V in product lg64; do
  if [ $V = product ]; then unset RIBOPHASE_LIB; else export RIBOPHASE_LIB=$PWD/variants/$V.so; fi
  timeout 600 python3 bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-fused --no-verify --no-slice-projection 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pipelined']; print('$V value %.4e step %.4f | two streams %.4f one stream %.4f' % (d['value'], d['ms_per_step'], p['two_streams_ms_per_step'], p['one_stream_ms_per_step']))"
done; done
unset RIBOPHASE_LIB
echo "== fused finish at 5 waves per SIMD"
python3 scripts/ab_variants.py product=ribotricer_amd/csrc/libribophase.so fw5=variants/fw5.so --cfgs cfg3:11000000 --rounds 2 --calls 20 --fused
