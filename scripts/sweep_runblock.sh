#!/bin/bash
# Sweep the unroll block of the lane run (kRunBlock) on the GPU box.
cd ${GRAFT_REPO_ROOT:-.}
for RB in $1; do
  sed -i "s/^constexpr int kRunBlock = [0-9]*;/constexpr int kRunBlock = $RB;/" ribotricer_amd/csrc/rp_tile.hpp
  make -C ribotricer_amd/csrc 2>&1 | grep -E " error"
  make -C ribotricer_amd/csrc resources 2>&1 | grep -A12 "k_tile_scoreE" | grep -E "VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '
  timeout 120 python bench.py --algo tile --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('RB=$RB kernel_ms=%.4f GB/s=%.0f' % (r['kernel_ms'], r['achieved']))"
done
