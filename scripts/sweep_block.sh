#!/bin/bash
# Sweep (threads per workgroup, kRun, kTile) of the tile kernel on the GPU box.
# usage: bash scripts/sweep_block.sh "256:15:7936 512:15:15872"
cd ${GRAFT_REPO_ROOT:-.}
for x in $1; do
  B=$(echo $x | cut -d: -f1); K=$(echo $x | cut -d: -f2); T=$(echo $x | cut -d: -f3)
  sed -i "s/^constexpr int kTileBlock = [0-9]*;/constexpr int kTileBlock = $B;/; s/^constexpr int kTile = [0-9]*;/constexpr int kTile = $T;/; s/^constexpr int kRun = [0-9]*;/constexpr int kRun = $K;/" ribotricer_amd/csrc/rp_tile.hpp
  make -C ribotricer_amd/csrc 2>&1 | grep -E " error"
  timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile and (synthetic or ragged or edge)" 2>&1 | tail -1
  timeout 120 python bench.py --algo tile --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('B=$B K=$K TILE=$T  kernel_ms=%.4f  GB/s=%.0f  frac=%.3f  step_ms=%.4f' % (r['kernel_ms'], r['achieved'], r['frac'], r['step_device_ms']))"
done
