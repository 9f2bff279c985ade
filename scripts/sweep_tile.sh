#!/bin/bash
# Sweep (kRun, kTile) of the tile kernel on the GPU box: rebuild + bench each.
# usage: bash scripts/sweep_tile.sh "9:6144 15:10240 21:14336"
cd ${GRAFT_REPO_ROOT:-.}
for kt in $1; do
  K=${kt%%:*}; T=${kt##*:}
  sed -i "s/^constexpr int kTile = [0-9]*;/constexpr int kTile = $T;/; s/^constexpr int kRun = [0-9]*;/constexpr int kRun = $K;/" ribotricer_amd/csrc/rp_tile.hpp
  make -C ribotricer_amd/csrc 2>&1 | grep -E "error" 
  python bench.py --algo tile --steps 20 --warmup 3 --cpu-sample 0 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('K=$K TILE=$T  kernel_ms=%.4f  GB/s=%.0f  frac=%.3f  step_ms=%.4f  aux=%s' % (r['kernel_ms'], r['achieved'], r['frac'], r['step_device_ms'], r['aux_kernels_ms']))"
done
