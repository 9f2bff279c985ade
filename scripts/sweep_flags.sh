#!/bin/bash
# Build the library with each extra flag set in turn and time the tile kernel on the GPU box.
# usage: scripts/sweep_flags.sh "<flags A>" "<flags B>" ...   ("" = the default build)
cd ${GRAFT_REPO_ROOT:-.}
BASE="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -fno-slp-vectorize"
for X in "$@"; do
  rm -f ribotricer_amd/csrc/libribophase.so
  make -C ribotricer_amd/csrc HIPFLAGS="$BASE $X" 2>&1 | grep -E " error"
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile" 2>&1 | tail -2
  timeout 120 python bench.py --algo tile --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('flags=[$X] kernel_ms=%.4f GB/s=%.0f step_ms=%.4f' % (r['kernel_ms'], r['achieved'], d['ms_per_step']))"
done
