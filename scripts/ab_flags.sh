#!/bin/bash
# A/B on one box: for each flag set build, run the tile parity tests, the precision check and the bench.
cd ${GRAFT_REPO_ROOT:-.}
BASE="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -fno-slp-vectorize"
for X in "$@"; do
  rm -f ribotricer_amd/csrc/libribophase.so
  make -C ribotricer_amd/csrc HIPFLAGS="$BASE $X" 2>&1 | grep -E " error"
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile" 2>&1 | tail -1
  timeout 300 python tests/tools/check_precision.py 2>&1 | tail -4
  timeout 120 python bench.py --algo tile --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('flags=[$X] kernel_ms=%.4f step_ms=%.4f' % (r['kernel_ms'], d['ms_per_step']), r.get('aux_kernels_ms'))"
done
