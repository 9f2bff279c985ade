#!/bin/bash
# Build the library with each extra flag set in turn; parity-test and time the persistent kernel.
# usage: [BPCS="3 4"] scripts/sweep_pipe.sh "<flags A>" "<flags B>" ...   ("" = the default build)
cd ${GRAFT_REPO_ROOT:-.}
BASE="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -fno-slp-vectorize"
for X in "$@"; do
  rm -f ribotricer_amd/csrc/libribophase.so
  make -C ribotricer_amd/csrc HIPFLAGS="$BASE $X" 2>&1 | grep -E " error"
  make -C ribotricer_amd/csrc resources HIPFLAGS="$BASE $X" 2>&1 | grep -A12 "k_tile_score_pipe" | grep -E "VGPRs:|ScratchSize|LDS Size" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '
  echo
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipe" 2>&1 | tail -1
  for BPC in ${BPCS:-0}; do
  RP_PIPE_BLOCKS_PER_CU=$BPC timeout 120 python bench.py --algo pipe --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('flags=[$X] bpc=$BPC kernel_ms=%.4f GB/s=%.0f step_ms=%.4f' % (r['kernel_ms'], r['achieved'], d['ms_per_step']))"
  done
done
