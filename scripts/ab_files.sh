#!/bin/bash
# A/B of two versions of the tile headers: put them under scripts/_ab/ as rp_tile.hpp.A / .B and
# rp_tile_pipe.hpp.A / .B (A = reference, e.g. from `git show HEAD:...`, B = candidate):
# parity tests + two bench runs each, alternating, on one box.
cd ${GRAFT_REPO_ROOT:-.}
for V in A B A B; do
  for f in rp_tile.hpp rp_tile_pipe.hpp; do cp scripts/_ab/$f.$V ribotricer_amd/csrc/$f; done
  rm -f ribotricer_amd/csrc/libribophase.so
  make -C ribotricer_amd/csrc 2>&1 | grep -E " error"
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
  for rep in 1 2; do
  timeout 120 python bench.py --algo tile --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('version $V kernel_ms=%.4f step_ms=%.4f' % (r['kernel_ms'], d['ms_per_step']))"
  done
done
