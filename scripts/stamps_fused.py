#!/usr/bin/env python3
"""Phase timeline of the FUSED rp::k_tile_score<true> from a -DRP_STAMPS build
(RIBOPHASE_LIB=...): usage: python scripts/stamps_fused.py [exon_len] [exons_per_orf] [gap]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ribotricer_amd import _lib
from ribotricer_amd.engine import get_engine, make_filter
from ribotricer_amd.gather import GatherPlan, IntervalTable

exon_len = int(sys.argv[1]) if len(sys.argv) > 1 else 120
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gap = int(sys.argv[3]) if len(sys.argv) > 3 else 0
total = 300_000_000
n_iv = total // exon_len // k * k
n = n_iv // k
t = IntervalTable(np.arange(n_iv, dtype=np.int64) * (exon_len + gap), np.full(n_iv, exon_len, np.int32),
                  np.arange(n + 1, dtype=np.int64) * k, (np.arange(n) % 2).astype(np.uint8),
                  np.arange(n + 1, dtype=np.int64) * (exon_len * k))
cov_len = int(t.iv_start[-1] + exon_len)
cov = torch.poisson(torch.full((cov_len,), 0.3, device="cuda")).to(torch.int32)
plan = GatherPlan(t, cov_len)
eng = get_engine("cuda:0")
lib = _lib.load()
SL = 4096
buf = np.zeros((SL, 4, 8), np.uint64)
for fused in (True, False):
    counts = plan.gather(cov)
    run = (lambda tm=None: eng.score_coverage(cov, plan, thresholds=make_filter(), reuse_outputs=True, timings=tm)) if fused else \
          (lambda tm=None: eng.score(counts, plan.offsets, thresholds=make_filter(), algo="tile", reuse_outputs=True, timings=tm))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    tm = []
    run(tm)
    torch.cuda.synchronize()
    assert lib.rp_debug_read_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
    tiles = (int(t.offsets[-1]) + 7935) // 7936
    used = min(SL, (tiles + 63) // 64)
    b = buf[:used].astype(np.int64)
    print(f"{'fused' if fused else 'plain'}: exon {exon_len} x {k}, gap {gap}; tiles={tiles}; launch {tm[0][1]:.4f} ms")
    names = ["issued", "own-landed", "at-barrier1", "barrier1", "runs-done", "barrier2", "end"]
    rel = b[:, :, 1:8] - b[:, :, 0:1]
    print("wave " + " ".join(f"{x:>11}" for x in names))
    for w in range(4):
        print(f"{w:4d} " + " ".join(f"{rel[:, w, kk].mean():11.0f}" for kk in range(7)))
