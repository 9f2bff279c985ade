#!/usr/bin/env python3
"""Phase timeline of rp::k_tile_score_ring from a -DRP_STAMPS build (RIBOPHASE_LIB=...): round 8 of
every workgroup stores its s_memtime stamps; printed: mean ticks from barrier 1 of that round to
each stamp, per wave (wave 3 = the loader), and how far apart the arrivals at barrier 1 are.
usage: python scripts/stamps_ring.py [cfg] [n_orfs]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ribotricer_amd import _lib
from ribotricer_amd.engine import PhaseScoreEngine, make_filter
from ribotricer_amd.synth import synth_csr_device

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
eng = PhaseScoreEngine("cuda:0")
counts, offsets = synth_csr_device(n, cfg=cfg, device="cuda:0")
lib = _lib.load()
SL = 4096
buf = np.zeros((SL, 4, 8), np.uint64)
for _ in range(5):
    eng.score(counts, offsets, thresholds=make_filter(), algo="tile", reuse_outputs=True)
torch.cuda.synchronize()
t = []
eng.score(counts, offsets, thresholds=make_filter(), algo="tile", reuse_outputs=True, timings=t)
torch.cuda.synchronize()
assert lib.rp_debug_read_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
grid = int((buf[:, 0, 0] != 0).sum())
b = buf[:grid].astype(np.int64)
tiles = (counts.numel() + 7935) // 7936
print(f"{cfg} n={n} tiles={tiles} workgroups={grid} ({tiles / max(1, grid):.0f} tiles each); launch {t[0][1]:.4f} ms "
      f"-> {t[0][1] * 1e3 / (tiles / max(1, grid)):.3f} us per tile and workgroup")
names = ["next-issued", "mapped", "runs-done", "barrier2", "next-landed", "records"]
rel = b[:, :, 1:7] - b[:, :, 0:1]
print("wave " + " ".join(f"{x:>12}" for x in names) + "   arrive-b1(prev round end -> b1)")
for w in range(4):
    wait_b1 = b[:, w, 0] - b[:, w, 7]
    print(f"{w:4d} " + " ".join(f"{rel[:, w, k].mean():12.0f}" for k in range(6)) + f"   {wait_b1.mean():10.0f}")
