#!/usr/bin/env python3
"""Phase timeline of rp::k_tile_score from a -DRP_STAMPS build (RIBOPHASE_LIB=...): every 64th
workgroup stores its s_memtime stamps; printed: mean ticks from entry to each stamp, per wave,
and the tick rate (span of all stamps over the HIP-event duration of the launch)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ribotricer_amd import _lib
from ribotricer_amd.engine import PhaseScoreEngine, make_filter
from ribotricer_amd.synth import synth_csr_device

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
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
tiles = (counts.numel() + 7935) // 7936
used = min(SL, (tiles + 63) // 64)
b = buf[:used].astype(np.int64)
span = int(b[:, :, :8].max() - b[:, :, 0].min())
print(f"{cfg} n={n} tiles={tiles} sampled={used}; launch {t[0][1]:.4f} ms, stamp span {span} ticks -> {span / t[0][1] / 1e3:.1f} ticks/us")
names = ["issued", "own-landed", "at-barrier1", "barrier1", "runs-done", "barrier2", "end"]
rel = b[:, :, 1:8] - b[:, :, 0:1]
print("wave " + " ".join(f"{x:>11}" for x in names))
for w in range(4):
    print(f"{w:4d} " + " ".join(f"{rel[:, w, k].mean():11.0f}" for k in range(7)))
print("p90  " + " ".join(f"{np.percentile(rel[:, 0, k], 90):11.0f}" for k in range(7)))
