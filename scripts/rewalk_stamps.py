#!/usr/bin/env python3
"""Where does a re-walk's time go?  A library built with -DRP_REWALK_STAMPS (variants/rewalk_stamps.so) sums s_memtime deltas
per phase of rp_tile.hpp::finish_unsafe -- float64 walk | reductions + frame scores + state machine | tie replay -- separately
for ties and non-ties.  (The tick of s_memtime is not calibrated here and the stamps' own atomics slow the pass down: read the RATIOS.)
usage: RIBOPHASE_LIB=variants/rewalk_stamps.so python scripts/rewalk_stamps.py [n_orfs] [cfg]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ribotricer_amd import _lib  # noqa: E402
from ribotricer_amd.engine import PhaseScoreEngine, make_filter  # noqa: E402
from ribotricer_amd.synth import synth_csr_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
eng = PhaseScoreEngine("cuda:0")
counts, offsets = synth_csr_device(n, cfg=cfg, device="cuda:0")
thr = make_filter()
plan = eng.plan_for(offsets, counts.numel(), (counts.data_ptr() // 4) % 4)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
for _ in range(5):
    eng.score(counts, offsets, thresholds=thr, algo="tile", reuse_outputs=True, plan=plan)
torch.cuda.synchronize()
lib.rp_debug_rewalk_stamps(out, 1)
out8 = (ctypes.c_ulonglong * 8)()
lib.rp_debug_replay_stamps(out8, 1)
reps = 20
tm = []
for _ in range(reps):
    eng.score(counts, offsets, thresholds=thr, algo="tile", reuse_outputs=True, plan=plan, timings=tm)
torch.cuda.synchronize()
lib.rp_debug_rewalk_stamps(out, 1)
v = np.array(list(out), np.float64)
tick_us = 0.01  # (printed as hundreds of s_memtime ticks: the tick is not calibrated on this part -- compare the phases, not the microseconds)
print(f"{cfg} {n} ORFs: finish pass median {np.median([t[2] for t in tm]) * 1e3:.1f} us")
for name, b in (("not a tie", 0), ("exact tie", 8)):
    c = v[b]
    if c == 0:
        continue
    print(f"  {name}: {c / reps:.0f} per launch, mean length {v[b + 4] / c:.0f} nt; per re-walk: float64 walk {v[b + 1] / c * tick_us:.2f}, "
          f"reductions + scores {v[b + 2] / c * tick_us:.2f}, replay {v[b + 3] / c * tick_us:.2f}  (sum {(v[b + 1] + v[b + 2] + v[b + 3]) / c * tick_us:.2f}; unit: 100 s_memtime ticks)")
lib.rp_debug_replay_stamps(out8, 1)
r = np.array(list(out8), np.float64)
if r[0]:
    print(f"  inside the replay ({r[0] / reps:.0f} per launch, {r[4] / r[0]:.1f} non-zero codons in frame 0 on average): counts + table rows + terms into LDS "
          f"{r[1] / r[0] * tick_us:.2f}, serial fold {r[2] / r[0] * tick_us:.2f}, scores + state machine {r[3] / r[0] * tick_us:.2f}  (unit: 100 s_memtime ticks)")
