#!/usr/bin/env python3
"""One-off stress of the short-ORF paths: random indexes built around the switch points of the tile kernel -- 64 / 65
segments per tile (head-row path vs rounds), 22 / 23 triplets (lane-per-segment rounds vs 64-slot rounds), 256 / 257
segments per tile (one vs two lane rounds), tiles cut through ORFs, every 16-byte phase of the counts array -- CSR scorer
against the C oracle (integers exact, phase <= 1e-6, ties bit for bit).  usage: stress_lane_rounds.py [cases]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from helpers import assert_matches_oracle  # noqa: E402
from ribotricer_amd.engine import PhaseScoreEngine, make_filter  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = PhaseScoreEngine("cuda:0")
rng = np.random.default_rng(4242)
total_orfs = 0
for case in range(cases):
    kind = case % 6
    n = int(rng.integers(2000, 9000))
    if kind == 0:    # right at 64 / 65 segments per 6 144-position tile: lengths around 96
        lens = rng.integers(90, 100, size=n)
    elif kind == 1:  # around 22 / 23 triplets
        lens = rng.choice([63, 64, 65, 66, 67, 68, 69, 70, 71], size=n)
    elif kind == 2:  # around 256 segments per tile: lengths around 24
        lens = rng.integers(20, 28, size=n * 3)
    elif kind == 3:  # mostly tiny with a few long ones (tiles that fall back, multi-tile ORFs)
        lens = rng.integers(1, 67, size=n * 2)
        lens[rng.integers(0, lens.size, size=5)] = rng.integers(5000, 20000, size=5)
    elif kind == 4:  # geometric tail from the CLI minimum (a realistic short-heavy law)
        lens = 60 + 3 * rng.geometric(3 / 64, size=n)
    else:            # everything mixed, empty ORFs included
        lens = rng.choice([0, 1, 2, 3, 59, 60, 61, 66, 69, 96, 97, 150, 400, 3000], size=n)
    lens = lens.astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offsets[-1])
    lam = np.repeat(rng.choice([0.005, 0.02, 0.2, 2.0], size=lens.size), lens)
    counts = rng.poisson(lam).astype(np.int32)
    mis = case % 4
    dev = torch.zeros(total + 8, dtype=torch.int32, device="cuda:0")
    view = dev[mis : mis + total]
    view.copy_(torch.from_numpy(counts))
    res = eng.score_host(view, offsets, thresholds=make_filter(), algo="tile")
    assert_matches_oracle(res, counts, offsets)
    total_orfs += lens.size
print(f"stress_lane_rounds: {cases} random indexes, {total_orfs} ORFs, all equal to the oracle")
