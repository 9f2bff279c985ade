"""Ad-hoc longer run of tests/test_gpu_fuzz.py's generator (GPU box): seeds [a, b)."""
import sys
import numpy as np
import torch

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import assert_matches_oracle
from test_gpu_fuzz import TILES, lengths_for
from ribotricer_amd.engine import PhaseScoreEngine

a, b = int(sys.argv[1]), int(sys.argv[2])
eng = PhaseScoreEngine("cuda:0")
for seed in range(a, b):
    for algo in ("tile",):
        rng = np.random.default_rng(1000 + seed)
        lens = lengths_for(seed, TILES[algo])
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        total = int(offsets[-1])
        lam = rng.choice([0.02, 0.3, 3.0], size=lens.size)
        counts = rng.poisson(np.repeat(lam, lens)).astype(np.int32)
        mis = seed % 4
        dev = torch.zeros(total + 8, dtype=torch.int32, device="cuda:0")
        view = dev[mis : mis + total]
        view.copy_(torch.from_numpy(counts))
        r = eng.score(view, torch.from_numpy(offsets).cuda(), algo=algo)
        torch.cuda.synchronize()
        assert_matches_oracle(r.cpu_numpy(), counts, offsets)
print("fuzz ok", a, b)
