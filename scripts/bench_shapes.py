"""Kernel time per nucleotide for differently shaped batches (run on the GPU box):
short ORFs stress the 64-segment chunks of a tile, long ORFs the multi-tile records."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from ribotricer_amd.engine import PhaseScoreEngine, make_filter
from ribotricer_amd.synth import orf_lengths

eng = PhaseScoreEngine("cuda:0")
rng = np.random.default_rng(5)
total = 300_000_000
shapes = {
    "60 nt (CLI minimum)": lambda: np.full(total // 60, 60),
    "60-150 nt": lambda: 3 * rng.integers(20, 51, total // 105),
    "cfg2-like (mean 300)": lambda: 3 * np.clip(np.rint(rng.lognormal(np.log(80), 0.7, total // 307)), 20, 33333).astype(np.int64),
    "gencode-like (60 % 60-150 nt)": lambda: orf_lengths(total // 279, 5, "gencode"),
    "cfg3 law (mean 361)": lambda: orf_lengths(total // 361, 5, "cfg3"),
    "3 000 nt": lambda: np.full(total // 3000, 3000),
    "30 000 nt": lambda: np.full(total // 30000, 30000),
}
for name, make in shapes.items():
    lengths = np.asarray(make(), np.int64)
    offsets = np.zeros(lengths.size + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    nt = int(offsets[-1])
    counts = torch.poisson(torch.full((nt,), 0.3, device="cuda")).to(torch.int32)
    d_off = torch.from_numpy(offsets).cuda()
    thr = make_filter()
    for algo in ("tile",):
        timings = []
        for _ in range(8):
            eng.score(counts, d_off, thresholds=thr, algo=algo, reuse_outputs=True, timings=timings)
        t = np.array(timings[2:]).mean(axis=0)
        print(f"{name:30s} n={lengths.size:9d} nt={nt:11d}  index {t[0]:.3f}  score {t[1]:.3f}  finish {t[2]:.3f} ms"
              f"  -> {4*nt/t[1]/1e6:.0f} GB/s counts through the tile kernel, {lengths.size/t[3]/1e3:.0f} M ORFs/s")
    del counts, d_off
