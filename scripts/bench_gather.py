"""Gather-side kernels for differently fragmented indexes (run on the GPU box): the per-ORF
gather, the tile gather (gather plan), the fused gather + score and the plain scoring kernel.
usage: python scripts/bench_gather.py [total_nt]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ribotricer_amd.engine import get_engine, make_filter
from ribotricer_amd.gather import GatherPlan, IntervalTable, gather_profiles_device

total = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
rng = np.random.default_rng(3)
eng = get_engine("cuda:0")
th = make_filter()


def table(exon_len, exons_per_orf, gap, reverse_frac=0.5):
    """ORFs laid one after the other along one coverage array, `gap` positions between exons."""
    n_iv = total // exon_len
    n = n_iv // exons_per_orf
    n_iv = n * exons_per_orf
    iv_len = np.full(n_iv, exon_len, np.int32)
    iv_start = np.arange(n_iv, dtype=np.int64) * (exon_len + gap)
    orf_iv = np.arange(n + 1, dtype=np.int64) * exons_per_orf
    reverse = (rng.random(n) < reverse_frac).astype(np.uint8)
    offsets = np.arange(n + 1, dtype=np.int64) * (exon_len * exons_per_orf)
    return IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets), int(iv_start[-1] + exon_len)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


cases = {
    "one 3 000-nt exon, no gaps, forward": (3000, 1, 0, 0.0),
    "one 3 000-nt exon, no gaps, reversed": (3000, 1, 0, 1.0),
    "3 x 120-nt exons, 240-nt gaps": (120, 3, 240, 0.5),
    "3 x 120-nt exons, no gaps": (120, 3, 0, 0.5),
    "3 x 128-nt exons, no gaps, forward": (128, 3, 0, 0.0),
    "one 60-nt exon, 100-nt gaps": (60, 1, 100, 0.5),
}
for name, (exon_len, k, gap, rf) in cases.items():
    t, cov_len = table(exon_len, k, gap, rf)
    cov = torch.poisson(torch.full((cov_len,), 0.3, device="cuda")).to(torch.int32)
    plan = GatherPlan(t, cov_len)
    nt = int(t.offsets[-1])
    legacy = timed(lambda: gather_profiles_device(cov, t))
    tile = timed(lambda: plan.gather(cov))
    counts = plan.gather(cov)
    tm = []
    for _ in range(6):
        eng.score_coverage(cov, plan, thresholds=th, reuse_outputs=True, timings=tm)
    fused = float(np.median([x[1] for x in tm[1:]]))
    tm = []
    for _ in range(6):
        eng.score(counts, plan.offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=tm)
    plain = float(np.median([x[1] for x in tm[1:]]))
    print(f"{name:40s} nt={nt:10d}  per-ORF gather {legacy:.3f} ms ({8*nt/legacy/1e6:5.0f} GB/s)  tile gather {tile:.3f} ms ({8*nt/tile/1e6:5.0f} GB/s)"
          f"  fused kernel {fused:.3f} ms ({4*nt/fused/1e6:5.0f} GB/s)  plain kernel {plain:.3f} ms")
    del cov, counts, plan
