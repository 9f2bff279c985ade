"""One-off stress run (GPU box): many random interval tables / coverage depths, fused gather + score
against gather-then-score (bit for bit) and the C oracle; random CSR batches of every length law
against the oracle.  usage: python scripts/stress_fused.py [n_rounds]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from helpers import assert_matches_oracle  # noqa: E402
from ribotricer_amd.engine import get_engine, make_filter  # noqa: E402
from ribotricer_amd.gather import GatherPlan, IntervalTable  # noqa: E402
from ribotricer_amd.synth import synth_csr_host  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = get_engine("cuda:0")
th = make_filter()
for r in range(rounds):
    rng = np.random.default_rng(9000 + r)
    cov_len = int(rng.integers(50_000, 800_000))
    lam = float(rng.choice([0.005, 0.02, 0.2, 1.0, 4.0]))
    cov = rng.poisson(lam, size=cov_len).astype(np.int32)
    n = int(rng.integers(50, 6000))
    max_exons = int(rng.choice([1, 3, 8, 30]))
    hi = int(rng.choice([3, 40, 150, 600, 5000]))
    n_iv = rng.integers(1, max_exons + 1, size=n)
    orf_iv = np.concatenate([[0], np.cumsum(n_iv)]).astype(np.int64)
    iv_len = rng.integers(1, hi + 1, size=int(orf_iv[-1])).astype(np.int32)
    iv_start = rng.integers(0, cov_len - hi - 1, size=int(orf_iv[-1])).astype(np.int64)
    reverse = rng.integers(0, 2, size=n).astype(np.uint8)
    lengths = np.add.reduceat(iv_len.astype(np.int64), orf_iv[:-1])
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    t = IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets)
    plan = GatherPlan(t, cov_len)
    d_cov = torch.from_numpy(cov).cuda()
    fused = eng.score_coverage(d_cov, plan, thresholds=th).cpu_numpy()
    counts = plan.gather(d_cov)
    plain = eng.score(counts, plan.offsets, thresholds=th, algo="tile").cpu_numpy()
    for k in fused:
        assert np.array_equal(fused[k], plain[k]), (r, k)
    assert_matches_oracle(fused, counts.cpu().numpy(), offsets)
    cfg = ["cfg2", "cfg3", "cfg5", "gencode", "orf60"][r % 5]
    c, o = synth_csr_host(int(rng.integers(2000, 40000)), seed=int(rng.integers(1, 10**6)), cfg=cfg)
    res = eng.score(c, o, thresholds=th, algo="tile").cpu_numpy()
    assert_matches_oracle(res, c, o)
    print(f"round {r}: table n={n} exons<={max_exons} len<={hi} lam={lam} total={int(offsets[-1])} ties={int((fused['flags'] & 1).sum())}; {cfg} ok", flush=True)
print("stress ok")
