#!/usr/bin/env python3
"""A longer run of tests/test_gpu_fuzz.py's generator: seeds lo..hi of the boundary-hugging / empty / multi-tile / many-segment /
ragged batch shapes through the tile path (CSR counts at every 16-byte phase) against the C oracle, plus -- every fourth seed --
the same batch as a FUSED call (every ORF one piece of a coverage that is the counts array) checked bit for bit against the CSR
result.  usage: soak_fuzz.py [lo] [hi]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import numpy as np
    import torch

    from helpers import assert_matches_oracle
    from test_gpu_fuzz import lengths_for
    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.gather import GatherPlan, IntervalTable

    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    eng = PhaseScoreEngine("cuda:0")
    th = make_filter()
    th_printed = make_filter(printed_only=True)  # RP_FILTER_PRINTED_ONLY: status everywhere and everything not left open must not change
    from ribotricer_amd._lib import FLAG_RECHECK64, FLAG_UNRESOLVED

    t0 = time.time()
    n_orfs = n_fused = n_open = n_rewalked = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(1000 + seed)
        lens = lengths_for(seed, 7936)
        if seed % 5 == 0:  # (the generator's shapes again, with other lengths mixed in)
            lens = np.concatenate([lens, rng.integers(0, 700, int(rng.integers(1, 4000)))])
            rng.shuffle(lens)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        total = int(offsets[-1])
        lam = rng.choice([0.02, 0.3, 3.0], size=lens.size)
        counts = rng.poisson(np.repeat(lam, lens)).astype(np.int32)
        if seed % 4 == 1 and total:
            counts[rng.integers(0, total, 20)] = 2**24 - 1
        mis = seed % 4
        dev = torch.zeros(total + 8, dtype=torch.int32, device="cuda:0")
        view = dev[mis : mis + total]
        view.copy_(torch.from_numpy(counts))
        d_off = torch.from_numpy(offsets).cuda()
        res = eng.score(view, d_off, thresholds=th, algo="tile").cpu_numpy()
        assert_matches_oracle({k: v.copy() for k, v in res.items()}, counts, offsets)
        n_orfs += lens.size
        pr = eng.score(view, d_off, thresholds=th_printed, algo="tile").cpu_numpy()
        left = (pr["flags"] & FLAG_UNRESOLVED) != 0
        assert np.array_equal(pr["status"], res["status"]) and not pr["status"][left].any(), seed
        assert not (left & ((res["flags"] & FLAG_RECHECK64) == 0)).any(), seed  # only ORFs that would have been re-walked
        for key in res:
            assert np.array_equal(pr[key][~left], res[key][~left], equal_nan=True), (seed, key)
        for key in ("read_count", "min_codon_cov"):
            assert np.array_equal(pr[key][left], res[key][left]), (seed, key)
        n_open += int(left.sum())
        n_rewalked += int(((res["flags"] & FLAG_RECHECK64) != 0).sum())
        if seed % 4 == 0 and total and (lens > 0).all():
            table = IntervalTable(offsets[:-1].copy(), lens.astype(np.int32), np.arange(lens.size + 1, dtype=np.int64),
                                  np.zeros(lens.size, np.uint8), offsets)
            try:
                gplan = GatherPlan(table, total)
            except Exception as e:  # (tables the plan does not take: reported, not a failure of the kernels)
                print(f"seed {seed}: no gather plan ({type(e).__name__}: {e})")
                continue
            cov = torch.from_numpy(counts).cuda()
            fused = eng.score_coverage(cov, gplan, thresholds=th).cpu_numpy()
            plain = eng.score(cov, d_off, thresholds=th, algo="tile").cpu_numpy()
            for key in plain:
                assert np.array_equal(fused[key], plain[key], equal_nan=True), (seed, key)
            n_fused += 1
    print(f"seeds {lo}..{hi - 1}: {n_orfs} ORFs against the oracle, {n_fused} batches fused == CSR bit for bit; RP_FILTER_PRINTED_ONLY on every batch: "
          f"{n_open} of {n_rewalked} too-close-to-call ORFs left open, status and every other ORF's outputs unchanged; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
