#!/usr/bin/env python3
"""A/B of libribophase builds on ONE box: every variant is timed in its own process,
variants interleaved over several rounds (boxes and clocks drift by a few percent).

    python scripts/ab_variants.py base=ribotricer_amd/csrc/libribophase.so k11=variants/k11.so \
        --cfgs cfg2:1000000,cfg3:3000000 --rounds 3

Each run checks parity against the C oracle on a 20 000-ORF sample first (a fast kernel
with different results is not a candidate) and prints the HIP-event medians of the
scoring and finish kernels.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(cfg, n_orfs, calls, fused=False):
    sys.path.insert(0, REPO)
    import numpy as np
    import torch

    from oracle import c_oracle
    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device, synth_csr_host

    eng = PhaseScoreEngine("cuda:0")
    ok = True
    msg = ""
    for pc, pn in (("cfg3", 20000), ("cfg5", 6000)):
        c, o = synth_csr_host(pn, seed=77, cfg=pc)
        ref = c_oracle.phase_score_csr(c, o, n_threads=8)
        r = eng.score(c, o, thresholds=make_filter(), algo="tile").cpu_numpy()
        tie = (ref.flags & 1) != 0  # exact frame ties: the device replays the reference's float64 arithmetic
        rep = c_oracle.replay_csr(c, o)
        want_valid = np.where(tie, rep.valid, ref.valid)
        good = (np.abs(r["phase"] - ref.phase).max() <= 1e-6 and np.array_equal(r["valid"], want_valid)
                and np.array_equal(r["read_count"], ref.read_count) and np.array_equal(r["min_codon_cov"], ref.min_codon_cov)
                and np.array_equal(r["flags"] & 1, ref.flags & 1))
        if not good:
            ok = False
            msg += f"{pc}: dphase={np.abs(r['phase'] - ref.phase).max():.2e} valid_diff={(r['valid'] != want_valid).sum()} "
    counts, offsets = synth_csr_device(n_orfs, cfg=cfg, device="cuda:0")
    th = make_filter()
    for _ in range(10):
        eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True)
    torch.cuda.synchronize()
    t = []
    for _ in range(calls):
        eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=t)
    # whole-step time without the per-launch events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True)
    e1.record()
    torch.cuda.synchronize()
    res = {"ok": ok, "msg": msg, "main_ms": statistics.median(x[1] for x in t),
           "finish_ms": statistics.median(x[2] for x in t), "step_ms": e0.elapsed_time(e1) / calls,
           "nt": counts.numel(), "n": offsets.numel() - 1}
    if fused:  # the fused gather + score on an exon layout of the same length law (synth_exon_layout)
        from ribotricer_amd.gather import GatherPlan, IntervalTable
        from ribotricer_amd.synth import orf_lengths, synth_coverage_device, synth_exon_layout

        del counts
        lay = synth_exon_layout(orf_lengths(n_orfs, 20260213, cfg), 20260213)
        cov = synth_coverage_device(lay[5], 20260213, device="cuda:0")
        gplan = GatherPlan(IntervalTable(*lay[:5]), lay[5], "cuda:0")
        for _ in range(5):
            eng.score_coverage(cov, gplan, thresholds=th, reuse_outputs=True)
        tf = []
        for _ in range(calls):
            eng.score_coverage(cov, gplan, thresholds=th, reuse_outputs=True, timings=tf)
        res["fused_main_ms"] = statistics.median(x[1] for x in tf)
        res["fused_finish_ms"] = statistics.median(x[2] for x in tf)
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--cfgs", default="cfg2:1000000,cfg3:3000000")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--one", nargs=2)
    ap.add_argument("--fused", action="store_true", help="also time the fused gather + score on an exon layout of the same law")
    a = ap.parse_args()
    if a.one:
        one(a.one[0], int(a.one[1]), a.calls, a.fused)
        return
    variants = [v.split("=", 1) for v in a.variants]
    cfgs = [c.split(":") for c in a.cfgs.split(",")]
    res = {}
    for r in range(a.rounds):
        for name, path in variants:
            for cfg, n in cfgs:
                env = dict(os.environ, RIBOPHASE_LIB=os.path.abspath(path))
                out = subprocess.run([sys.executable, __file__, "--one", cfg, n, "--calls", str(a.calls)] + (["--fused"] if a.fused else []), env=env,
                                     capture_output=True, text=True, timeout=600)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                if out.returncode != 0 or not line:
                    res.setdefault((name, cfg), []).append(None)
                    print(f"# {name} {cfg} FAILED: {out.stderr[-400:]}", flush=True)
                    continue
                res.setdefault((name, cfg), []).append(json.loads(line[-1]))
    print(f"{'variant':<14} {'cfg':<6} {'parity':<7} {'main_ms (rounds)':<34} {'best':>8} {'GB/s':>7} {'frac':>6} {'finish':>8} {'step':>8}")
    for (name, cfg), runs in res.items():
        good = [x for x in runs if x]
        if not good:
            print(f"{name:<14} {cfg:<6} FAILED")
            continue
        mains = [x["main_ms"] for x in good]
        best = min(mains)
        x0 = good[0]
        by = 4 * x0["nt"] + 8 * (x0["n"] + 1) + 24 * x0["n"]
        gbs = by / (best * 1e-3) / 1e9
        par = "ok" if all(x["ok"] for x in good) else "BAD"
        print(f"{name:<14} {cfg:<6} {par:<7} {' '.join(f'{m:.4f}' for m in mains):<34} {best:8.4f} {gbs:7.0f} {gbs / 8000:6.3f} "
              f"{min(x['finish_ms'] for x in good):8.4f} {min(x['step_ms'] for x in good):8.4f}"
              + (f"   fused {min(x['fused_main_ms'] for x in good):8.4f} + {min(x['fused_finish_ms'] for x in good):7.4f}" if "fused_main_ms" in good[0] else "")
              + ("   " + good[0]["msg"] if par == "BAD" else ""), flush=True)


if __name__ == "__main__":
    main()
