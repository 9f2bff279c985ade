#!/usr/bin/env python3
"""Is the finish pass of SMALL batches bound by its longest chain of sequential re-walks in one wave?  Timing experiment:
libraries built with -DRP_EXPERIMENT_MAX_REWALKS=k (a wave re-walks at most k of its too-close-to-call ORFs; RESULTS WRONG)
and -DRP_EXPERIMENT_NO_REWALK against the product, each in its own process, on cfg2 / cfg3 batches of several sizes.
usage: python scripts/ab_finish_tail.py            (parent: runs every library)      |  ... child <n_orfs> <cfg>"""
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(n, cfg):
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    eng = PhaseScoreEngine("cuda:0")
    counts, offsets = synth_csr_device(n, cfg=cfg, device="cuda:0")
    thr = make_filter()
    plan = eng.plan_for(offsets, counts.numel(), (counts.data_ptr() // 4) % 4)
    for _ in range(10):
        eng.score(counts, offsets, thresholds=thr, algo="tile", reuse_outputs=True, plan=plan)
    tm = []
    for _ in range(40):
        eng.score(counts, offsets, thresholds=thr, algo="tile", reuse_outputs=True, plan=plan, timings=tm)
    torch.cuda.synchronize()
    print(f"{statistics.median(t[1] for t in tm):.4f} {statistics.median(t[2] for t in tm):.4f} {statistics.median(t[3] for t in tm):.4f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    libs = [("product", None), ("max 1 re-walk per wave", "variants/max_rewalks_1.so"), ("max 2", "variants/max_rewalks_2.so"), ("no re-walk", "variants/no_rewalk.so")]
    if len(sys.argv) > 1:  # explicit list: name=path ...
        libs = [tuple(a.split("=", 1)) for a in sys.argv[1:]]
    for n, cfg in ((344_000, "cfg3"), (1_000_000, "cfg2"), (1_374_000, "cfg3"), (2_750_000, "cfg3"), (11_000_000, "cfg3")):
        for name, lib in libs:
            env = dict(os.environ)
            if lib:
                env["RIBOPHASE_LIB"] = os.path.join(REPO, lib)
            out = subprocess.run([sys.executable, __file__, "child", str(n), cfg], capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
            k, f, st = out.split()
            print(f"{cfg} {n:>9d} ORFs  {name:24s} kernel {k} ms  finish {f} ms  step {st} ms", flush=True)
