#!/usr/bin/env python3
"""engine.tune_workspace on a cfg3 batch: step time before, what the search saw, step time after.
usage: tune_workspace_probe.py [cfg:n_orfs] [spacer GiB] [tries] [spread (1 = try them all)]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    cfg, n = (sys.argv[1] if len(sys.argv) > 1 else "cfg3:11000000").split(":")
    chunk = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    tries = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    counts, offsets = synth_csr_device(int(n), cfg=cfg, device="cuda:0")
    th = make_filter()
    eng = PhaseScoreEngine("cuda:0")

    def now():
        tm = []
        eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True)
        for _ in range(10):
            eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=tm)
        return [round(sorted(x[k] for x in tm)[5], 3) for k in (1, 2)]

    before = now()
    if os.environ.get("KEEP_CACHE") == "1":  # (experiment: do the freed candidates have to go back to the driver?)
        torch.cuda.empty_cache = lambda: None
    t = time.perf_counter()
    rep = eng.tune_workspace(counts, offsets, thresholds=th, spacer_gib=chunk, tries=tries, spread=float(sys.argv[4]) if len(sys.argv) > 4 else 0.06)
    dt = time.perf_counter() - t
    print(json.dumps({"before_kernel_finish_ms": before, "tune_s": round(dt, 2), "search_step_ms": rep["step_ms"], "chosen": rep["chosen"],
                      "after_kernel_finish_ms": now()}))


if __name__ == "__main__":
    main()
