#!/usr/bin/env python3
"""Does it matter which of several workspace allocations the scoring kernel writes its records to?  One set, one
engine, K workspaces allocated one after the other (spacer allocations in between), the kernel timed on each in two
rounds (order bias), then engine.tune_workspace.  usage: tune_workspace_probe.py [K] [spacer GiB]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from ribotricer_amd import _lib
    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    k_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    spacer = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    counts, offsets = synth_csr_device(11_000_000, cfg="cfg3", device="cuda:0")
    eng = PhaseScoreEngine("cuda:0")
    th = make_filter()
    eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True)
    key = int(torch.cuda.current_stream().cuda_stream)
    nbytes = eng._workspace[key].numel()
    pool, spacers = [eng._workspace[key]], []
    for _ in range(1, k_cand):
        spacers.append(torch.empty(int(spacer * (1 << 30)), dtype=torch.uint8, device="cuda:0"))
        pool.append(torch.empty(nbytes, dtype=torch.uint8, device="cuda:0"))

    def ms(ws):
        eng._workspace[key] = ws
        t = []
        eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True)
        for _ in range(5):
            eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=t)
        return round(sorted(x[1] for x in t)[2], 3)

    rounds = [[ms(w) for w in pool] for _ in range(2)] + [[ms(w) for w in reversed(pool)][::-1]]
    print(json.dumps({"counts": hex(counts.data_ptr()), "workspaces": [hex(w.data_ptr()) for w in pool], "kernel_ms_rounds": rounds}))


if __name__ == "__main__":
    main()
