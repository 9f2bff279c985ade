#!/usr/bin/env python3
"""Same process, same bytes, different allocations: `k_tile_score` over two copies of the counts x
two engines (plan + workspace + outputs), LAUNCHES launches per combination in a fixed order, so that
scripts/placement_pmc.sh can line rocprofv3's per-dispatch counters up with the combinations.
(The kernel was seen at 2.67-2.93 ms inside ONE process depending on which allocations it ran on:
scripts/clock_trace.py --placement.)"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LAUNCHES = 10


def main():
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
    eng1 = PhaseScoreEngine("cuda:0")
    counts1, offsets1 = synth_csr_device(n, cfg="cfg3", device="cuda:0")
    th = make_filter()
    eng1.score(counts1, offsets1, thresholds=th, algo="tile", reuse_outputs=True)  # builds plan + workspace of engine 1
    torch.cuda.synchronize()
    pad = torch.empty((1 << 30) + 4096 * 37, dtype=torch.uint8, device="cuda:0")
    counts2 = counts1.clone()
    eng2 = PhaseScoreEngine("cuda:0")
    offsets2 = offsets1.clone()
    eng2.score(counts1, offsets2, thresholds=th, algo="tile", reuse_outputs=True)
    torch.cuda.synchronize()
    # (2 launches so far; the combinations below are what the wrapper groups)
    out = []
    for label, e, c, o in (("c1e1", eng1, counts1, offsets1), ("c2e1", eng1, counts2, offsets1), ("c1e2", eng2, counts1, offsets2),
                           ("c2e2", eng2, counts2, offsets2), ("c1e1_again", eng1, counts1, offsets1)):
        t = []
        for _ in range(LAUNCHES):
            e.score(c, o, thresholds=th, algo="tile", reuse_outputs=True, timings=t)
        ms = sorted(x[1] for x in t)
        out.append({"label": label, "main_ms_median": round(ms[len(ms) // 2], 4), "main_ms_min": round(ms[0], 4)})
    del pad
    print(json.dumps({"skip": 2, "launches": LAUNCHES, "combos": out}))


if __name__ == "__main__":
    main()
