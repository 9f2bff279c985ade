#!/usr/bin/env python3
"""K copies of the counts x K engines (plan + workspace + outputs each) in ONE process: the median
`k_tile_score` time of every combination.  The bytes and the virtual alignment are the same in all
of them; what differs is where the buffers lie physically.  usage: placement_grid.py [n_orfs] [K]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from ribotricer_amd._probe import stream_read_GBps
    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    th = make_filter()
    c0, o0 = synth_csr_device(n, cfg="cfg3", device="cuda:0")
    copies, engines, pads = [c0], [], []
    for k in range(K):
        if k:
            pads.append(torch.empty((64 << 20) * k + 4096 * 37, dtype=torch.uint8, device="cuda:0"))  # keeps later allocations from reusing a shape
            copies.append(c0.clone())
        e, o = PhaseScoreEngine("cuda:0"), o0.clone()
        e.score(c0, o, thresholds=th, algo="tile", reuse_outputs=True)
        engines.append((e, o))
    torch.cuda.synchronize()
    reads = [round(stream_read_GBps(c, launches=10)[0]) for c in copies]
    grid, fin = [], []
    for ci, c in enumerate(copies):
        row = []
        for e, o in engines:
            t = []
            for _ in range(2):
                e.score(c, o, thresholds=th, algo="tile", reuse_outputs=True)
            for _ in range(12):
                e.score(c, o, thresholds=th, algo="tile", reuse_outputs=True, timings=t)
            ms = sorted(x[1] for x in t)
            row.append(round(ms[len(ms) // 2], 4))
            fin.append(sorted(x[2] for x in t)[len(t) // 2])
        grid.append(row)
    flat = [x for r in grid for x in r]
    doc = {"lib": os.environ.get("RIBOPHASE_LIB", "product"), "n_orfs": n, "K": K, "plain_read_GBps_per_copy": reads,
           "main_ms[counts copy][engine]": grid, "min": min(flat), "median": sorted(flat)[len(flat) // 2], "max": max(flat),
           "finish_ms_min_median_max": [round(min(fin), 4), round(sorted(fin)[len(fin) // 2], 4), round(max(fin), 4)], "spread": round(max(flat) / min(flat) - 1, 4)}
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
