#!/usr/bin/env python3
"""Does the ORDER of the two big allocations decide the class the scoring kernel runs in?  Fresh processes, alternating:
  counts_first     the bench's order: the 16 GB of counts, then (at the first score) the engine's record workspace
  workspace_first  the record workspace (and outputs) first -- one throw-away score of a tiny batch sized like the real one is
                   not possible, so: a 1.1 GB placeholder tensor is allocated and FREED to the caching allocator before the
                   counts, which the workspace then reuses
Prints kernel / finish / step per process.  usage: ab_alloc_order.py [rounds]   |   ab_alloc_order.py child <order>"""
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(order):
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    dev = torch.device("cuda", 0)
    eng = PhaseScoreEngine(dev)
    if order == "workspace_first":
        hold = torch.empty(1_200_000_000, dtype=torch.uint8, device=dev)  # (what the workspace + outputs of this batch take)
        hold2 = torch.empty(300_000_000, dtype=torch.uint8, device=dev)
        del hold, hold2  # back to torch's caching allocator: the workspace's allocation will take these blocks
    counts, offsets = synth_csr_device(11_000_000, cfg="cfg3", device=dev)
    thr = make_filter()
    plan = eng.plan_for(offsets, counts.numel(), (counts.data_ptr() // 4) % 4)
    for _ in range(10):
        eng.score(counts, offsets, thresholds=thr, algo="tile", reuse_outputs=True, plan=plan)
    tm = []
    for _ in range(30):
        eng.score(counts, offsets, thresholds=thr, algo="tile", reuse_outputs=True, plan=plan, timings=tm)
    torch.cuda.synchronize()
    print(f"{statistics.median(t[1] for t in tm):.4f} {statistics.median(t[2] for t in tm):.4f} {statistics.median(t[3] for t in tm):.4f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for r in range(rounds):
        for order in ("counts_first", "workspace_first"):
            out = subprocess.run([sys.executable, __file__, "child", order], capture_output=True, text=True).stdout.strip().splitlines()[-1]
            k, f, s = out.split()
            print(f"round {r} {order:16s} kernel {k} ms  finish {f}  step {s}", flush=True)
