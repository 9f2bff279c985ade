#!/usr/bin/env python3
"""Round 5: is the slow class of a scoring step a property of where the COUNTS lie?  One process, the 11 M-ORF set: copies of the
counts array made one after the other, each behind a spacer allocation, against two record workspaces -- the scoring kernel's
HIP-event time on every combination, and a plain read of each copy.  usage: probe_source_placement.py [copies] [spacer GiB]"""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    spacer_gib = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    eng = PhaseScoreEngine("cuda:0")
    th = make_filter()
    counts, offsets = synth_csr_device(11_000_000, cfg="cfg3", device="cuda:0")
    plan = eng.plan_for(offsets, counts.numel(), (counts.data_ptr() // 4) % 4)
    eng.score(counts, offsets, thresholds=th, reuse_outputs=True, plan=plan)
    key = int(torch.cuda.current_stream().cuda_stream)
    workspaces = [eng._workspace[key]]
    keep = [torch.empty(int(spacer_gib * 2**30), dtype=torch.uint8, device="cuda:0")]
    workspaces.append(torch.empty(workspaces[0].numel(), dtype=torch.uint8, device="cuda:0"))
    sources = [counts]
    for _ in range(copies - 1):
        keep.append(torch.empty(int(spacer_gib * 2**30), dtype=torch.uint8, device="cuda:0"))
        sources.append(counts.clone())

    def kernel_ms(src, ws):
        eng._workspace[key] = ws
        for _ in range(3):
            eng.score(src, offsets, thresholds=th, reuse_outputs=True, plan=plan)
        t = []
        for _ in range(8):
            eng.score(src, offsets, thresholds=th, reuse_outputs=True, plan=plan, timings=t)
        return statistics.median(x[1] for x in t), statistics.median(x[1] + x[2] for x in t)

    def read_ms(src):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        src.sum()
        e0.record()
        for _ in range(4):
            src.sum()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 4

    for rnd in range(2):
        for i, src in enumerate(sources):
            line = f"round {rnd} counts copy {i} at {src.data_ptr() / 2**30:9.2f} GiB:"
            for j, ws in enumerate(workspaces):
                k, s = kernel_ms(src, ws)
                line += f"  workspace {j}: kernel {k:.3f} step {s:.3f}"
            print(line + f"  | torch sum of the copy {read_ms(src):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
