#!/usr/bin/env python3
"""Does the scoring kernel's time depend on WHERE INSIDE its allocation the record workspace starts?  One set, one
engine, the workspace tensor over-allocated by 4 MiB and handed to the library at a series of byte offsets (the
low 21 address bits are the same in virtual and physical space for 2 MiB pages).  usage: ws_offset_probe.py [n_orfs]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
    th = make_filter()
    counts, offsets = synth_csr_device(n, cfg="cfg3", device="cuda:0")
    eng = PhaseScoreEngine("cuda:0")
    eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True)
    torch.cuda.synchronize()
    stream_key = int(torch.cuda.current_stream().cuda_stream)
    ws0 = eng._workspace[stream_key]
    need = ws0.numel()
    far = len(sys.argv) > 2 and sys.argv[2] == "far"  # offsets of 64 MiB ... 24 GiB inside one 26 GiB allocation
    big = torch.empty(need + ((26 << 30) if far else (8 << 20)), dtype=torch.uint8, device="cuda:0")
    rows = []
    near = (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 2 << 20, (2 << 20) + 4096, 4 << 20, 0)
    steps = (0, 64 << 20, 256 << 20, 512 << 20, 1 << 30, 3 << 29, 2 << 30, 3 << 30, 4 << 30, 6 << 30, 8 << 30, 12 << 30, 16 << 30, 20 << 30, 24 << 30, 0)
    for off in (steps if far else near):
        eng._workspace[stream_key] = big[off : off + need]
        t = []
        for _ in range(3):
            eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True)
        for _ in range(12):
            eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=t)
        ms = sorted(x[1] for x in t)
        rows.append({"offset": off, "main_ms": round(ms[len(ms) // 2], 4), "finish_ms": round(sorted(x[2] for x in t)[len(t) // 2], 4)})
        print(rows[-1], file=sys.stderr, flush=True)
    print(json.dumps({"counts_ptr": hex(counts.data_ptr()), "ws_ptr": hex(big.data_ptr()), "rows": rows}))


if __name__ == "__main__":
    main()
