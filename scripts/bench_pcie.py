"""PCIe-inclusive rate of the scoring path for a caller whose counts live in HOST memory (the C ABI
takes device pointers; the drop-in path builds its coverage on the device and never does this):
pinned H2D of counts + offsets, score, D2H of the per-ORF outputs.  usage: python scripts/bench_pcie.py [cfg] [n_orfs]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from ribotricer_amd.engine import get_engine, make_filter
from ribotricer_amd.synth import synth_csr_device

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
eng = get_engine("cuda:0")
d_counts, d_offsets = synth_csr_device(n, cfg=cfg, device="cuda:0")
h_counts = torch.empty(d_counts.shape, dtype=d_counts.dtype, pin_memory=True).copy_(d_counts)
h_offsets = torch.empty(d_offsets.shape, dtype=d_offsets.dtype, pin_memory=True).copy_(d_offsets)
th = make_filter()
plan_offsets = d_offsets  # the index (offsets + tile plan) stays resident: one index, many samples
eng.score(d_counts, plan_offsets, thresholds=th, algo="tile")
torch.cuda.synchronize()
out_host = None
times = {"h2d": [], "score": [], "d2h": [], "total": []}
buf = torch.empty_like(d_counts)
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    buf.copy_(h_counts, non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res = eng.score(buf, plan_offsets, thresholds=th, algo="tile", reuse_outputs=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out_host = {k: (None if v is None else v.cpu()) for k, v in res._asdict().items()}
    t3 = time.perf_counter()
    for k, v in zip(times, (t1 - t0, t2 - t1, t3 - t2, t3 - t0)):
        times[k].append(v)
best = {k: min(v) for k, v in times.items()}
nbytes = h_counts.numel() * 4
print(json.dumps({"cfg": cfg, "n_orfs": n, "counts_bytes": nbytes, "h2d_s": round(best["h2d"], 5), "h2d_GBps": round(nbytes / best["h2d"] / 1e9, 1),
                  "score_s": round(best["score"], 5), "d2h_outputs_s": round(best["d2h"], 5), "total_s": round(best["total"], 5),
                  "orfs_per_s_pcie_inclusive": round(n / best["total"]), "orfs_per_s_resident": round(n / best["score"])}))
