"""Latency of one scoring call for small batches, tile vs wave path (run on the GPU box)."""
import sys, time
import torch
sys.path.insert(0, ".")
from ribotricer_amd.engine import PhaseScoreEngine, make_filter
from ribotricer_amd.synth import synth_csr_host

eng = PhaseScoreEngine("cuda:0")
thr = make_filter()
for n in (1, 10, 100, 1000, 10000, 100000):
    c, o = synth_csr_host(n, seed=3, cfg="cfg2")
    dc, do = torch.from_numpy(c).cuda(), torch.from_numpy(o).cuda()
    row = []
    for algo in ("tile", "wave"):
        for _ in range(5):
            eng.score(dc, do, thresholds=thr, algo=algo, reuse_outputs=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 200
        for _ in range(reps):
            eng.score(dc, do, thresholds=thr, algo=algo, reuse_outputs=True)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t) / reps * 1e6)
    print(f"n={n:7d} nt={c.size:9d}  tile {row[0]:8.1f} us/call   wave {row[1]:8.1f} us/call")
