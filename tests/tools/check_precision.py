"""GPU-box check: max |phase - float64 oracle| per synthetic config, and the PCIe-inclusive rate."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import c_oracle
from ribotricer_amd.engine import PhaseScoreEngine, make_filter
from ribotricer_amd.synth import synth_csr_host

eng = PhaseScoreEngine("cuda:0")
for cfg, n in (("cfg2", 200000), ("cfg3", 200000), ("cfg5", 100000)):
    c, o = synth_csr_host(n, seed=99, cfg=cfg)
    ref = c_oracle.phase_score_csr(c, o, n_threads=64)
    for algo in ("tile", "wave"):
        r = eng.score(c, o, algo=algo); torch.cuda.synchronize()
        d = np.abs(r.phase.cpu().numpy() - ref.phase)
        print(f"{cfg} {algo}: n={n} nt={c.size} max|dphase|={d.max():.3e} mean={d.mean():.2e} valid_eq={np.array_equal(r.valid.cpu().numpy(), ref.valid)} ties={(ref.flags&1).sum()}")
# PCIe-inclusive: pageable and pinned host CSR -> device -> score -> 26 B/ORF back
c, o = synth_csr_host(1_000_000, seed=20260213, cfg="cfg2")
for pin in (False, True):
    ct, ot = torch.from_numpy(c), torch.from_numpy(o)
    if pin: ct, ot = ct.pin_memory(), ot.pin_memory()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = eng.score(ct.to("cuda:0", non_blocking=True), ot.to("cuda:0", non_blocking=True), thresholds=make_filter())
        host = r.cpu_numpy(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"PCIe-inclusive pinned={pin}: {dt*1e3:.1f} ms per 1M-ORF batch -> {1e6/dt:.3e} ORFs/s")
