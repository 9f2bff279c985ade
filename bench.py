#!/usr/bin/env python3
"""bench.py -- ORFs phase-scored per second on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (libribophase rp_phase_score_csr_dev: tile index
+ scoring kernel + per-ORF finish) over one synthetic CSR batch that is already
resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU -- 1 M
synthetic ORFs, mean ~300 nt, Poisson P-site counts (ribotricer_amd/synth.py "cfg2").
ORFs are independent, so ranks hold disjoint ORF-index slices and exchange nothing on
the data path ("scaling": "weak"; the only collectives are the barrier and the
max-over-ranks of the elapsed time).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (algorithmic bytes
B = 4*sum(L) + 8*(n+1) + 24*n per launch over its HIP-event duration, vs the 8 TB/s
HBM peak); `cpu_baseline` is the literal scipy restatement of the reference's
phasescore (oracle/phasescore_literal.py, 1 core) on a bounded sample of the same batch.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CFG_NAMES = {"cfg2": "BASELINE configs[1]", "cfg3": "BASELINE configs[2]", "cfg5": "BASELINE configs[4]"}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--orfs", type=int, default=1_000_000, help="ORFs per GPU")
    ap.add_argument("--cfg", default="cfg2", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--algo", default="auto", choices=["auto", "wave", "tile", "pipe"])
    ap.add_argument("--cpu-sample", type=int, default=30000, help="ORFs for the CPU baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=20260213)
    return ap.parse_args()


def cpu_baseline(counts_host, offsets_host, n_sample):
    """Reference-equivalent CPU path on a bounded sample: pure-Python triplet loop +
    scipy.signal.coherence per frame (oracle/phasescore_literal.py), single core."""
    import numpy as np

    from oracle import c_oracle
    from oracle.phasescore_literal import phasescore_literal

    n = min(n_sample, offsets_host.size - 1)
    profiles = [counts_host[offsets_host[i] : offsets_host[i + 1]].tolist() for i in range(n)]
    t0 = time.perf_counter()
    for p in profiles:
        phasescore_literal(p)
    dt = time.perf_counter() - t0
    out = {
        "value": n / dt,
        "unit": "ORFs/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n} ORFs of the rank-0 batch ({int(offsets_host[n])} nt), "
        f"oracle/phasescore_literal.py (python loop + scipy.signal.coherence), {dt:.1f} s",
    }
    # best-effort CPU: C closed form on all host cores over a larger sample
    cores = os.cpu_count() or 1
    n_c = min(200_000, offsets_host.size - 1)
    c_oracle.build()
    c = np.ascontiguousarray(counts_host[: offsets_host[n_c]])
    o = np.ascontiguousarray(offsets_host[: n_c + 1])
    t0 = time.perf_counter()
    c_oracle.phase_score_csr(c, o, n_threads=cores)
    dt_c = time.perf_counter() - t0
    extra = {"value": n_c / dt_c, "unit": "ORFs/s", "cores": cores, "kind": "port",
             "sample": f"first {n_c} ORFs, oracle/phase_oracle.c closed form + OpenMP, {dt_c:.2f} s"}
    return out, extra


def measured_traffic(args, n_orfs, algo):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json: FETCH_SIZE and WRITE_SIZE in KiB, separate --pmc runs of this
    very command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 wide
    streaming reads).  None when no profile matches the workload being run."""
    path = os.path.join(REPO, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        for rec in json.load(fh):
            if rec["cfg"] == args.cfg and rec["orfs_per_gpu"] == n_orfs and rec["algo"] == algo and rec["seed"] == args.seed:
                return int((2 * rec["fetch_size_kib"] + rec["write_size_kib"]) * 1024)
    return None


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        sys.exit("bench.py needs a HIP device (ribotricer_amd has no CPU path)")
    local_dev = local_rank % n_dev  # one rank per GPU on the driver's node; wraps only in 1-GPU self-tests
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        # RCCL ("nccl") carries only the barrier and the max-over-ranks of the elapsed time;
        # RP_BENCH_BACKEND=gloo lets the same control flow be exercised with ranks sharing a GPU
        backend = os.environ.get("RP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from ribotricer_amd import _lib
    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    # each rank owns a disjoint slice of the ORF index: its own seeded batch
    counts, offsets = synth_csr_device(args.orfs, seed=args.seed + 1000 * rank, cfg=args.cfg, device=dev)
    n_orfs = offsets.numel() - 1
    total_nt = counts.numel()
    eng = PhaseScoreEngine(dev)
    thresholds = make_filter()
    algo = args.algo

    def step():
        return eng.score(counts, offsets, thresholds=thresholds, algo=algo, reuse_outputs=True)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # torch's current stream IS the stream the library launches on
    for _ in range(args.steps):
        out = step()
    ev1.record()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    dev_ms_per_step = ev0.elapsed_time(ev1) / args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel HIP-event timing (events recorded inside the library, on the same stream)
    timings: list = []
    for _ in range(max(5, min(args.steps, 20))):
        eng.score(counts, offsets, thresholds=thresholds, algo=algo, reuse_outputs=True, timings=timings)
    k_index = sum(t[0] for t in timings) / len(timings)
    k_main = sum(t[1] for t in timings) / len(timings)
    k_fin = sum(t[2] for t in timings) / len(timings)

    if rank == 0:
        algo_bytes = 4 * total_nt + 8 * (n_orfs + 1) + 24 * n_orfs
        achieved = algo_bytes / (k_main * 1e-3) / 1e9
        # RP_ALGO_AUTO -> RP_ALGO_TILE from 2 Mi nucleotides up (ribophase.hip), RP_ALGO_WAVE below
        resolved = ("tile" if total_nt >= (2 << 20) else "wave") if algo == "auto" else algo
        traffic = measured_traffic(args, n_orfs, resolved)
        flags = out.flags
        result = {
            "metric": "ORFs phase-scored/sec (whole node)",
            "value": world * n_orfs * args.steps / elapsed,
            "unit": "ORFs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 codon math, f64 reduction/score (int32 counts)",
            "data": "synthetic",
            "config": {
                "workload": f"{CFG_NAMES[args.cfg]}: synthetic {n_orfs} ORFs/GPU, mean {total_nt / n_orfs:.0f} nt, "
                f"Poisson P-site counts ({args.cfg}), resident in HBM",
                "orfs_per_gpu": n_orfs,
                "nt_per_gpu": total_nt,
                "algo": algo,
                "sharding": "independent ORF-index slices, no collective on the data path",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "rp::k_tile_score" if resolved == "tile" else "rp::k_wave_score",
                "kernel_ms": k_main,
                "algorithmic_bytes_per_launch": algo_bytes,
                "step_device_ms": dev_ms_per_step,
                "aux_kernels_ms": {"tile_index": k_index, "orf_finish": k_fin},
                "step_achieved": algo_bytes / (dev_ms_per_step * 1e-3) / 1e9,
            },
            "quality": {
                "tie_flag_rate": float((flags & _lib.FLAG_TIE).ne(0).double().mean()),
                "recheck64_rate": float((flags & _lib.FLAG_RECHECK64).ne(0).double().mean()),
                "translating": int(out.status.sum()),
            },
        }
        if world == 1 and args.cpu_sample > 0:
            n_s = min(max(args.cpu_sample, 200_000), n_orfs)
            o_host = offsets[: n_s + 1].cpu().numpy()
            c_host = counts[: int(o_host[-1])].cpu().numpy()
            base, extra = cpu_baseline(c_host, o_host, args.cpu_sample)
            result["cpu_baseline"] = base
            result["cpu_closed_form_c"] = extra
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
