#!/usr/bin/env python3
"""Transcript-level reuse for nested indexes: what could a position-major scorer gain?  (round-5 verdict, Next #5)

On the nested index the fused kernel scores 3.50 G profile nt over 1.57 G distinct coverage positions: nested ORFs of one
transcript recompute identical codon terms 2.2 times, and the kernel is VALU-bound (87 % busy), not HBM-bound.  The design
that shares the terms is POSITION-MAJOR: tiles of the compact coverage, ONE codon term per position, per-class (position
mod 3) sums over the ELEMENTARY INTERVALS between consecutive ORF boundaries, and every ORF a sum of the intervals it
covers (sums and counts by addition, the codon minimum by min -- which is why it must be intervals, not prefix differences:
min has no inverse; and why per-position prefixes in LDS are out: 24 B x 7 936 positions leave one workgroup per CU).

This script measures the parts of that design that EXIST, at full size, before anyone builds the rest:
  A  the term pass: the CSR tile kernel (k_tile_score<false>: contiguous 16-byte LDS-DMA, all three classes per segment, one
     48-byte record per segment) run over the compact coverage with the elementary intervals as its segments -- the same
     arithmetic per position and the same record traffic per interval as the term pass would have (its codons would reach
     two positions past an interval's end: a halo the probe ignores);
  B  the census of the combination pass: elementary intervals per ORF (records it must read and fold per ORF);
  C  today's fused step on the same layout (bench.py's fused_nested protocol) for the ratio.
usage: python scripts/reuse_probe.py [n_orfs] [out.json]"""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ribotricer_amd.engine import get_engine, make_filter  # noqa: E402
from ribotricer_amd.gather import CoverageMap, GatherPlan, IntervalTable  # noqa: E402
from ribotricer_amd.synth import synth_coverage_device, synth_nested_layout  # noqa: E402

n_set = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
seed = 20260213
dev = torch.device("cuda", 0)
eng = get_engine(dev)
iv_start, iv_len, orf_iv, reverse, offsets, dense_len = synth_nested_layout(n_set, seed)
table = IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets)
cmap = CoverageMap(table, dense_len, dev)
ctab, compact_len = cmap.table, cmap.compact_len
c_start = np.asarray(ctab.iv_start, np.int64)
c_len = np.asarray(ctab.iv_len, np.int64)
cmap.release_device_intervals()
cov = synth_coverage_device(compact_len, seed, device=dev)
n = offsets.size - 1
total_nt = int(offsets[-1])
rep = {"orfs": n, "exons": int(c_len.size), "profile_nt": total_nt, "compact_positions": int(compact_len), "reuse": total_nt / compact_len}

# ---- C: today's fused step (default mode of the export) ----------------------------------------------------------
gplan = GatherPlan(ctab, compact_len, dev)
thr = make_filter(printed_only=True)
for _ in range(5):
    eng.score_coverage(cov, gplan, thresholds=thr, reuse_outputs=True)
tm: list = []
for _ in range(10):
    eng.score_coverage(cov, gplan, thresholds=thr, reuse_outputs=True, timings=tm)
rep["fused_today"] = {"kernel_ms": statistics.median(t[1] for t in tm), "finish_ms": statistics.median(t[2] for t in tm),
                      "step_ms": statistics.median(t[3] for t in tm)}
del gplan
torch.cuda.empty_cache()

# ---- the elementary intervals -------------------------------------------------------------------------------------
t0 = time.perf_counter()
bounds = np.unique(np.concatenate([c_start, c_start + c_len, np.array([0, compact_len], np.int64)]))
n_int = bounds.size - 1
lens = np.diff(bounds)
# intervals per exon piece (searchsorted on the sorted boundaries), summed per ORF
per_piece = np.searchsorted(bounds, c_start + c_len) - np.searchsorted(bounds, c_start)
per_orf = np.add.reduceat(per_piece, np.asarray(orf_iv[:-1], np.int64))
rep["intervals"] = {"count": int(n_int), "mean_nt": float(lens.mean()), "median_nt": float(np.median(lens)), "share_under_30_nt": float((lens < 30).mean()),
                    "per_orf_mean": float(per_orf.mean()), "per_orf_p50": float(np.median(per_orf)), "per_orf_p99": float(np.percentile(per_orf, 99)),
                    "per_orf_max": int(per_orf.max()), "interval_records_read_by_the_combination_pass": int(per_orf.sum()),
                    "host_build_s": time.perf_counter() - t0}

# ---- A: the term pass, as the CSR tile kernel over (compact coverage, elementary intervals) -----------------------
d_off = torch.from_numpy(bounds).to(dev)
plan = eng.plan_for(d_off, int(compact_len), (cov.data_ptr() // 4) % 4)
for _ in range(5):
    eng.score(cov, d_off, thresholds=None, algo="tile", reuse_outputs=True, plan=plan)
tm = []
for _ in range(10):
    eng.score(cov, d_off, thresholds=None, algo="tile", reuse_outputs=True, plan=plan, timings=tm)
k_main, k_fin = statistics.median(t[1] for t in tm), statistics.median(t[2] for t in tm)
rep["term_pass_as_csr_kernel"] = {
    "kernel_ms": k_main, "finish_ms_of_the_probe_not_of_the_design": k_fin,
    "bytes_read": 4 * int(compact_len), "record_bytes_written": 48 * int(n_int),
    "achieved_GBps_of_coverage_bytes": 4 * compact_len / (k_main * 1e-3) / 1e9,
}
# ---- the estimate ------------------------------------------------------------------------------------------------------
# combination pass: per ORF, read per_orf interval records (48 B each, neighbours in memory: the intervals of a transcript are
# consecutive) and write its own 48-byte record; then today's finish pass.  Priced at the finish pass's measured rate
# (it reads 48-byte records and writes 25 bytes per ORF: ~1.1 GB in finish_ms).
fin = rep["fused_today"]["finish_ms"]
fin_bytes = 48.0 * n * 1.15 + 8.0 * n + 25.0 * n
comb_bytes = 48.0 * float(per_orf.sum()) + 48.0 * n
comb_ms = fin * comb_bytes / fin_bytes
rep["estimate"] = {"term_pass_ms": k_main, "combination_pass_ms_at_the_finish_pass_rate": comb_ms, "finish_ms": fin,
                   "step_ms": k_main + comb_ms + fin, "gain_over_today": rep["fused_today"]["step_ms"] / (k_main + comb_ms + fin)}
print(json.dumps(rep, indent=1))
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(rep, fh, indent=1)
