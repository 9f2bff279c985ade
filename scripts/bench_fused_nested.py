#!/usr/bin/env python3
"""The fused gather + score kernel on the NESTED candidate index of scripts/gen_big_index.cpp (transcripts with
nested candidate ORFs, 65 % of them 60-150 nt, ~2 exons per ORF, consecutive transcripts on different chromosomes):
the launch `export_orf_coverages` makes in default mode on a human-sized index.  Prints how the gather plan stages
the tiles (GatherPlan.stats), the fused kernel's time next to the plain kernel's on the gathered profiles of the
same index, and checks fused == gather-then-score bit for bit.  Made to be run under rocprofv3 as well
(`rocprofv3 --pmc ... -- python3 scripts/bench_fused_nested.py ...`): the launches of interest are
rp::k_tile_score<true, ...> (fused) and rp::k_tile_score<false, ...> (plain).
usage: python scripts/bench_fused_nested.py [n_orfs] [n_alignment_entries] [launches] [out.json]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ribotricer_amd.alignments import MergedColumns, build_coverage_device  # noqa: E402
from ribotricer_amd.engine import get_engine, make_filter  # noqa: E402
from ribotricer_amd.gather import interval_table_from_index, make_gather_plan  # noqa: E402
from ribotricer_amd.index import NativeIndex  # noqa: E402

n_orfs = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
n_align = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
out_json = sys.argv[4] if len(sys.argv) > 4 else None
tmp = tempfile.mkdtemp(prefix="rpnest_")
exe = os.path.join(tmp, "gen_big_index")
subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(REPO, "scripts", "gen_big_index.cpp")])
prefix = os.path.join(tmp, "big")
subprocess.check_call([exe, prefix, str(n_orfs)])
index_path = prefix + "_candidate_orfs.tsv"
ex = np.fromfile(prefix + "_exons.bin", dtype=np.dtype([("chrom", "<i4"), ("strand", "<i4"), ("start", "<i8"), ("end", "<i8")]))
rng = np.random.default_rng(3)
pick = rng.integers(0, ex.size, n_align)
pos = ex["start"][pick] + (rng.random(n_align) * (ex["end"][pick] - ex["start"][pick] + 1)).astype(np.int64)
cols = MergedColumns(ex["strand"][pick].astype(np.uint8), ex["chrom"][pick].astype(np.int32), pos, np.ones(n_align, np.int64) + (rng.random(n_align) < 0.2),
                     [f"chr{k}" for k in range(1, 25)])
del ex, pick, pos
index = NativeIndex.from_file(index_path)
os.remove(index_path)
os.remove(prefix + "_exons.bin")
coverage, base = build_coverage_device(cols, index, big={})
table = interval_table_from_index(index, base)
plan = make_gather_plan(table, coverage.numel())
eng = get_engine("cuda:0")
thr = make_filter()
report = {"n_orfs": n_orfs, "exons": int(table.iv_start.size), "total_nt": int(table.offsets[-1]), "coverage_positions": int(coverage.numel()),
          "orfs_60_150_nt": float(((np.diff(table.offsets) >= 60) & (np.diff(table.offsets) <= 150)).mean()), "gather_plan": plan.stats()}
n, total_nt = n_orfs, int(table.offsets[-1])
algo_bytes = 4 * total_nt + 8 * (n + 1) + 24 * n
tm: list = []
for _ in range(3):
    fused = eng.score_coverage(coverage, plan, thresholds=thr, reuse_outputs=True)
for _ in range(launches):
    fused = eng.score_coverage(coverage, plan, thresholds=thr, reuse_outputs=True, timings=tm)
k = sorted(t[1] for t in tm)[len(tm) // 2]
report["fused"] = {"kernel_ms": k, "finish_ms": sorted(t[2] for t in tm)[len(tm) // 2], "step_ms": sorted(t[3] for t in tm)[len(tm) // 2],
                   "frac": algo_bytes / (k * 1e-3) / 8e12, "algorithmic_bytes": algo_bytes}
fused = {kk: getattr(fused, kk).clone() for kk in ("phase", "valid", "read_count", "min_codon_cov", "flags", "status")}
counts = plan.gather(coverage)
tm = []
for _ in range(3):
    csr = eng.score(counts, plan.offsets, thresholds=thr, algo="tile", reuse_outputs=True)
for _ in range(launches):
    csr = eng.score(counts, plan.offsets, thresholds=thr, algo="tile", reuse_outputs=True, timings=tm)
k = sorted(t[1] for t in tm)[len(tm) // 2]
report["plain_on_gathered_profiles"] = {"kernel_ms": k, "finish_ms": sorted(t[2] for t in tm)[len(tm) // 2], "frac": algo_bytes / (k * 1e-3) / 8e12}
torch.cuda.synchronize()
report["fused_equals_gather_then_score_bitwise"] = all(bool(torch.equal(fused[kk], getattr(csr, kk))) for kk in fused)
report["fused_over_plain"] = report["fused"]["kernel_ms"] / report["plain_on_gathered_profiles"]["kernel_ms"]
print(json.dumps(report, indent=1))
if out_json:
    with open(out_json, "w") as fh:
        json.dump(report, fh, indent=1)
assert report["fused_equals_gather_then_score_bitwise"]
