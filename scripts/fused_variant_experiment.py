"""Fused kernel of whichever library RIBOPHASE_LIB names, on the two bench layouts (6 M ORFs each) over the compact
coverage with one-position blocks: median kernel ms.  For A/B runs of build variants (ribotricer_amd/csrc/variants/*.so)."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ribotricer_amd.engine import get_engine, make_filter
from ribotricer_amd.gather import CoverageMap, GatherPlan, IntervalTable
from ribotricer_amd.synth import orf_lengths, synth_coverage_device, synth_exon_layout, synth_nested_layout

eng = get_engine("cuda:0")
th = make_filter()
out = {}
n = int(os.environ.get("RP_EXPERIMENT_ORFS", 6_000_000))
for name in ("nested", "exons"):
    lay = synth_nested_layout(n, 20260213) if name == "nested" else synth_exon_layout(orf_lengths(n, 20260213, "cfg3"), 20260213)
    cmap = CoverageMap(IntervalTable(*lay[:5]), lay[5], "cuda:0", block_positions=int(os.environ.get("RP_EXPERIMENT_BLOCK", 1)))
    cmap.release_device_intervals()
    cov = synth_coverage_device(cmap.compact_len, 20260213, device="cuda:0")
    plan = GatherPlan(cmap.table, cmap.compact_len)
    for _ in range(4):
        eng.score_coverage(cov, plan, thresholds=th, reuse_outputs=True)
    tm = []
    for _ in range(15):
        res = eng.score_coverage(cov, plan, thresholds=th, reuse_outputs=True, timings=tm)
    out[name] = round(statistics.median(t[1] for t in tm), 4)
    out[name + "_sum"] = int(res.read_count.sum().item())  # (a variant that computes something else shows here)
    if os.environ.get("RP_EXPERIMENT_PLAIN"):  # the plain kernel on the same profiles, gathered into a CSR array first
        from ribotricer_amd.gather import gather_profiles_device

        counts, offsets = gather_profiles_device(cov, cmap.table, plan=plan)
        tm = []
        for _ in range(12):
            eng.score(counts, offsets, thresholds=th, reuse_outputs=True, timings=tm)
        out[name + "_plain"] = round(statistics.median(t[1] for t in tm[3:]), 4)
        del counts, offsets
    del cov, plan, cmap, res
    torch.cuda.empty_cache()
print(os.path.basename(os.environ.get("RIBOPHASE_LIB", "default")), json.dumps(out))
