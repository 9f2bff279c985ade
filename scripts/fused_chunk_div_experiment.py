import sys, os, statistics, json
sys.path.insert(0, os.getcwd())
import torch
from ribotricer_amd.engine import get_engine, make_filter
from ribotricer_amd.gather import GatherPlan, IntervalTable
from ribotricer_amd.synth import orf_lengths, synth_coverage_device, synth_exon_layout, synth_nested_layout
eng = get_engine("cuda:0"); th = make_filter(); out = {}
for name in ("nested", "exons"):
    lay = synth_nested_layout(6_000_000, 20260213) if name == "nested" else synth_exon_layout(orf_lengths(6_000_000, 20260213, "cfg3"), 20260213)
    cov = synth_coverage_device(lay[5], 20260213, device="cuda:0"); plan = GatherPlan(IntervalTable(*lay[:5]), lay[5])
    for _ in range(4): eng.score_coverage(cov, plan, thresholds=th, reuse_outputs=True)
    tm = []
    for _ in range(15): eng.score_coverage(cov, plan, thresholds=th, reuse_outputs=True, timings=tm)
    out[name] = round(statistics.median(t[1] for t in tm), 4)
    del cov, plan; torch.cuda.empty_cache()
print(os.path.basename(os.environ.get("RIBOPHASE_LIB", "default")), json.dumps(out))
