#!/usr/bin/env python3
# ARCHIVED (round 6): measured the opt-in RIBOTRICER_AMD_PLACE_WORKSPACE knob of export_orf_coverages, which round 6 removed
# (profiles/archive/r05_placement_check.txt is its record).  Kept for the record; the product half no longer places anything.
"""Does the PRODUCT get the placement the bench enjoys?  (round-3 verdict, item 2)

P fresh processes each of
  bench     `python bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-fused`: kernel time of the timed steps (after
            engine.tune_workspace), the step on the workspace as first allocated, memory reserved after the search;
  product   this file with `child`: `export_orf_coverages` (default mode) twice on a nested candidate index
            (scripts/gen_big_index.cpp) -- the first call parses, plans and places the workspace (once per cached
            index), the second is a further sample -- then the fused kernel timed on the product's own cached plan
            and workspace, and, for reference, an exhaustive search (every candidate tried) in the same process.
usage: placement_check.py [processes] [n_orfs of the product index]"""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(n_orfs: int) -> None:
    import tempfile

    import numpy as np
    import torch

    os.environ["RIBOTRICER_AMD_PLACE_WORKSPACE"] = "1"  # (opt-in in the product: this script measures what it would buy)
    from ribotricer_amd import detect_orfs as d
    from ribotricer_amd.alignments import MergedColumns, build_coverage_device
    from ribotricer_amd.engine import get_engine, make_filter

    tmp = tempfile.mkdtemp(prefix="rpplace_")
    exe = os.path.join(tmp, "gen_big_index")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(REPO, "scripts", "gen_big_index.cpp")])
    prefix = os.path.join(tmp, "big")
    subprocess.check_call([exe, prefix, str(n_orfs)], stderr=subprocess.DEVNULL)
    index_path = prefix + "_candidate_orfs.tsv"
    ex = np.fromfile(prefix + "_exons.bin", dtype=np.dtype([("chrom", "<i4"), ("strand", "<i4"), ("start", "<i8"), ("end", "<i8")]))
    rng = np.random.default_rng(3)
    n_align = 30_000_000
    pick = rng.integers(0, ex.size, n_align)
    pos = ex["start"][pick] + (rng.random(n_align) * (ex["end"][pick] - ex["start"][pick] + 1)).astype(np.int64)
    cols = MergedColumns(ex["strand"][pick].astype(np.uint8), ex["chrom"][pick].astype(np.int32), pos, np.ones(n_align, np.int64),
                         [f"chr{k}" for k in range(1, 25)])
    del ex, pick, pos
    walls, stages = [], []
    for k in range(3):  # (the placement search runs when the second sample of a cached index arrives)
        tm: dict = {}
        t0 = time.perf_counter()
        d.export_orf_coverages(index_path, cols, prefix + f"_s{k}", timings=tm)
        walls.append(time.perf_counter() - t0)
        stages.append(tm)
    index = next(reversed(d._INDEX_CACHE.values()))
    (table, plan, extras), = index.__dict__["_layout_cache"].values()
    coverage, _ = build_coverage_device(cols, index, cmap=extras.get("coverage_map"))
    eng = get_engine("cuda:0")
    th = make_filter()

    def kernel_ms():
        tm: list = []
        for _ in range(2):
            eng.score_coverage(coverage, plan, thresholds=th, reuse_outputs=True)
        for _ in range(7):
            eng.score_coverage(coverage, plan, thresholds=th, reuse_outputs=True, timings=tm)
        return sorted(x[1] for x in tm)[3], sorted(x[1] + x[2] for x in tm)[3]

    product = kernel_ms()
    reserved = int(torch.cuda.memory_reserved())
    allocated = int(torch.cuda.memory_allocated())
    full = eng.tune_workspace(coverage, thresholds=th, gather_plan=plan, spread=1.0)  # try every candidate
    best = kernel_ms()
    print(json.dumps({"n_orfs": n_orfs, "total_nt": int(table.offsets[-1]), "export_wall_s": walls,
                      "placement_in_export": extras.get("workspace_placement"), "placement_stage_s": stages[1].get("workspace_placement"),
                      "product_kernel_ms": product[0], "product_step_ms": product[1], "reserved_bytes": reserved, "allocated_bytes": allocated,
                      "exhaustive_search_step_ms": full["step_ms"], "kernel_ms_after_exhaustive_search": best[0],
                      "product_over_best": product[0] / min(best[0], product[0])}))


def main() -> None:
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n_orfs = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000_000
    for k in range(procs):
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "30", "--warmup", "5", "--cpu-sample", "0", "--no-fused"],
                             capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print("bench FAILED", out.stderr[-300:], flush=True)
            continue
        b = json.loads(line[-1])
        r, wp = b["roofline"], b["config"]["workspace_placement"]
        print("bench  ", json.dumps({"value": b["value"], "kernel_ms": round(r["kernel_ms"], 4), "frac": round(r["frac"], 4), "step_frac": round(r["step_frac"], 4),
                                     "value_first_allocation": b.get("value_first_allocation"), "first_allocation_step_ms": b.get("first_allocation", {}).get("ms_per_step"),
                                     "search_step_ms": wp["step_ms"], "chosen": wp["chosen"], "reserved_GB_after_search": round(wp["reserved_bytes_after"] / 1e9, 2),
                                     "workspace_GB": round(wp["workspace_bytes"] / 1e9, 3)}), flush=True)
    for k in range(procs):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n_orfs)], capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        print("product", line[-1] if line else "FAILED " + out.stderr[-400:], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        main()
