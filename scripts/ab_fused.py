#!/usr/bin/env python3
"""A/B of libribophase builds on the FUSED kernel, one process per variant and round (RIBOPHASE_LIB selects the build),
variants interleaved: the nested index law (synth_nested_layout) and the bench's exon layout, fused kernel time
(HIP events in the library, median), fused == gather-then-score bit for bit checked in every run.
    python scripts/ab_fused.py base=ribotricer_amd/csrc/libribophase.so pf0=ribotricer_amd/csrc/variants/pf0.so --orfs 6000000 --rounds 3"""
import argparse
import json
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n_orfs: int) -> None:
    sys.path.insert(0, REPO)
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan, IntervalTable
    from ribotricer_amd.synth import orf_lengths, synth_coverage_device, synth_exon_layout, synth_nested_layout

    eng = get_engine("cuda:0")
    th = make_filter()
    out = {}
    for name in ("nested", "exons"):
        if name == "nested":
            layout = synth_nested_layout(n_orfs, 20260213)
        else:
            layout = synth_exon_layout(orf_lengths(n_orfs, 20260213, "cfg3"), 20260213)
        table = IntervalTable(*layout[:5])
        cov = synth_coverage_device(layout[5], 20260213, device="cuda:0")
        plan = GatherPlan(table, layout[5])
        for _ in range(4):
            fused = eng.score_coverage(cov, plan, thresholds=th, reuse_outputs=True)
        tm: list = []
        for _ in range(15):
            fused = eng.score_coverage(cov, plan, thresholds=th, reuse_outputs=True, timings=tm)
        fused = {k: getattr(fused, k).clone() for k in ("phase", "valid", "read_count", "min_codon_cov", "flags", "status")}
        counts = plan.gather(cov)
        tp: list = []
        for _ in range(3):
            csr = eng.score(counts, plan.offsets, thresholds=th, algo="tile", reuse_outputs=True)
        for _ in range(9):
            csr = eng.score(counts, plan.offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=tp)
        torch.cuda.synchronize()
        same = all(bool(torch.equal(fused[k], getattr(csr, k))) for k in fused)
        out[name] = {"fused_ms": statistics.median(t[1] for t in tm), "plain_ms": statistics.median(t[1] for t in tp), "bitwise": same}
        del cov, plan, counts, csr, fused
        torch.cuda.empty_cache()
    print(json.dumps(out))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help="name=path/to/libribophase.so")
    ap.add_argument("--orfs", type=int, default=6_000_000)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    variants = [v.split("=", 1) for v in a.variants]
    seen: dict = {}
    for r in range(a.rounds):
        for name, path in variants:
            env = dict(os.environ, RIBOPHASE_LIB=os.path.abspath(path))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(a.orfs)], env=env, capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(name, "FAILED", p.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            seen.setdefault(name, []).append(d)
            print(f"round {r} {name:8s} " + "  ".join(f"{k}: fused {v['fused_ms']:.4f} plain {v['plain_ms']:.4f} ratio {v['fused_ms'] / v['plain_ms']:.3f} {'ok' if v['bitwise'] else 'MISMATCH'}"
                                                    for k, v in d.items()), flush=True)
    for name, runs in seen.items():
        print(f"median {name:8s} " + "  ".join(f"{k}: fused {statistics.median(x[k]['fused_ms'] for x in runs):.4f} ratio "
                                               f"{statistics.median(x[k]['fused_ms'] / x[k]['plain_ms'] for x in runs):.3f}" for k in runs[0]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        main()
