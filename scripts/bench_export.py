"""End-to-end timing of the drop-in export path on a synthetic index (run on the GPU box):
index text -> rp_index_parse_host (f3) -> dense coverage -> rp_gather_profiles_dev (f1)
-> rp_phase_score_csr_dev -> D2H -> rp_format_rows_host (f2) -> file.
usage: python scripts/bench_export.py [n_orfs] [reads_per_orf]"""
import json
import os
import sys
import tempfile
import time
from collections import Counter, defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ribotricer_amd import detect_orfs as d  # noqa: E402
from ribotricer_amd import tsv  # noqa: E402
from ribotricer_amd.alignments import MergedColumns, build_coverage_device  # noqa: E402
from ribotricer_amd.gather import gather_profiles_device, interval_table_from_index  # noqa: E402
from ribotricer_amd.index import NativeIndex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
reads_per_orf = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
rng = np.random.default_rng(7)
chroms = [f"chr{k}" for k in range(1, 23)]
lines = ["ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n"]
cursor = {(s, c): 1000 for s in "+-" for c in chroms}
spans = []
for i in range(n):
    c, s = chroms[i % 22], "+-"[(i // 22) % 2]
    pos = cursor[(s, c)]
    k = int(rng.integers(1, 5))
    per = max(10, int(rng.lognormal(np.log(240), 0.7)) // k // 3 * 3 // 1)
    ivs = []
    for _ in range(k):
        ivs.append((pos, pos + per - 1))
        pos += per + int(rng.integers(80, 400))
    cursor[(s, c)] = pos + 50
    spans.append((s, c, ivs))
    lines.append(f"x\tannotated\tENST{i:011d}\tprotein_coding\tENSG{i//3:011d}\tGENE{i//3}\tprotein_coding\t{c}\t{s}\tATG\t{','.join(f'{a}-{b}' for a, b in ivs)}\n")
text = "".join(lines).encode()
align = defaultdict(Counter)
n_reads = int(n * reads_per_orf)
pick = rng.integers(0, n, n_reads)
for j in pick:
    s, c, ivs = spans[j]
    a, b = ivs[int(rng.integers(0, len(ivs)))]
    align[s][(c, int(rng.integers(a, b + 1)))] += 1
tmp = tempfile.mkdtemp()
index_path = os.path.join(tmp, "candidate_orfs.tsv")
open(index_path, "wb").write(text)
NativeIndex(b"")
torch.zeros(1, device="cuda")

T = {}
def lap(name, t0):
    torch.cuda.synchronize()
    T[name] = time.perf_counter() - t0
    return time.perf_counter()

t = time.perf_counter()
index = NativeIndex.from_file(index_path); t = lap("parse_index_f3", t)
coverage, base = build_coverage_device(align, index); t = lap("counter_shim_to_device_coverage", t)
cols = MergedColumns.from_counters(align)  # what the native front end (split_bam + merge_read_lengths) hands over
t = time.perf_counter()
coverage2, _ = build_coverage_device(cols, index); T_cols = time.perf_counter() - t; torch.cuda.synchronize()
assert torch.equal(coverage, coverage2)
t = time.perf_counter()
table = interval_table_from_index(index, base); t = lap("interval_table_native", t)
d_counts, d_offsets = gather_profiles_device(coverage, table); t = lap("table_h2d_plus_gather_f1", t)
res = d.score_profiles(d_counts, d_offsets, 0.428571428571, 5, 0, 0, 0.0); t = lap("score_plus_d2h_outputs", t)
counts, offsets = d_counts.cpu().numpy(), d_offsets.cpu().numpy(); t = lap("d2h_profiles", t)
out_path = os.path.join(tmp, "out.tsv")
with open(out_path, "wb") as fh:
    fh.write(("\t".join(d.COLUMNS) + "\n").encode())
    for chunk in tsv.format_rows_native(counts, offsets, res["phase"], res["valid"], res["read_count"], res["status"], index.tables, True):
        fh.write(chunk)
t = lap("format_f2_plus_write", t)
size = os.path.getsize(out_path)
# the gather kernel alone: interval table already on the device
import ctypes
from ribotricer_amd import _lib
from ribotricer_amd.engine import _ptr
dt = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (table.iv_start, table.iv_len, table.orf_iv, table.reverse, table.offsets)]
d_out = torch.empty(int(table.offsets[-1]), dtype=torch.int32, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def gather_once():
    _lib.check(_lib.load().rp_gather_profiles_dev(0, _ptr(coverage), coverage.numel(), _ptr(dt[0]), _ptr(dt[1]), _ptr(dt[2]), _ptr(dt[3]), _ptr(dt[4]), n, _ptr(d_out), stream))
for _ in range(3):
    gather_once()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    gather_once()
e1.record(); torch.cuda.synchronize()
gather_ms = e0.elapsed_time(e1) / 10
assert torch.equal(d_out, d_counts)
# the gather plan: tile gather and fused gather + score against the two-step path (kernel times, HIP events)
from ribotricer_amd.engine import get_engine, make_filter
from ribotricer_amd.gather import GatherPlan
t0 = time.perf_counter()
gplan = GatherPlan(table, coverage.numel()); torch.cuda.synchronize()
plan_s = time.perf_counter() - t0
eng = get_engine("cuda:0")
th = make_filter()
def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
counts_buf = torch.empty_like(d_counts)
def tile_gather():
    _lib.check(_lib.load().rp_gather_profiles_plan_dev(gplan.handle, _ptr(coverage), coverage.numel(), _ptr(counts_buf), stream))
tile_gather_ms = timed(tile_gather)
assert torch.equal(counts_buf, d_counts)
plain_ms = timed(lambda: eng.score(d_counts, gplan.offsets, thresholds=th, algo="tile", reuse_outputs=True))
fused_ms = timed(lambda: eng.score_coverage(coverage, gplan, thresholds=th, reuse_outputs=True))
tm = []
for _ in range(5):
    eng.score_coverage(coverage, gplan, thresholds=th, reuse_outputs=True, timings=tm)
fused_kernel_ms = float(np.median([x[1] for x in tm]))
tm = []
for _ in range(5):
    eng.score(d_counts, gplan.offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=tm)
plain_kernel_ms = float(np.median([x[1] for x in tm]))
a = eng.score_coverage(coverage, gplan, thresholds=th).cpu_numpy()
b = eng.score(d_counts, gplan.offsets, thresholds=th, algo="tile").cpu_numpy()
assert all(np.array_equal(a[k], b[k]) for k in a)
total = sum(T.values())
# the whole thing once more through the public entry point (report_all=False, the CLI default)
t0 = time.perf_counter()
d.export_orf_coverages(index_path, align, os.path.join(tmp, "x"))
t_export = time.perf_counter() - t0
t0 = time.perf_counter()
d.export_orf_coverages(index_path, cols, os.path.join(tmp, "y"), report_all=True)
t_export_cols_all = time.perf_counter() - t0
t0 = time.perf_counter()
d.export_orf_coverages(index_path, cols, os.path.join(tmp, "z"))
t_export_cols = time.perf_counter() - t0
print(json.dumps({
    "n_orfs": n, "total_nt": int(offsets[-1]), "reads": n_reads, "tsv_bytes_report_all": size,
    "seconds": {k: round(v, 4) for k, v in T.items()}, "total_s": round(total, 3), "orfs_per_s_report_all": round(n / total),
    "export_orf_coverages_default_s": round(t_export, 3), "columns_to_device_coverage_s": round(T_cols, 4),
    "export_orf_coverages_columns_report_all_s": round(t_export_cols_all, 3), "export_orf_coverages_columns_default_s": round(t_export_cols, 3), "gather_kernel_ms": round(gather_ms, 4), "gather_GBps": round(8 * int(offsets[-1]) / gather_ms / 1e6),
    "gather_plan_build_s": round(plan_s, 4), "n_intervals": int(table.iv_start.size),
    "tile_gather_ms": round(tile_gather_ms, 4), "tile_gather_GBps": round(8 * int(offsets[-1]) / tile_gather_ms / 1e6),
    "score_step_ms": round(plain_ms, 4), "score_kernel_ms": round(plain_kernel_ms, 4),
    "fused_step_ms": round(fused_ms, 4), "fused_kernel_ms": round(fused_kernel_ms, 4),
    "gather_plus_score_ms": {"per_orf_gather_then_score": round(gather_ms + plain_ms, 4), "tile_gather_then_score": round(tile_gather_ms + plain_ms, 4), "fused": round(fused_ms, 4)}, "translating": int(res["status"].sum()),
}))
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
