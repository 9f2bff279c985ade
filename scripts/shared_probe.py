#!/usr/bin/env python3
"""Speed probe of the LDS stage of a position-major scorer for nested indexes (csrc/shared_probe.hip; the design and the
question: profiles/r06_prefix_prototype.txt).  Builds the plan on the host with numpy -- the elementary intervals of every tile
of the compact coverage and one segment per (ORF piece x tile) -- runs the probe kernel over the nested layout of bench.py's
`fused_nested` section at full size, times it, and checks its sums on the ORFs it covers completely (one piece, inside one
tile, L % 3 == 0: N_f, M_f, read count and codon minimum against numpy; the unit-vector sums against float64 within 1e-5).

usage: python scripts/shared_probe.py [n_orfs] [out.json]"""
import ctypes
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

KT = 6912  # csrc/shared_probe.hip: positions per tile
n_set = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
seed = 20260213
dev = torch.device("cuda", 0)
eng = get_engine(dev)
iv_start, iv_len, orf_iv, reverse, offsets, dense_len = synth_nested_layout(n_set, seed)
table = IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets)
cmap = CoverageMap(table, dense_len, dev)
ctab, n_pos = cmap.table, int(cmap.compact_len)
s = np.asarray(ctab.iv_start, np.int64)
ln = np.asarray(ctab.iv_len, np.int64)
cmap.release_device_intervals()
cov = synth_coverage_device(n_pos, seed, device=dev)
n = offsets.size - 1
orf_iv = np.asarray(orf_iv, np.int64)
rep = {"orfs": n, "pieces": int(s.size), "profile_nt": int(offsets[-1]), "compact_positions": n_pos, "tile_positions": KT}

# ---- today's fused step on the same layout (default mode of the export), for the ratio -------------------------------------
gplan = GatherPlan(ctab, n_pos, dev)
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

# ---- the plan (host, numpy) --------------------------------------------------------------------------------------------------
t_plan = time.perf_counter()
n_pieces = np.diff(orf_iv)
orf_of = np.repeat(np.arange(n, dtype=np.int64), n_pieces)
e = s + ln
csum = np.cumsum(ln)
before = csum - ln - np.repeat((csum - ln)[orf_iv[:-1]], n_pieces)  # profile nt of the ORF in front of the piece ('+' order)
total = np.repeat(np.diff(offsets), n_pieces)
after = total - before - ln  # ... behind it: in front of it in a '-' strand profile
rev_p = np.repeat(np.asarray(reverse, np.uint8), n_pieces).astype(bool)
c0 = np.where(rev_p, (after + e) % 3, (s - before) % 3).astype(np.uint32)
has = ln >= 3  # (shorter pieces hold no whole codon of their own)
n_tiles = (n_pos + KT - 1) // KT
tile_starts = np.arange(n_tiles + 1, dtype=np.int64) * KT
bg = np.unique(np.concatenate([s[has], e[has] - 2, tile_starts]))
bg = bg[bg <= n_tiles * KT]
tile_b_off = np.searchsorted(bg, tile_starts).astype(np.uint32)  # tile t: bg[tile_b_off[t] : tile_b_off[t + 1]]
bounds = (bg[: int(tile_b_off[-1])] % KT).astype(np.uint16)
per_tile = np.diff(tile_b_off.astype(np.int64))
# segments: (piece x tile), in ORF order
ps, pe2 = s[has], e[has] - 2
t_first, t_last = ps // KT, (pe2 - 1) // KT
n_seg_of = (t_last - t_first + 1).astype(np.int64)
seg_piece = np.repeat(np.arange(ps.size, dtype=np.int64), n_seg_of)
seg_tile = np.repeat(t_first, n_seg_of) + (np.arange(seg_piece.size, dtype=np.int64) - np.repeat(np.cumsum(n_seg_of) - n_seg_of, n_seg_of))
a = np.maximum(ps[seg_piece], seg_tile * KT)
bnd = np.minimum(pe2[seg_piece], (seg_tile + 1) * KT)
ia = (np.searchsorted(bg, a) - tile_b_off[seg_tile].astype(np.int64)).astype(np.uint16)
ib = (np.searchsorted(bg, bnd) - tile_b_off[seg_tile].astype(np.int64)).astype(np.uint16)
n_seg = int(seg_piece.size)
seg_dtype = np.dtype([("ia", "<u2"), ("ib", "<u2"), ("slot", "<u4"), ("c0_dir", "<u4")])
segs = np.empty(n_seg, seg_dtype)
segs["ia"], segs["ib"], segs["slot"] = ia, ib, np.arange(n_seg, dtype=np.uint32)
segs["c0_dir"] = c0[has][seg_piece] | (rev_p[has][seg_piece].astype(np.uint32) << 2)
order = np.argsort(seg_tile, kind="stable")
segs_by_tile = segs[order]
tile_s_off = np.searchsorted(seg_tile[order], np.arange(n_tiles + 1, dtype=np.int64)).astype(np.uint32)
tile_rev = np.zeros(n_tiles, np.uint8)
first_seg = np.minimum(tile_s_off[:-1].astype(np.int64), max(0, n_seg - 1))
tile_rev[:] = (segs_by_tile["c0_dir"][first_seg] >> 2) & 1
rep["plan"] = {"host_build_s": time.perf_counter() - t_plan, "tiles": int(n_tiles), "boundaries": int(bounds.size), "intervals_per_tile_mean": float(per_tile.mean()),
               "intervals_per_tile_p99": float(np.percentile(per_tile, 99)), "intervals_per_tile_max": int(per_tile.max()),
               "tiles_over_192_intervals": int((per_tile > 192).sum()), "segments": n_seg, "segments_per_orf": n_seg / n,
               "plan_bytes_read_per_launch": int(bounds.nbytes + segs_by_tile.nbytes + tile_b_off.nbytes + tile_s_off.nbytes + tile_rev.nbytes),
               "intervals_folded_per_segment_mean": float((ib.astype(np.int64) - ia).mean())}

# ---- run -----------------------------------------------------------------------------------------------------------------
lib = ctypes.CDLL(os.path.join(REPO, "ribotricer_amd", "csrc", "libstreamprobe.so"))
lib.sp_shared_probe.restype = ctypes.c_int
lib.sp_shared_probe.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong] + [ctypes.c_void_p] * 5 + [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype.names else v.view(np.uint8)).to(dev) for k, v in
     dict(tile_b_off=tile_b_off, bounds=bounds, tile_s_off=tile_s_off, segs=segs_by_tile, tile_rev=tile_rev).items()}
n_rec = (n_seg + 15) // 16 * 16
rec = torch.zeros(3 * n_rec * 4, dtype=torch.int32, device=dev)
skipped = torch.zeros(1, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)


def launch():
    rc = lib.sp_shared_probe(cov.data_ptr(), n_pos, n_tiles, d["tile_b_off"].data_ptr(), d["bounds"].data_ptr(), d["tile_s_off"].data_ptr(),
                             d["segs"].data_ptr(), d["tile_rev"].data_ptr(), rec.data_ptr(), n_rec, skipped.data_ptr(), stream.cuda_stream)
    assert rc == 0, rc


for _ in range(3):
    launch()
torch.cuda.synchronize(dev)
times = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    skipped.zero_()
    e0.record(stream)
    launch()
    e1.record(stream)
    torch.cuda.synchronize(dev)
    times.append(e0.elapsed_time(e1))
k_ms = statistics.median(times)
rep["probe_kernel"] = {"ms": k_ms, "min_ms": min(times), "tiles_skipped": int(skipped.item()), "coverage_GBps": 4 * n_pos / (k_ms * 1e-3) / 1e9,
                       "record_bytes_written": 48 * n_seg}

# ---- sanity: ORFs the probe covers completely (one piece of >= 3 nt inside one tile, L % 3 == 0) ----------------------------
single = np.flatnonzero((n_pieces == 1) & (np.diff(offsets) % 3 == 0))
pk = orf_iv[single]
inside = (s[pk] // KT) == ((e[pk] - 1 + 2) // KT)  # the halo of the last codons too
skipped_tiles = per_tile > 192
pick = single[inside & ~skipped_tiles[s[pk] // KT]]
rng = np.random.default_rng(1)
pick = rng.choice(pick, size=min(3000, pick.size), replace=False)
piece_to_seg = np.full(s.size, -1, np.int64)
piece_to_seg[np.flatnonzero(has)[seg_piece]] = np.arange(n_seg)  # (one segment for these pieces)
rec_h = rec.view(torch.int32).reshape(3, n_rec, 4)
bad = 0
worst = 0.0
for i in pick.tolist():
    k = int(orf_iv[i])
    prof = cov[int(s[k]) : int(e[k])].cpu().numpy().astype(np.int64)
    if reverse[i]:
        prof = prof[::-1]
    g = int(piece_to_seg[k])
    r = rec_h[:, g, :].cpu().numpy()
    for f in range(3):
        cod = prof[f : f + 3 * ((prof.size - f) // 3)].reshape(-1, 3)
        a_, b_, c_ = cod[:, 0].astype(np.float64), cod[:, 1].astype(np.float64), cod[:, 2].astype(np.float64)
        x2, y = 2 * a_ - b_ - c_, b_ - c_
        q = x2 * x2 + 3 * y * y
        use = q > 0
        rr = np.where(use, 1 / np.sqrt(np.where(use, q, 1)), 0)
        P, Q = float((x2 * rr).sum()), float((np.sqrt(3.0) * y * rr).sum())
        N, M = int((cod.sum(1) != 0).sum()), int(use.sum())
        gp, gq = np.array([r[f, 0], r[f, 1]], np.int32).view(np.float32)
        nm = int(np.uint32(r[f, 2]))
        ok = (nm & 0xFFFF) == N and (nm >> 16) == M and abs(gp - P) <= 1e-5 * max(1, M) and abs(gq - Q) <= 1e-5 * max(1, M)
        worst = max(worst, abs(gp - P), abs(gq - Q))
        if f == 0:
            count = (int(np.uint32(r[2, 3])) << 16) + int(np.uint32(r[0, 3]))
            ok = ok and count == int(prof.sum()) and int(np.uint32(r[1, 3])) == int(cod.sum(1).min())
        bad += not ok
rep["sanity"] = {"orfs_checked": int(pick.size), "frames_wrong": int(bad), "max_abs_dPQ": float(worst)}

# ---- the estimate ------------------------------------------------------------------------------------------------------------
fin = rep["fused_today"]["finish_ms"]
fin_new = fin * (48.0 * n_seg + 33.0 * n) / (48.0 * n * 1.15 + 33.0 * n)  # the finish pass reads n_seg records instead of ~1.15 n
rep["estimate"] = {"probe_kernel_ms": k_ms, "finish_ms_scaled_to_its_records": fin_new, "step_ms": k_ms + fin_new,
                   "gain_over_today": rep["fused_today"]["step_ms"] / (k_ms + fin_new),
                   "left_out": "junction codons of spliced ORFs (~2.2 per ORF, scattered loads), partial last codons, strand changes inside a tile, tiles over 192 intervals"}
print(json.dumps(rep, indent=1))
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(rep, fh, indent=1)
