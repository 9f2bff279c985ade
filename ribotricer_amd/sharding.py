"""Sharding of a candidate-ORF batch over the GPUs of one node.

ORFs are independent (the loop body of ribotricer/detect_orfs.py:274-324 carries no
state between iterations besides the output order), so the batch is cut into
contiguous ORF-index slices balanced on the prefix sum of profile lengths (nucleotides,
not ORF counts), each rank scores its slice on its own GPU, and the per-ORF result
arrays are concatenated on the host in slice order.  No collective is on the data
path; ``gather_results`` moves 26 bytes per ORF at the very end.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np


def slice_bounds(offsets: np.ndarray, world_size: int) -> np.ndarray:
    """ORF-index cut points ``b[0]=0 <= b[1] <= ... <= b[world]=n_orfs``, nt-balanced.

    Cut r is the first ORF whose start offset is >= r/world of the total length; long
    ORFs are never split across ranks.
    """
    offsets = np.asarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    total = int(offsets[-1])
    targets = (np.arange(1, world_size, dtype=np.float64) * total / world_size).astype(np.int64)
    cuts = np.searchsorted(offsets[:-1], targets, side="left") if n > 0 else np.zeros(world_size - 1, np.int64)
    if total == 0:  # only empty profiles: balance on ORF count instead
        cuts = (np.arange(1, world_size) * n) // world_size
    bounds = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def shard_csr(counts, offsets, world_size: int, rank: int):
    """Rank ``rank``'s slice as its own CSR batch: ``(counts_r, offsets_r, first_orf, last_orf)``.

    Works on numpy arrays and on torch tensors (views, no copy of ``counts``).
    """
    off_np = offsets.cpu().numpy() if hasattr(offsets, "cpu") else np.asarray(offsets)
    bounds = slice_bounds(off_np, world_size)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    a, b = int(off_np[lo]), int(off_np[hi])
    return counts[a:b], offsets[lo : hi + 1] - offsets[lo], lo, hi


def concat_results(parts: Sequence[dict]) -> dict:
    """Host-side concat of per-rank result dicts (numpy arrays) in slice order."""
    keys = [k for k, v in parts[0].items() if v is not None]
    return {k: np.concatenate([p[k] for p in parts]) for k in keys}


def gather_results(local: dict, group=None) -> dict:
    """All ranks receive the concatenated results (torch.distributed, any backend)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, {k: v for k, v in local.items() if v is not None}, group=group)
    return concat_results(parts)


def coverage_windows(iv_start: np.ndarray, iv_len: np.ndarray, gap: int = 1 << 20):
    """The part of the dense coverage a set of exon intervals touches, as a few WINDOWS.

    A slice of the candidate-ORF index reads only the coverage under its own intervals -- on a
    human-sized index one eighth of 25 GB, in two stretches (the slice's transcripts on the '+'
    and on the '-' arrays) -- so a device that scores the slice needs those windows, not the whole
    array.  Intervals are merged into windows wherever they lie closer than ``gap`` positions.

    Returns ``(win_start int64[w], win_len int64[w], win_base int64[w], total)``: window k covers
    coverage ``[win_start[k], win_start[k] + win_len[k])`` and sits at ``win_base[k]`` of the
    compacted array of ``total`` positions (windows in ascending order, 16-position aligned so
    that 16-byte aligned staging stays aligned).
    """
    iv_start = np.asarray(iv_start, np.int64)
    iv_len = np.asarray(iv_len, np.int64)
    if iv_start.size == 0:
        z = np.zeros(0, np.int64)
        return z, z, z, 0
    order = np.argsort(iv_start, kind="stable")
    s = iv_start[order]
    e = s + iv_len[order]
    reach = np.maximum.accumulate(e)
    new = np.ones(s.size, bool)
    new[1:] = s[1:] > reach[:-1] + gap
    first = np.flatnonzero(new)
    win_start = (s[first] // 16) * 16
    last_reach = np.concatenate([reach[first[1:] - 1], reach[-1:]])
    win_len = ((last_reach - win_start + 15) // 16) * 16
    win_base = np.zeros(first.size, np.int64)
    np.cumsum(win_len[:-1], out=win_base[1:])
    return win_start, win_len, win_base, int(win_len.sum())


def coverage_windows_native(iv_start: np.ndarray, iv_len: np.ndarray, gap_shift: int = 20):
    """:func:`coverage_windows` with ``gap = 2 ** gap_shift`` and :func:`remap_to_windows` in one native pass over the
    intervals (``rp_coverage_windows_host``: no sort -- a block of ``gap`` positions can hold only one window): ``(win_start,
    win_len, win_base, total, rebased iv_start)``.  22.8 M intervals: 0.1 s instead of 2.8 s of numpy (argsort + searchsorted
    per slice), which was most of a sharded export's first sample."""
    import ctypes

    from . import _lib

    iv_start = np.ascontiguousarray(iv_start, dtype=np.int64)
    iv_len = np.ascontiguousarray(iv_len, dtype=np.int32)
    rebased = np.empty_like(iv_start)
    if iv_start.size == 0:
        z = np.zeros(0, np.int64)
        return z, z, z, 0, rebased
    cap = 1024
    while True:
        ws, wl, wb = (np.empty(cap, np.int64) for _ in range(3))
        n_w, total = ctypes.c_int64(0), ctypes.c_int64(0)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        rc = _lib.load().rp_coverage_windows_host(ptr(iv_start), ptr(iv_len), iv_start.size, gap_shift, ptr(ws), ptr(wl), ptr(wb), cap,
                                                  ctypes.byref(n_w), ctypes.byref(total), ptr(rebased))
        if rc == _lib.ERR_SIZE and n_w.value > cap:
            cap = int(n_w.value)
            continue
        _lib.check(rc)
        k = int(n_w.value)
        return ws[:k].copy(), wl[:k].copy(), wb[:k].copy(), int(total.value), rebased


def remap_to_windows(iv_start: np.ndarray, win_start: np.ndarray, win_base: np.ndarray) -> np.ndarray:
    """Interval starts in the compacted coverage of :func:`coverage_windows`."""
    iv_start = np.asarray(iv_start, np.int64)
    if iv_start.size == 0:
        return iv_start.copy()
    k = np.searchsorted(win_start, iv_start, side="right") - 1
    return iv_start - win_start[k] + win_base[k]


def compact_coverage(coverage, win_start, win_len, win_base, total: int, device=None):
    """The compacted coverage of a slice on ``device``: window k of ``coverage`` (a host array or a
    tensor on any device) copied to ``[win_base[k], win_base[k] + win_len[k])``; reads past the end
    of ``coverage`` (the 16-position rounding of the last window) stay 0."""
    import torch

    is_tensor = isinstance(coverage, torch.Tensor)
    n = coverage.numel() if is_tensor else int(np.asarray(coverage).size)
    dev = torch.device(device) if device is not None else (coverage.device if is_tensor else torch.device("cpu"))
    out = torch.zeros(total, dtype=torch.int32, device=dev)
    for a, ln, at in zip(win_start.tolist(), win_len.tolist(), win_base.tolist()):
        b = min(a + ln, n)
        if b <= a:
            continue
        part = coverage[a:b]
        if not is_tensor:
            part = torch.from_numpy(np.ascontiguousarray(part, dtype=np.int32))
        out[at : at + (b - a)].copy_(part, non_blocking=True)
    return out
