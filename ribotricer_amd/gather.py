"""Device-side profile gather.

The reference fills each ORF's profile with one dict lookup per nucleotide
(``orf_coverage``, ribotricer/detect_orfs.py:134-203).  Here the merged P-site
alignments (``strand -> Counter{(chrom, pos): count}``, what ``merge_read_lengths``
returns, detect_orfs.py:54-83) are laid out ONCE as dense int32 coverage arrays, one per
(strand, chromosome), concatenated in HBM; the ORFs' exon intervals become a small
interval table; ``rp_gather_profiles_dev`` then writes the CSR ``counts`` array for every
ORF in one launch -- reversed for '-' strand ORFs -- ready for ``rp_phase_score_csr_dev``.

Host work is O(reads + intervals) numpy, no per-nucleotide Python.
"""

from __future__ import annotations

import ctypes
from operator import itemgetter
from typing import NamedTuple

import numpy as np
import torch

from . import _lib
from .engine import _as_device, _ptr, get_engine


class IntervalTable(NamedTuple):
    iv_start: np.ndarray  # int64 [n_intervals]  index into the dense coverage array
    iv_len: np.ndarray  # int32 [n_intervals]
    orf_iv: np.ndarray  # int64 [n_orfs + 1]
    reverse: np.ndarray  # uint8 [n_orfs]       1 for '-' strand
    offsets: np.ndarray  # int64 [n_orfs + 1]   CSR offsets of the profiles


def build_dense_coverage(merged_alignments, records):
    """Dense coverage for every (strand, chrom) that an ORF of ``records`` lives on.

    Each array spans the union of the ORFs' extents on that (strand, chrom), so every
    interval position has a slot (0 where no read maps: the missing-key case of
    detect_orfs.py:176-187).  Returns ``(coverage int32[...], base)`` with
    ``base[(strand, chrom)] = (index_of_position_lo, lo)``.
    """
    extent: dict = {}
    for r in records:
        key = (r.strand, r.chrom)
        real = [(s, e) for s, e in r.intervals if e >= s] or [(1, 1)]  # (blocks with end < start hold no position)
        lo, hi = min(s for s, _ in real), max(e for _, e in real)  # (an exon nested in an earlier one ends before it)
        if key in extent:
            e = extent[key]
            extent[key] = (min(e[0], lo), max(e[1], hi))
        else:
            extent[key] = (lo, hi)
    return build_dense_coverage_from_extents(merged_alignments, extent)


def coverage_layout(extent: dict):
    """``base[(strand, chrom)] = (index_of_position_lo, lo)`` and the total length of the
    concatenated dense coverage, groups in sorted key order."""
    base = {}
    total = 0
    for key in sorted(extent):
        lo, hi = extent[key]
        base[key] = (total, lo)
        total += hi - lo + 1
    return base, total


def coverage_entries(merged_alignments, extent: dict, base: dict):
    """The non-zero entries of the dense coverage as ``(index int64[], count int32[])``:
    one per (strand, chrom, pos) key that falls inside a group's extent.  O(reads) numpy."""
    all_idx, all_cnt = [], []
    for strand, table in merged_alignments.items():
        if not table:
            continue
        # the Counter's keys are (chrom, pos) tuples: split them at C speed, then numpy
        keys = list(table.keys())
        pos = np.fromiter(map(itemgetter(1), keys), np.int64, len(keys))
        cnt = np.fromiter(table.values(), np.int64, len(keys))
        names, code = _factorize(list(map(itemgetter(0), keys)))
        start = np.full(len(names), -1, np.int64)
        lo = np.zeros(len(names), np.int64)
        hi = np.full(len(names), -1, np.int64)
        for k, chrom in enumerate(names):
            key = (strand, chrom)
            if key in base:
                start[k], lo[k] = base[key]
                hi[k] = extent[key][1]
        keep = (start[code] >= 0) & (pos >= lo[code]) & (pos <= hi[code])
        code, pos, cnt = code[keep], pos[keep], cnt[keep]
        # the kernels' input contract (include/ribophase.h): 0 <= count <= RP_MAX_COUNT, checked on
        # the int64 values before they are narrowed -- a silent wrap would only show up as wrong scores
        if cnt.size and (int(cnt.min()) < 0 or int(cnt.max()) > _lib.MAX_COUNT):
            bad = int(cnt.min()) if int(cnt.min()) < 0 else int(cnt.max())
            raise _lib.RibophaseError(-7, f"P-site count {bad} on strand {strand!r} outside [0, {_lib.MAX_COUNT}]")
        all_idx.append(start[code] + (pos - lo[code]))
        all_cnt.append(cnt.astype(np.int32))
    if not all_idx:
        return np.zeros(0, np.int64), np.zeros(0, np.int32)
    return np.concatenate(all_idx), np.concatenate(all_cnt)


def build_dense_coverage_from_extents(merged_alignments, extent: dict):
    """Same as :func:`build_dense_coverage`, from ``extent[(strand, chrom)] = (lo, hi)`` (what
    the native index parser reports per group).  Host array (tests and small inputs); the export path
    builds its coverage on the device from alignment columns (``alignments.build_coverage_device``)."""
    base, total = coverage_layout(extent)
    coverage = np.zeros(total, np.int32)
    idx, cnt = coverage_entries(merged_alignments, extent, base)
    coverage[idx] = cnt  # (strand, chrom, pos) keys are unique, so a plain scatter is enough
    return coverage, base


def _factorize(values):
    """(unique values in order of first appearance, int codes) -- hash based."""
    try:
        import pandas as pd

        code, names = pd.factorize(np.asarray(values, dtype=object))
        return list(names), code.astype(np.int64)
    except ImportError:  # pragma: no cover - pandas ships with the image
        seen: dict = {}
        code = np.fromiter((seen.setdefault(v, len(seen)) for v in values), np.int64, len(values))
        return list(seen), code


def build_interval_table(records, base) -> IntervalTable:
    n = len(records)
    n_iv = sum(1 for r in records for s, e in r.intervals if e >= s)  # (a block with end < start holds no position: no interval)
    iv_start = np.empty(n_iv, np.int64)
    iv_len = np.empty(n_iv, np.int32)
    orf_iv = np.zeros(n + 1, np.int64)
    reverse = np.zeros(n, np.uint8)
    lengths = np.zeros(n, np.int64)
    k = 0
    for i, r in enumerate(records):
        start0, lo = base[(r.strand, r.chrom)]
        for s, e in r.intervals:  # ascending (orf.py:100)
            if e < s:
                continue
            iv_start[k] = start0 + (s - lo)
            iv_len[k] = e - s + 1
            lengths[i] += e - s + 1
            k += 1
        orf_iv[i + 1] = k
        reverse[i] = 1 if r.strand == "-" else 0
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    return IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets)


def interval_table_from_index(index, base) -> IntervalTable:
    """IntervalTable of a natively parsed index (``ribotricer_amd.index.NativeIndex``): one native
    pass (``rp_interval_table_host``), no per-ORF Python."""
    keys = index.group_keys
    g_start = np.array([base[k][0] for k in keys], np.int64)
    g_lo = np.array([base[k][1] for k in keys], np.int64)
    n, m = int(index.n_orfs), int(index.iv_start.size)
    iv_start = np.empty(m, np.int64)
    iv_len = np.empty(m, np.int32)
    offsets = np.empty(n + 1, np.int64)
    arrays = [np.ascontiguousarray(a, dtype=t) for a, t in (
        (index.iv_start, np.int64), (index.iv_end, np.int64), (index.orf_iv, np.int64), (index.group, np.int32), (index.length, np.int64))]

    def ptr(a):
        return ctypes.c_void_p(a.ctypes.data if a.size else 0)

    _lib.check(_lib.load().rp_interval_table_host(
        *[ptr(a) for a in arrays], n, m, ptr(g_start), ptr(g_lo), len(keys), ptr(iv_start), ptr(iv_len), ptr(offsets)))
    # (orf_iv and reverse are the index's own arrays: views that keep the parsed index alive, read-only here)
    return IntervalTable(iv_start, iv_len, arrays[2], np.ascontiguousarray(index.reverse, dtype=np.uint8), offsets)


class CompactTable:
    """The interval table of an index in the coordinates of a compact coverage: the fields of :class:`IntervalTable`,
    with ``iv_start`` made on first use -- the one-GPU export never asks (its gather plan is built from the device copy),
    the sharded path and the big-count fixup do.  While the device copy lives (until the gather plan has digested it:
    ``CoverageMap.release_device_intervals``) the host copy is one read-back; afterwards the dense starts go through the
    coverage map once more (``rp_coverage_map_remap_dev``: an upload, one small kernel, a read-back) -- nothing of the
    180 MB (22.8 M exons) stays on the device for a caller that may never come."""

    def __init__(self, dense: IntervalTable, d_iv_start, remap):
        self._d_iv_start = d_iv_start  # (device tensor, dropped once the plan is built or the host copy exists)
        self._dense_iv_start = dense.iv_start  # (host; what the remap starts from)
        self._remap = remap  # dense starts (host int64 array) -> compact starts (host), through the map on the device
        self._iv_start = None
        self.iv_len, self.orf_iv, self.reverse, self.offsets = dense.iv_len, dense.orf_iv, dense.reverse, dense.offsets

    @property
    def iv_start(self) -> np.ndarray:
        if self._iv_start is None:
            self._iv_start = self._d_iv_start.cpu().numpy() if self._d_iv_start is not None else self._remap(self._dense_iv_start)
            self._d_iv_start = self._dense_iv_start = self._remap = None
        return self._iv_start

    def drop_device_copy(self) -> None:
        self._d_iv_start = None


def _alive(ref):
    obj = ref()
    if obj is None:
        raise RuntimeError("the CoverageMap of this compact table is gone: ask for table.iv_start while the map is alive")
    return obj


COVERAGE_BLOCK = 1  # positions per block of a compact coverage (RIBOTRICER_AMD_COVERAGE_BLOCK: a power of two, 1 ... 64)


def coverage_block() -> int:
    import os

    block = int(os.environ.get("RIBOTRICER_AMD_COVERAGE_BLOCK", COVERAGE_BLOCK))
    if block < 1 or block > 64 or block & (block - 1):
        raise ValueError(f"RIBOTRICER_AMD_COVERAGE_BLOCK must be a power of two from 1 to 64, got {block}")
    return block


class CoverageMap:
    """Block map of a COMPACT coverage (``rp_coverage_map_create_dev``): of the dense layout only the blocks of
    ``block_positions`` positions under an exon interval get a slot -- the reference never looks anything else up
    (detect_orfs.py:176-187) -- so the coverage of a human-sized index takes a tenth of the dense array's memory,
    allocation time and memset.  With one-position blocks (the default) the coverage holds exonic positions only: exons
    that face each other across an intron become neighbours, a spliced ORF's pieces merge into one run of the gather
    plan, and the fused kernel stops paying for the cache lines at both ends of every exon.  Built once per index (it
    depends on the interval table only); ``table`` is the interval table in COMPACT coordinates (an interval stays
    contiguous), ``compact_len`` the coverage length, ``dense_len`` what the alignment rows' layout spans.  Owns its
    device memory (dense_len / 4 bytes for one-position blocks, dense_len / 256 for 64-position ones)."""

    def __init__(self, dense_table: IntervalTable, dense_len: int, device=None, block_positions: int | None = None):
        dev = get_engine(device).device
        self.device = dev
        self.dense_len = int(dense_len)
        self.block_positions = int(block_positions) if block_positions is not None else coverage_block()
        nbytes = ctypes.c_size_t(0)
        _lib.check(_lib.load().rp_coverage_map_bytes(self.dense_len, self.block_positions, ctypes.byref(nbytes)))
        self._mem = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        iv_start = _as_device(dense_table.iv_start, torch.int64, dev).clone()  # (rewritten in place: dense -> compact)
        iv_len = _as_device(dense_table.iv_len, torch.int32, dev)
        compact = ctypes.c_int64(0)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.load().rp_coverage_map_create_dev(dev.index, _ptr(iv_start), _ptr(iv_len), iv_start.numel(), self.dense_len, self.block_positions,
                                                         _ptr(self._mem), self._mem.numel(), stream, ctypes.byref(compact)))
        self.compact_len = int(compact.value)
        import weakref

        me = weakref.ref(self)  # (no cycle: the table must not keep the map -- and its dense_len / 4 bytes of device memory -- alive)
        self.table = CompactTable(dense_table, iv_start, lambda dense: _alive(me).remap(dense))
        self.device_intervals = (iv_start, iv_len)  # (for the gather plan that follows; dropped by release_device_intervals)

    def release_device_intervals(self) -> None:
        """The gather plan has digested the device copies of the interval table (0.3 GB for 22.8 M exons): drop them,
        the compact starts' copy inside ``table`` included -- its host copy, if anybody asks, is re-made through the map."""
        self.device_intervals = None
        self.table.drop_device_copy()

    def remap(self, dense_positions: np.ndarray) -> np.ndarray:
        """Dense-layout positions that lie under an exon (interval starts) -> their slots in the compact coverage."""
        d = _as_device(np.ascontiguousarray(dense_positions, dtype=np.int64), torch.int64, self.device).clone()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().rp_coverage_map_remap_dev(self.device.index, _ptr(d), d.numel(), _ptr(self._mem), self.dense_len,
                                                        self.block_positions, stream))
        return d.cpu().numpy()

    @property
    def ptr(self):
        return _ptr(self._mem)


class GatherPlan:
    """A gather plan (``rp_gather_plan``): the profile space of an index as pieces of the dense
    coverage, one fixed-stride row of clipped pieces per tile.  Depends on the interval table
    (index + coverage layout) only -- built once per index, reused for every sample.  Owns its
    device memory.  Raises ``RibophaseError`` with status ``_lib.ERR_INTERVALS`` for a table with
    an empty or off-array interval (the per-ORF kernel behind :func:`gather_profiles_device`
    handles those)."""

    def __init__(self, table: IntervalTable, coverage_len: int, device=None, device_intervals=None):
        """``device_intervals``: ``(iv_start int64, iv_len int32)`` of ``table`` already on the device (a
        :class:`CoverageMap` has them: no second upload)."""
        dev = get_engine(device).device
        self.device = dev
        self.n_orfs = int(len(table.offsets) - 1)
        self.n_intervals = int(len(table.iv_len))
        self.total_nt = int(table.offsets[-1])
        self.coverage_len = int(coverage_len)
        self.offsets = _as_device(table.offsets, torch.int64, dev)
        iv_start = _as_device(table.iv_start if device_intervals is None else device_intervals[0], torch.int64, dev)
        iv_len = _as_device(table.iv_len if device_intervals is None else device_intervals[1], torch.int32, dev)
        orf_iv = _as_device(table.orf_iv, torch.int64, dev)
        reverse = _as_device(table.reverse, torch.uint8, dev)
        self._mem = torch.empty(_lib.gather_plan_bytes(self.n_orfs, self.n_intervals, self.total_nt), dtype=torch.uint8, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        handle = ctypes.c_void_p(0)
        _lib.check(
            _lib.load().rp_gather_plan_create_dev(
                dev.index, _ptr(iv_start), _ptr(iv_len), _ptr(orf_iv), _ptr(reverse), _ptr(self.offsets), self.n_orfs,
                self.n_intervals, self.total_nt, self.coverage_len, _ptr(self._mem), self._mem.numel(), stream,
                ctypes.byref(handle),
            )
        )
        self.handle = handle

    def gather(self, coverage) -> torch.Tensor:
        """The CSR counts array of the whole index (``rp_gather_profiles_plan_dev``)."""
        cov = _as_device(coverage, torch.int32, self.device)
        counts = torch.empty(self.total_nt, dtype=torch.int32, device=self.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().rp_gather_profiles_plan_dev(self.handle, _ptr(cov), cov.numel(), _ptr(counts), stream))
        return counts

    def gather_selected(self, coverage, keep: np.ndarray, lengths: np.ndarray = None, reuse_arrays: bool = False):
        """Profiles of the ORFs with ``keep[i]`` true (host bool array of n_orfs), packed one after the other on the
        device: ``(counts int32 device tensor, offsets int64 host array [n_orfs + 1])`` where every other ORF has an
        empty range -- what the TSV writer takes in default mode (only translating ORFs are printed,
        detect_orfs.py:301-303).  One wave per chosen ORF copies its pieces (``rp_gather_selected_plan_dev``); the ids, the
        packed starts and the all-ORF offsets come from one native pass on the host (``rp_select_profiles_host``; torch ops would
        do them in a millisecond, but each op's first use in a process costs ~0.1 s of code loading on ROCm, and a
        single-sample run is the common case).
        ``lengths``: profile lengths of all ORFs (default: from this plan's offsets, one device read-back).
        ``reuse_arrays``: the returned ``offsets`` is this plan's scratch array, overwritten by its next call (the export
        is done with one sample's offsets before it scores the next); default: a copy."""
        cov = _as_device(coverage, torch.int32, self.device)
        keep = np.ascontiguousarray(keep, dtype=np.bool_)
        if lengths is None:
            lengths = np.diff(self.offsets.cpu().numpy())
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        # ids, packed starts and the all-ORF offsets in one native pass (rp_select_profiles_host); the three arrays are kept
        # for the next sample of this plan (88 MB each for 11 M ORFs: a fresh one costs more in page faults than the pass)
        scratch = self.__dict__.setdefault("_select_scratch", {})
        if scratch.get("n") != self.n_orfs:
            scratch.update(n=self.n_orfs, chosen=np.empty(self.n_orfs, np.int64), chosen_off=np.empty(self.n_orfs + 1, np.int64),
                           offsets=np.empty(self.n_orfs + 1, np.int64))
        n_chosen = ctypes.c_int64(0)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        _lib.check(_lib.load().rp_select_profiles_host(ptr(keep.view(np.uint8)), ptr(lengths), self.n_orfs, ptr(scratch["chosen"]),
                                                      ptr(scratch["chosen_off"]), ptr(scratch["offsets"]), ctypes.byref(n_chosen)))
        k = int(n_chosen.value)
        chosen, out_off, offsets = scratch["chosen"][:k], scratch["chosen_off"][: k + 1], scratch["offsets"]
        counts = torch.empty(int(out_off[-1]), dtype=torch.int32, device=self.device)
        d_chosen = torch.from_numpy(chosen).to(self.device)
        d_off = torch.from_numpy(out_off[:-1].copy()).to(self.device)  # (a temporary: a blocking copy)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().rp_gather_selected_plan_dev(self.handle, _ptr(cov), cov.numel(), _ptr(d_chosen), d_chosen.numel(),
                                                          _ptr(d_off), _ptr(counts), stream))
        return counts, (offsets if reuse_arrays else offsets.copy())

    def stats(self) -> dict:
        """Diagnostics (scripts): how the tiles of this plan are staged -- chunk rows per tile (<= 64
        positions of one run each; 256 fit a row -- less the padding of the two directions' regions to blocks of 16 --,
        more take the scalar slow path), read back from the plan memory
        (layout: csrc/rp_pieces.hpp ``carve_piece_plan`` behind the 128-byte header)."""
        tile = _lib.tile_positions(self.n_orfs, self.total_nt)
        n_tiles = max(1, -(-self.total_nt // tile))
        up = lambda b: (b + 127) & ~127  # noqa: E731
        at = 128 + 2 * up((self.n_intervals + 1) * 8) + up((self.n_orfs + 1) * 8) + up(n_tiles * 8)
        torch.cuda.synchronize(self.device)
        words = self._mem[at : at + 16 * n_tiles].cpu().numpy().view(np.int64).reshape(n_tiles, 2)
        lo, chunks = words[:, 0], (words[:, 1] & 0xFFFF) + ((words[:, 1] >> 16) & 0xFFFF)  # forward + '-' strand chunks
        wide = ((words[:, 1] >> 32) & 1) != 0  # a chunk of the tile lies > 4 GiB from the tile's lowest source: one more instruction per chunk
        slow = lo == np.iinfo(np.int64).min
        q = np.percentile(chunks, [50, 90, 99]) if n_tiles else [0, 0, 0]
        return {"tile_positions": tile, "tiles": int(n_tiles), "slow_tiles": int(slow.sum()), "chunks_per_tile_mean": float(chunks.mean()),
                "chunks_per_tile_p50_p90_p99": [float(x) for x in q], "chunks_per_tile_max": int(chunks.max()),
                "tiles_over_256_chunks": int((chunks > 256).sum()), "wide_tiles": int((wide & ~slow).sum()), "positions_per_chunk_mean": float(self.total_nt / max(1, chunks[~slow].sum())),
                "pieces_per_tile_mean": float(self.n_intervals / n_tiles)}

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _lib.load().rp_gather_plan_free(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass


def make_gather_plan(table: IntervalTable, coverage_len: int, device=None, device_intervals=None):
    """:class:`GatherPlan`, or None when the table cannot be planned (an interval hangs off the
    coverage array: the per-ORF kernel reads such positions as 0)."""
    try:
        return GatherPlan(table, coverage_len, device, device_intervals=device_intervals)
    except _lib.RibophaseError as e:
        if e.status == _lib.ERR_INTERVALS:
            return None
        raise


def select_orfs(table: IntervalTable, orf_ids: np.ndarray) -> IntervalTable:
    """The interval table of a subset of the ORFs (in the given order), offsets re-packed."""
    orf_ids = np.asarray(orf_ids, np.int64)
    k0, k1 = table.orf_iv[orf_ids], table.orf_iv[orf_ids + 1]
    nk = k1 - k0
    orf_iv = np.zeros(orf_ids.size + 1, np.int64)
    np.cumsum(nk, out=orf_iv[1:])
    pick = np.repeat(k0 - orf_iv[:-1], nk) + np.arange(int(orf_iv[-1]), dtype=np.int64)
    lengths = table.offsets[orf_ids + 1] - table.offsets[orf_ids]
    offsets = np.zeros(orf_ids.size + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    return IntervalTable(table.iv_start[pick], table.iv_len[pick], orf_iv, table.reverse[orf_ids], offsets)


def slice_orfs(table, lo: int, hi: int) -> IntervalTable:
    """The interval table of ORFs [lo, hi): views of the table's arrays (no copy of the intervals), offsets and interval
    ranges re-based -- what :func:`select_orfs` returns for ``np.arange(lo, hi)`` without its fancy indexing."""
    orf_iv = np.asarray(table.orf_iv, np.int64)
    offsets = np.asarray(table.offsets, np.int64)
    k0, k1 = int(orf_iv[lo]), int(orf_iv[hi])
    return IntervalTable(table.iv_start[k0:k1], table.iv_len[k0:k1], orf_iv[lo : hi + 1] - k0, table.reverse[lo:hi], offsets[lo : hi + 1] - offsets[lo])


def orfs_touching(table: IntervalTable, positions: np.ndarray) -> np.ndarray:
    """Sorted ids of the ORFs with an exon interval that contains one of ``positions`` (coverage indices):
    who holds the saturated positions ``alignments.build_coverage_device`` reported.  Two binary searches per
    interval -- a rare path (no such position on ordinary samples)."""
    positions = np.sort(np.asarray(positions, np.int64))
    if positions.size == 0 or len(table.iv_start) == 0:
        return np.zeros(0, np.int64)
    start = np.asarray(table.iv_start, np.int64)
    end = start + np.asarray(table.iv_len, np.int64)
    hit = np.searchsorted(positions, end, side="left") > np.searchsorted(positions, start, side="left")
    iv = np.flatnonzero(hit)
    return np.unique(np.searchsorted(np.asarray(table.orf_iv, np.int64), iv, side="right") - 1)


def gather_profiles_device(coverage, table: IntervalTable, device=None, plan=None):
    """Run the gather on the GPU: ``(counts int32 device tensor, offsets int64 device tensor)``.
    With a :class:`GatherPlan` of the same table: the tile kernel; otherwise one wave per ORF."""
    if plan is not None:
        return plan.gather(coverage), plan.offsets
    eng = get_engine(device)
    dev = eng.device
    cov = _as_device(coverage, torch.int32, dev)
    iv_start = _as_device(table.iv_start, torch.int64, dev)
    iv_len = _as_device(table.iv_len, torch.int32, dev)
    orf_iv = _as_device(table.orf_iv, torch.int64, dev)
    reverse = _as_device(table.reverse, torch.uint8, dev)
    offsets = _as_device(table.offsets, torch.int64, dev)
    n = offsets.numel() - 1
    counts = torch.empty(int(table.offsets[-1]), dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(
        _lib.load().rp_gather_profiles_dev(
            dev.index, _ptr(cov), cov.numel(), _ptr(iv_start), _ptr(iv_len), _ptr(orf_iv), _ptr(reverse),
            _ptr(offsets), n, _ptr(counts), stream,
        )
    )
    return counts, offsets


def coverage_profiles_of(coverage, table: IntervalTable, device=None):
    """``profiles_of`` for ``engine.resolve_big_ties`` over an interval table: the chosen ORFs'
    profiles gathered on the device (per-ORF kernel) and copied to the host."""

    def fetch(idx):
        counts, offsets = gather_profiles_device(coverage, select_orfs(table, idx), device)
        return counts.cpu().numpy(), offsets.cpu().numpy()

    return fetch
