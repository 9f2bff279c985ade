"""Columnar P-site alignments: the hand-over format between the BAM
front end and the profile gather, without per-key Python.

The reference keeps the 5'-end histogram of a BAM as ``alignments[length][strand][(chrom, pos)]
-> count`` (``split_bam``, ribotricer/bam.py:33-153), shifts every key by its read length's
P-site offset and adds the read lengths up in a ``strand -> Counter`` (``merge_read_lengths``,
ribotricer/detect_orfs.py:54-83), and ``export_orf_coverages`` then looks every nucleotide up
in that Counter.  Here the histogram is five numpy columns; merging is a vectorised shift (the
adding-up happens on the device, ``rp_coverage_build_dev``), and the reference's nested dicts
are accepted everywhere through one conversion pass (the compatibility shim).
"""

from __future__ import annotations

import ctypes
import os
from collections import Counter, defaultdict
from operator import itemgetter
from typing import NamedTuple

import numpy as np

from . import _lib

STRANDS = ("+", "-")


class OtherStrand(tuple):
    """``(strand, chrom)`` standing in for a chromosome name in a ``chroms`` list: rows filed under a strand key other
    than '+' / '-'.  split_bam never makes one (bam.py:127-133), but the reference looks ``alignments[strand]`` up with
    whatever the index line's strand column says (detect_orfs.py:176-187), so a hand-made mapping with such a key has
    defined results.  The rows carry strand code 0 and this pair as their chromosome: every (strand, chrom) -> group
    lookup below goes through :func:`group_lut`, which knows the convention; nothing on the device changes."""

    __slots__ = ()


def group_lut(chroms, group_keys) -> np.ndarray:
    """``lut[strand_code, chrom_code]`` = group number of a candidate-ORF index (``group_keys`` = its (strand, chrom)
    pairs), -1 where no ORF lives.  Shape (2, max(1, len(chroms)))."""
    lut = np.full((2, max(1, len(chroms))), -1, np.int32)
    where = {name: k for k, name in enumerate(chroms)}
    for g, (strand, chrom) in enumerate(group_keys):
        if strand in STRANDS:
            k = where.get(chrom)
            if k is not None and not isinstance(chroms[k], OtherStrand):
                lut[STRANDS.index(strand), k] = g
        else:
            k = where.get(OtherStrand((strand, chrom)))
            if k is not None and isinstance(chroms[k], OtherStrand):
                lut[0, k] = g
    return lut


def _factorize(values):
    """(unique values in order of first appearance, int codes) -- hash based."""
    seen: dict = {}
    try:
        import pandas as pd

        code, names = pd.factorize(np.asarray(values, dtype=object))
        return list(names), code.astype(np.int32)
    except ImportError:  # pragma: no cover - pandas ships with the image
        code = np.fromiter((seen.setdefault(v, len(seen)) for v in values), np.int32, len(values))
        return list(seen), code


_PYDICT = [False, None]


def _pydict_reader():
    """``rp_counter_columns`` of csrc/libribopy.so (built where Python.h is installed), or None."""
    if not _PYDICT[0]:
        _PYDICT[0] = True
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libribopy.so")
        if os.path.exists(path):
            try:
                fn = ctypes.PyDLL(path).rp_counter_columns
                fn.restype = ctypes.c_longlong
                fn.argtypes = [ctypes.py_object, ctypes.py_object, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
                _PYDICT[1] = fn
            except (OSError, AttributeError):  # pragma: no cover
                _PYDICT[1] = None
    return _PYDICT[1]


class MergedColumns(NamedTuple):
    """What ``merge_read_lengths`` returns, as columns: one row per (read length, strand, chrom,
    shifted pos) key -- rows of different read lengths may name the same position; they add up."""

    strand: np.ndarray  # uint8  0 = '+', 1 = '-'
    chrom: np.ndarray  # int32  index into ``chroms``
    pos: np.ndarray  # int64  1-based, shifted by the P-site offset (detect_orfs.py:76-80)
    count: np.ndarray  # int64
    chroms: list  # chromosome names

    @classmethod
    def from_counters(cls, merged_alignments) -> "MergedColumns":
        """The reference's ``strand -> Counter{(chrom, pos): count}`` in one pass per strand."""
        fast = _pydict_reader()
        strands, chrom_codes, poss, counts = [], [], [], []
        names: dict = {}
        name_list: list = []
        for strand, table in merged_alignments.items():
            if not table:
                continue
            other = strand not in STRANDS  # (class OtherStrand: filed under strand code 0 with (strand, chrom) as the name)
            n = len(table)
            done = False
            if fast is not None and not other:  # one PyDict_Next loop in C (csrc/rp_pydict.cpp)
                code = np.empty(n, np.int32)
                pos = np.empty(n, np.int64)
                cnt = np.empty(n, np.int64)
                known = len(name_list)
                try:
                    done = fast(table, name_list, code.ctypes.data, pos.ctypes.data, cnt.ctypes.data, n) == n
                except (TypeError, ValueError, OverflowError):
                    del name_list[known:]  # e.g. float counts: let the Python loop below decide
                if done:
                    chrom_codes.append(code)
            if not done:
                fast = None  # (one numbering of the chromosomes per call: stay on this path)
                if name_list and not names:
                    names = {c: k for k, c in enumerate(name_list)}
                keys = list(table.keys())
                pos = np.fromiter(map(itemgetter(1), keys), np.int64, n)
                cnt = np.fromiter(table.values(), np.int64, n)
                local, code = _factorize(list(map(itemgetter(0), keys)))
                if other:
                    local = [OtherStrand((strand, c)) for c in local]
                remap = np.array([names.setdefault(c, len(names)) for c in local], np.int32)
                chrom_codes.append(remap[code])
                name_list = list(names)
            strands.append(np.full(n, 0 if other else STRANDS.index(strand), np.uint8))
            poss.append(pos)
            counts.append(cnt)
        if not poss:
            return cls(np.zeros(0, np.uint8), np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int64), [])
        return cls(np.concatenate(strands), np.concatenate(chrom_codes), np.concatenate(poss), np.concatenate(counts),
                   name_list)

    def as_counters(self):
        """Back to ``strand -> Counter`` (for ``export_wig`` and other consumers of the reference's format)."""
        out = defaultdict(Counter)
        for s, c, p, n in zip(self.strand.tolist(), self.chrom.tolist(), self.pos.tolist(), self.count.tolist()):
            name = self.chroms[c]
            if isinstance(name, OtherStrand):
                out[name[0]][(name[1], p)] += n
            else:
                out[STRANDS[s]][(name, p)] += n
        return out

    def group_codes(self, group_keys) -> np.ndarray:
        """Row -> group number of a candidate-ORF index (``NativeIndex.group_keys`` = (strand, chrom)
        pairs), -1 for rows on a (strand, chrom) no ORF lives on."""
        return group_lut(self.chroms, group_keys)[self.strand, self.chrom] if self.pos.size else np.zeros(0, np.int32)


class AlignmentColumns(NamedTuple):
    """What ``split_bam`` returns (bam.py:33-153), as columns: one row per (read length, strand,
    chrom, 5'-end position) key."""

    length: np.ndarray  # int32  aligned read length
    strand: np.ndarray  # uint8  0 = '+', 1 = '-' (after the protocol's strand assignment)
    chrom: np.ndarray  # int32  index into ``chroms``
    pos: np.ndarray  # int64  1-based 5'-end position (bam.py:133)
    count: np.ndarray  # int64
    chroms: list

    @classmethod
    def from_nested(cls, alignments) -> "AlignmentColumns":
        """The reference's ``length -> strand -> Counter{(chrom, pos): count}``."""
        parts = []
        names: dict = {}
        for length, by_strand in alignments.items():
            m = MergedColumns.from_counters(by_strand)
            remap = np.array([names.setdefault(c, len(names)) for c in m.chroms], np.int32)
            parts.append((np.full(m.pos.size, int(length), np.int32), m.strand, remap[m.chrom] if m.pos.size else m.chrom, m.pos, m.count))
        if not parts:
            z = np.zeros(0, np.int64)
            return cls(np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros(0, np.int32), z, z, [])
        return cls(*(np.concatenate(col) for col in zip(*parts)), list(names))

    def read_length_counts(self) -> dict:
        """``read_length_counts`` of ``split_bam``: reads per aligned length."""
        lengths, inverse = np.unique(self.length, return_inverse=True)
        sums = np.bincount(inverse, weights=self.count.astype(np.float64), minlength=lengths.size)
        return {int(length): int(s) for length, s in zip(lengths, sums)}

    def of_length(self, length: int) -> MergedColumns:
        keep = self.length == length
        return MergedColumns(self.strand[keep], self.chrom[keep], self.pos[keep], self.count[keep], self.chroms)

    def as_nested(self):
        out = defaultdict(lambda: defaultdict(Counter))
        for ln, s, c, p, n in zip(self.length.tolist(), self.strand.tolist(), self.chrom.tolist(), self.pos.tolist(), self.count.tolist()):
            name = self.chroms[c]
            if isinstance(name, OtherStrand):
                out[ln][name[0]][(name[1], p)] += n
            else:
                out[ln][STRANDS[s]][(name, p)] += n
        return out


def merge_read_lengths(alignments, psite_offsets) -> MergedColumns:
    """Drop-in for ``ribotricer.detect_orfs.merge_read_lengths`` (detect_orfs.py:54-83): keep the
    read lengths that have an offset, shift '+' rows by +offset and '-' rows by -offset.  Rows of
    different lengths that now name one position are NOT added up here -- the device does that
    when it builds the coverage (integer adds: same totals in any order).

    ``alignments``: :class:`AlignmentColumns`, or the reference's nested dicts.
    """
    cols = alignments if isinstance(alignments, AlignmentColumns) else AlignmentColumns.from_nested(alignments)
    lengths = np.fromiter(psite_offsets.keys(), np.int64, len(psite_offsets))
    offsets = np.fromiter(psite_offsets.values(), np.int64, len(psite_offsets))
    if lengths.size == 0 or cols.pos.size == 0:
        z = np.zeros(0, np.int64)
        return MergedColumns(np.zeros(0, np.uint8), np.zeros(0, np.int32), z, z, list(cols.chroms))
    order = np.argsort(lengths)
    lengths, offsets = lengths[order], offsets[order]
    at = np.searchsorted(lengths, cols.length)
    at_c = np.minimum(at, lengths.size - 1)
    keep = lengths[at_c] == cols.length
    shift = offsets[at_c[keep]]
    strand = cols.strand[keep]
    forward = strand == 0
    if any(isinstance(c, OtherStrand) for c in cols.chroms):  # only '+' moves downstream (detect_orfs.py:76-79)
        forward &= ~np.array([isinstance(c, OtherStrand) for c in cols.chroms])[cols.chrom[keep]]
    pos = cols.pos[keep] + np.where(forward, shift, -shift)
    return MergedColumns(strand, cols.chrom[keep], pos, cols.count[keep], list(cols.chroms))


def big_positions_device(values, device=None) -> np.ndarray:
    """Sorted indices of the entries of an int32 device array (a dense coverage, or CSR counts) that pass
    ``RP_MAX_COUNT`` (``rp_coverage_big_positions_dev``: one pass; the first call sizes the buffer)."""
    import torch

    from .engine import _ptr

    dev = values.device
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    lib = _lib.load()
    found = ctypes.c_int64(0)
    _lib.check(lib.rp_coverage_big_positions_dev(dev.index, _ptr(values), values.numel(), None, 0, ctypes.byref(found), stream))
    if found.value == 0:
        return np.zeros(0, np.int64)
    out = torch.empty(found.value, dtype=torch.int64, device=dev)
    _lib.check(lib.rp_coverage_big_positions_dev(dev.index, _ptr(values), values.numel(), _ptr(out), out.numel(), ctypes.byref(found), stream))
    return np.sort(out.cpu().numpy())


class UploadedColumns(NamedTuple):
    """The merged alignment columns on a device (:func:`upload_columns`): everything of the coverage build that does
    not depend on the candidate-ORF index -- so it can run beside the index parse."""

    strand: object  # uint8 device tensor
    chrom: object  # int32
    pos: object  # int64
    count: object  # int64
    chroms: list
    device: object


def upload_columns(merged, device=None) -> UploadedColumns:
    """Columns of ``merged`` (:class:`MergedColumns` or the reference's ``strand -> Counter``: converted first) on the
    device.  21 bytes per row across PCIe; no dependence on the index."""
    import torch

    from .engine import get_engine

    cols = merged if isinstance(merged, MergedColumns) else MergedColumns.from_counters(merged)
    dev = get_engine(device).device

    def to_dev(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # (blocking: the source may be a temporary, and it is pageable)

    return UploadedColumns(to_dev(cols.strand, np.uint8), to_dev(cols.chrom, np.int32), to_dev(cols.pos, np.int64),
                           to_dev(cols.count, np.int64), list(cols.chroms), dev)


def build_coverage_device(merged, index, device=None, big=None, cmap=None, out=None):
    """Dense P-site coverage of every (strand, chrom) group of ``index`` (a ``NativeIndex``) in
    HBM: ``(coverage int32 device tensor, base)`` with ``base[(strand, chrom)] = (index of position
    lo, lo)`` -- what ``gather.interval_table_from_index`` takes.  ``merged``: :class:`MergedColumns`
    or the reference's ``strand -> Counter``.  Only the histogram rows cross PCIe (16 bytes per
    row); the adding-up and the range check (RP_ERR_COUNTS) happen on the device.

    Counts beyond ``RP_MAX_COUNT`` = 2^24 - 1 (the reference has none: detect_orfs.py:176-187 work on Python
    ints; the scoring kernels' fp32 codon arithmetic is exact up to there): with ``big`` = a dict the build
    goes through and ``big["positions"]`` receives the sorted coverage indices of such positions (empty on
    ordinary samples) -- the caller finishes the ORFs that hold one with
    ``engine.rescore_big_count_orfs``; without it (callers that hand the coverage to the fp32 scorers
    themselves) such a count raises ``RibophaseError`` (status -7).  A negative count or a sum past 2^31 - 1
    raises either way.

    ``cmap`` (``gather.CoverageMap`` of this index): the COMPACT coverage -- only the blocks (positions, by default) under an exon
    interval have a slot (``cmap.compact_len`` positions instead of the dense layout's; ``cmap.table`` is the interval
    table that goes with it); rows under no exon are dropped on the device, as the reference never looks them up.

    ``out``: an int32 device tensor of at least the coverage's length to build into (zeroed here) instead of a fresh
    allocation -- a caller that scores many samples against one index keeps one buffer (``shards.IndexShards``)."""
    import torch

    from .engine import _ptr, get_engine
    from .gather import coverage_layout

    up = merged if isinstance(merged, UploadedColumns) else None  # (already on the device: upload_columns ran beside the index parse)
    cols = up if up is not None else (merged if isinstance(merged, MergedColumns) else MergedColumns.from_counters(merged))
    eng = get_engine(device if up is None else up.device)
    dev = eng.device
    extent = index.extents
    base, total = coverage_layout(extent)
    keys = index.group_keys
    if cmap is not None and cmap.dense_len != total:
        raise ValueError(f"the coverage map was built for a dense layout of {cmap.dense_len} positions, this index spans {total}")
    n_cov = total if cmap is None else cmap.compact_len
    if out is not None:
        if out.dtype != torch.int32 or out.device != dev or out.numel() < n_cov:
            raise ValueError("build_coverage_device(out=...): an int32 tensor of the coverage's length on the build's device is needed")
        coverage = out[:n_cov] if out.numel() > n_cov and n_cov > 0 else out
        coverage.zero_()
    else:
        coverage = torch.zeros(n_cov, dtype=torch.int32, device=dev)
    if big is not None:
        big["positions"] = np.zeros(0, np.int64)
    if int(cols.pos.numel() if up is not None else cols.pos.size) == 0 or total == 0:
        return coverage, base
    g_start = np.array([base[k][0] for k in keys], np.int64)
    g_lo = np.array([extent[k][0] for k in keys], np.int64)
    g_hi = np.array([extent[k][1] for k in keys], np.int64)
    # (strand, chromosome) -> group of the index; -1 where no ORF lives.  Everything per row -- the
    # lookup, the extent test, the range check of the counts that DO land (a position on a contig
    # without candidate ORFs is never a key of a lookup in the reference, whatever its count:
    # detect_orfs.py:176-187) and the adding-up -- happens on the device: the columns go up as they are.
    n_chroms = max(1, len(cols.chroms))
    lut = group_lut(cols.chroms, keys)

    def to_dev(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        return torch.from_numpy(a).to(dev)  # (blocking: `a` may be a converted temporary; pageable sources gain nothing from "async")

    if up is not None:
        d_strand, d_chrom, d_pos, d_count = up.strand, up.chrom, up.pos, up.count
    else:
        d_strand, d_chrom = to_dev(cols.strand, np.uint8), to_dev(cols.chrom, np.int32)
        d_pos, d_count = to_dev(cols.pos, np.int64), to_dev(cols.count, np.int64)
    d_lut, d_start, d_lo, d_hi = to_dev(lut.ravel(), np.int32), to_dev(g_start, np.int64), to_dev(g_lo, np.int64), to_dev(g_hi, np.int64)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    flag = ctypes.c_int32(0)
    _lib.check(
        _lib.load().rp_coverage_build_rows_dev(
            dev.index, _ptr(d_strand), _ptr(d_chrom), _ptr(d_pos), _ptr(d_count), d_pos.numel(), _ptr(d_lut), n_chroms,
            _ptr(d_start), _ptr(d_lo), _ptr(d_hi), len(keys), _ptr(coverage), coverage.numel(), stream,
            ctypes.byref(flag) if big is not None else None, cmap.ptr if cmap is not None else None, total if cmap is not None else 0,
            cmap.block_positions if cmap is not None else 0,
        )
    )
    if big is not None and flag.value:
        big["positions"] = big_positions_device(coverage)
    return coverage, base


class _BamView(ctypes.Structure):  # rp_bam_view
    _fields_ = (
        [("n_rows", ctypes.c_int64), ("n_refs", ctypes.c_int64)]
        + [(_n, ctypes.c_void_p) for _n in ("length", "strand", "chrom", "pos", "count", "ref_names", "ref_off")]
        + [("n_lengths", ctypes.c_int64), ("length_order", ctypes.c_void_p)]
        + [(_n, ctypes.c_int64) for _n in ("total", "valid", "qcfail", "duplicate", "secondary", "unmapped", "multi")]
    )


def split_bam(bam_path: str, protocol: str, prefix: str, read_lengths=None):
    """Drop-in for ``ribotricer.bam.split_bam`` (bam.py:33-153) without pysam: the BAM is read once
    by the native BGZF/BAM reader (``rp_bam_split_host``).  Returns ``(alignments, read_length_counts)``
    with ``alignments`` as :class:`AlignmentColumns` (``.as_nested()`` gives the reference's dicts)
    and writes ``{prefix}_bam_summary.txt`` in the reference's format (bam.py:139-151)."""
    from .index import _array

    if protocol not in ("forward", "reverse"):
        raise ValueError("protocol must be 'forward' or 'reverse' (bam.py:108-128)")
    lib = _lib.load()
    want = np.ascontiguousarray(sorted(set(int(x) for x in read_lengths)), dtype=np.int32) if read_lengths is not None else None
    handle = ctypes.c_void_p()
    _lib.check(
        lib.rp_bam_split_host(
            bam_path.encode(), 0 if protocol == "forward" else 1,
            ctypes.c_void_p(want.ctypes.data) if want is not None and want.size else None,
            0 if want is None else int(want.size), ctypes.byref(handle),
        )
    )
    try:
        v = _BamView()
        _lib.check(lib.rp_bam_view_host(handle, ctypes.byref(v)))
        n = int(v.n_rows)
        ref_off = _array(v.ref_off, int(v.n_refs) + 1, np.int64)
        names = ctypes.string_at(v.ref_names, int(ref_off[-1])) if int(ref_off[-1]) else b""
        chroms = [names[ref_off[k] : ref_off[k + 1]].decode("utf-8") for k in range(int(v.n_refs))]
        cols = AlignmentColumns(
            _array(v.length, n, np.int32), _array(v.strand, n, np.uint8), _array(v.chrom, n, np.int32),
            _array(v.pos, n, np.int64), _array(v.count, n, np.int64), chroms,
        )
        order = _array(v.length_order, int(v.n_lengths), np.int32).tolist()
        if read_lengths is not None and want.size == 0:  # an empty selection keeps nothing (bam.py:104)
            cols = AlignmentColumns(*(c[:0] for c in cols[:5]), chroms)
        stats = {k: int(getattr(v, k)) for k in ("total", "valid", "qcfail", "duplicate", "secondary", "unmapped", "multi")}
        if read_lengths is not None and want.size == 0:
            stats["valid"] = 0
    finally:
        lib.rp_bam_free(handle)
    by_length = cols.read_length_counts()
    counts = {length: by_length[length] for length in order if length in by_length}  # first-met order, as the reference's dict
    summary = (
        f"summary:\n\ttotal_reads: {stats['total']}\n\tunique_mapped: {stats['valid']}\n"
        f"\tqcfail: {stats['qcfail']}\n\tduplicate: {stats['duplicate']}\n\tsecondary: {stats['secondary']}\n"
        f"\tunmapped:{stats['unmapped']}\n\tmulti:{stats['multi']}\n\nlength dist:\n"
    )
    for length in sorted(counts):
        summary += f"\t{length}: {counts[length]}\n"
    with open(f"{prefix}_bam_summary.txt", "w") as output:
        output.write(summary)
    return cols, counts
