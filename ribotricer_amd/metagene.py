"""Metagene profiles and P-site offsets on the device.

``metagene_coverage`` / ``align_metagenes`` keep the reference's signatures, return values and
output files (ribotricer/metagene.py:160-328).  What changes is the inner double loop of the
reference -- every annotated ORF x every read length x up to 600 nucleotides, one dict lookup
and one pandas operation at a time ("This may take a long time", detect_orfs.py:444): per read
length the 5'-end histogram is laid out as dense coverage in HBM (``rp_coverage_build_dev``),
the leader + ORF profiles of all annotated ORFs are gathered in one launch
(``rp_gather_profiles_dev``) and ``rp_metagene_dev`` normalises and adds them up in the
reference's order with the reference's float64 operations, so the profiles carry the same
bits.  The two periodicity scores per read length come from the float-profile scorer
(``rp_phase_score_f64_csr_dev``); the lag search of ``align_metagenes`` is one
``numpy.correlate`` per read length on 600-point vectors and stays on the host.
"""

from __future__ import annotations

import ctypes
import sys
from collections import OrderedDict

import numpy as np

from . import _lib
from .const import CUTOFF, TYPICAL_OFFSET


class _CdsLayout:
    """(strand, chrom) extents of the annotated ORFs incl. their flanks -- what
    ``alignments.build_coverage_device`` needs of an index."""

    def __init__(self, records, flank: int):
        ext: dict = {}
        for r in records:
            key = (r.strand, r.chrom)
            # every position a profile can name: the flanks hang on the FIRST block's start and the LAST block's end
            # (metagene.py:66-71), the blocks themselves may reach further (an exon nested in an earlier one ends before
            # it; a block with end < start names nothing)
            lo, hi = r.intervals[0][0] - flank, r.intervals[-1][1] + flank
            for s, e in r.intervals:
                if e >= s:
                    lo, hi = min(lo, s), max(hi, e)
            if key in ext:
                ext[key] = (min(ext[key][0], lo), max(ext[key][1], hi))
            else:
                ext[key] = (lo, hi)
        self.extents = ext
        self.group_keys = sorted(ext)
        self.group_lo = np.array([ext[k][0] for k in self.group_keys], np.int64)
        self.group_hi = np.array([ext[k][1] for k in self.group_keys], np.int64)


def _profile_intervals(records, base, max_positions: int, offset_5p: int, offset_3p: int):
    """Interval table of the profiles of ``orf_coverage_length`` (metagene.py:97-157): leader,
    exons, trailer in ascending genome order, trimmed to the first ``max_positions`` nucleotides
    in TRANSCRIPT orientation ('-' strand profiles are the reversed ascending list, so they lose
    their ascending head instead of their tail)."""
    from .gather import IntervalTable

    iv_start, iv_len, orf_iv, reverse, lengths = [], [], [0], [], []
    for r in records:
        minus = r.strand == "-"
        lead, trail = (offset_3p, offset_5p) if minus else (offset_5p, offset_3p)  # metagene.py:128-129
        first, last = r.intervals[0][0], r.intervals[-1][1]
        ivs = [(first - lead, first - 1)] + list(r.intervals) + [(last + 1, last + trail)]
        ivs = [(s, e) for s, e in ivs if e >= s]
        total = sum(e - s + 1 for s, e in ivs)
        drop = max(0, total - max_positions)
        if drop and minus:  # keep the last max_positions of the ascending list
            kept = []
            for s, e in ivs:
                n = e - s + 1
                if drop >= n:
                    drop -= n
                    continue
                kept.append((s + drop, e))
                drop = 0
            ivs = kept
        elif drop:  # keep the first max_positions
            kept, room = [], max_positions
            for s, e in ivs:
                if room <= 0:
                    break
                n = min(e - s + 1, room)
                kept.append((s, s + n - 1))
                room -= n
            ivs = kept
        start0, lo = base[(r.strand, r.chrom)]
        for s, e in ivs:
            iv_start.append(start0 + (s - lo))
            iv_len.append(e - s + 1)
        orf_iv.append(len(iv_start))
        reverse.append(1 if minus else 0)
        lengths.append(sum(e - s + 1 for s, e in ivs))
    offsets = np.zeros(len(records) + 1, np.int64)
    np.cumsum(np.asarray(lengths, np.int64), out=offsets[1:])
    return IntervalTable(np.asarray(iv_start, np.int64), np.asarray(iv_len, np.int32), np.asarray(orf_iv, np.int64),
                         np.asarray(reverse, np.uint8), offsets)


def _as_records(cds):
    """Annotated ORFs as ``detect_orfs.IndexRecord`` rows: accepts those, or the reference's ORF
    objects (``.chrom .strand .intervals[k].start/.end``)."""
    from .detect_orfs import IndexRecord

    out = []
    for o in cds:
        if isinstance(o, IndexRecord):
            out.append(o)
        else:
            ivs = tuple((iv.start, iv.end) for iv in o.intervals)
            out.append(IndexRecord(getattr(o, "oid", ""), getattr(o, "category", "annotated"), getattr(o, "tid", ""), "", "", "", "",
                                   o.chrom, o.strand, None, ivs))
    return out


def annotated_records(ribotricer_index: str):
    """The annotated ORFs of an index file, the rows ``parse_ribotricer_index`` keeps
    (detect_orfs.py:86-131: leading rows whose text contains 'annotated', category == 'annotated')."""
    from .detect_orfs import parse_index_line

    out = []
    with open(ribotricer_index) as fh:
        fh.readline()
        for line in fh:
            if "annotated" not in line:
                break
            rec = parse_index_line(line)
            if rec.category == "annotated":
                out.append(rec)
    return out


def metagene_coverage(cds, alignments, read_lengths, prefix, max_positions=600, offset_5p=20, offset_3p=0,
                      meta_min_reads=100000, device=None):
    """Drop-in for ``ribotricer.metagene.metagene_coverage`` (metagene.py:160-265).

    ``cds``: annotated ORFs (``annotated_records``, or the reference's ORF objects);
    ``alignments``: ``alignments.AlignmentColumns`` or the reference's nested dicts;
    ``read_lengths``: reads per length -- lengths under ``meta_min_reads`` are DELETED from it,
    as the reference does.  Returns ``{length: (from_start, from_stop, phase_5p, valid_5p,
    phase_3p, valid_3p)}`` with pandas Series profiles and writes the two
    ``{prefix}_metagene_profiles_{5p,3p}.tsv`` files."""
    import pandas as pd

    from . import backend
    from .alignments import AlignmentColumns
    from .statistics import phasescore_batch

    for length, reads in list(read_lengths.items()):  # metagene.py:196-199
        if reads < meta_min_reads:
            del read_lengths[length]
    cols = alignments if isinstance(alignments, AlignmentColumns) else AlignmentColumns.from_nested(alignments)
    records = _as_records(cds)
    layout = _CdsLayout(records, max(offset_5p, offset_3p))
    lib = _lib.load()
    on_host = backend.selected() == "cpu"
    if on_host:  # the GPU-less backend: the same profiles through rp_gather_profiles_host / rp_metagene_host
        table = _profile_intervals(records, {k: (0, 0) for k in layout.group_keys}, max_positions, offset_5p, offset_3p)  # genomic coordinates
        group_of = {k: g for g, k in enumerate(layout.group_keys)}
        h_group = np.array([group_of[(r.strand, r.chrom)] for r in records], np.int32)
        h_start = np.ascontiguousarray(table.iv_start, np.int64)
        h_end = h_start + np.asarray(table.iv_len, np.int64) - 1
        ptr = backend._ptr
    else:
        import torch

        from .alignments import build_coverage_device
        from .engine import _ptr, get_engine
        from .gather import coverage_layout, gather_profiles_device

        eng = get_engine(device)
        dev = eng.device
        base, _ = coverage_layout(layout.extents)
        table = _profile_intervals(records, base, max_positions, offset_5p, offset_3p)
    metagenes = {}
    for length in read_lengths:
        if on_host:
            keys, vals = backend.merged_keys(cols.of_length(int(length)), layout)
            n = len(records)
            h_counts = np.empty(int(table.offsets[-1]), np.int32)
            _lib.check(lib.rp_gather_profiles_host(ptr(keys), ptr(vals), keys.size, ptr(h_start), ptr(h_end), ptr(table.orf_iv), ptr(h_group),
                                                   ptr(table.reverse), ptr(table.offsets), n, ptr(h_counts), 0))
            h_mean = np.empty(max(n, 1), np.float64)
            sums_h = np.empty(2 * max_positions, np.float64)
            seen_h = np.empty(2 * max_positions, np.int32)
            _lib.check(lib.rp_metagene_host(ptr(h_counts), ptr(table.offsets), n, max_positions, ptr(h_mean), ptr(sums_h), ptr(seen_h)))
        else:
            # (big={}: counts beyond 2^24 - 1 are fine here -- the metagene kernels add in int64 / float64)
            coverage, _ = build_coverage_device(cols.of_length(int(length)), layout, dev, big={})
            counts, offsets = gather_profiles_device(coverage, table, dev)
            n = offsets.numel() - 1
            mean = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
            sums = torch.empty(2 * max_positions, dtype=torch.float64, device=dev)
            seen = torch.empty(2 * max_positions, dtype=torch.int32, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.rp_metagene_dev(dev.index, _ptr(counts), _ptr(offsets), n, max_positions, _ptr(mean), _ptr(sums), _ptr(seen), stream))
            torch.cuda.synchronize(dev)
            sums_h, seen_h = sums.cpu().numpy(), seen.cpu().numpy()
        n_start = int((seen_h[:max_positions] > 0).sum())
        n_stop = int((seen_h[max_positions:] > 0).sum())
        start_vals = sums_h[:n_start] / seen_h[:n_start]  # metagene.py:236-238
        stop_vals = (sums_h[max_positions : max_positions + n_stop] / seen_h[max_positions : max_positions + n_stop])[::-1]
        from_start = pd.Series(start_vals, index=np.arange(-offset_5p, n_start - offset_5p))
        from_stop = pd.Series(stop_vals, index=np.arange(offset_3p - n_stop + 1, offset_3p + 1))
        phase, valid, _ = phasescore_batch([start_vals, stop_vals], device=None if on_host else dev)
        metagenes[length] = (from_start, from_stop, np.float64(phase[0]), int(valid[0]), np.float64(phase[1]), int(valid[1]))
    to_write_5p = "fragment_length\toffset_5p\tprofile\tphase_score\tvalid_codons\n"
    to_write_3p = "fragment_length\toffset_3p\tprofile\tphase_score\tvalid_codons\n"
    for length in sorted(metagenes):
        m = metagenes[length]
        to_write_5p += f"{length}\t{offset_5p}\t{m[0].tolist()}\t{m[2]}\t{m[3]}\n"
        to_write_3p += f"{length}\t{offset_3p}\t{m[1].tolist()}\t{m[4]}\t{m[5]}\n"
    with open(f"{prefix}_metagene_profiles_5p.tsv", "w") as output:
        output.write(to_write_5p)
    with open(f"{prefix}_metagene_profiles_3p.tsv", "w") as output:
        output.write(to_write_3p)
    return metagenes


def best_lag(reference: np.ndarray, profile: np.ndarray, bound: int) -> int:
    """The lag the reference's offset step reads off ``np.correlate(reference, profile, "full")`` (metagene.py:317-321):
    of the ``2 * bound`` lags around the middle of the full cross-correlation, the one with the largest overlap product,
    counted from the middle of that window.  Only those lags are computed here -- one dot product over the two aligned
    slices each -- instead of all ``len(reference) + len(profile) - 1`` of them."""
    n, m = int(reference.size), int(profile.size)
    if n == 0 or m == 0:
        raise ValueError("best_lag: a metagene profile is empty")  # (np.correlate refuses empty inputs the same way)
    middle = (n + m - 1) // 2
    window = range(n + m - 1)[middle - bound : middle + bound]  # (a range takes the slice the way the array would, ends included)
    products = np.empty(len(window))
    for slot, k in enumerate(window):
        shift = k - (m - 1)  # profile[i] lies over reference[i + shift]
        lo, hi = max(0, -shift), min(m, n - shift)
        products[slot] = np.dot(reference[lo + shift : hi + shift], profile[lo:hi])
    return int(np.argmax(products)) - len(window) // 2


def align_metagenes(metagenes, read_lengths, prefix, phase_score_cutoff=CUTOFF, remove_nonperiodic=False):
    """Drop-in for ``ribotricer.metagene.align_metagenes`` (metagene.py:268-328).  The contract: with
    ``remove_nonperiodic`` the read lengths whose 5' metagene scores below the cutoff leave BOTH dicts (in place); no
    length left -> ``sys.exit`` with the reference's warning; the base is the most abundant length (the first one on
    equal counts); every length gets ``best_lag`` against the base's 5' profile plus ``TYPICAL_OFFSET``, in the order
    of ``metagenes``; ``{prefix}_psite_offsets.txt`` lists the lags under a "relative lag to base" line."""
    if remove_nonperiodic:
        for length in [k for k, entry in metagenes.items() if entry[2] < phase_score_cutoff]:  # (a NaN score stays, as `<` has it)
            del read_lengths[length], metagenes[length]
    if not read_lengths:
        sys.exit(f"WARNING: no periodic read length found... using cutoff {phase_score_cutoff}")
    base = max(read_lengths, key=read_lengths.get)  # (max keeps the first of equals: dict order, as the reference's scan)
    if not read_lengths[base] > 0:
        base = 0  # (no length with reads: the reference's scan never leaves its initial 0, and fails on the lookup below)
    anchor = np.asarray(metagenes[base][0].values, dtype=np.float64)
    lags = {length: best_lag(anchor, np.asarray(entry[0].values, dtype=np.float64), min(base, length)) for length, entry in metagenes.items()}
    with open(f"{prefix}_psite_offsets.txt", "w") as output:
        output.write("".join([f"relative lag to base: {base}\n"] + [f"\tlag of {length}: {lag}\n" for length, lag in lags.items()]))
    return OrderedDict((length, lag + TYPICAL_OFFSET) for length, lag in lags.items())
