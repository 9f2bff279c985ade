"""Host-side mirror of the hot loop of ``ribotricer.detect_orfs``.

``export_orf_coverages`` keeps the reference's signature, TSV header, column order
and number formatting (ribotricer/detect_orfs.py:206-324) so that ``detect_orfs()``
(detect_orfs.py:510-520), ``learn_cutoff`` (learn_cutoff.py:231-263) and the
downstream parsers (count_orfs.py:63-77) work unchanged.  What changes is the loop
body: profiles are packed CSR-style once, scored for all ORFs in one launch of the
gfx950 kernels (phase score, valid codons, read count, codon minimum and the status
predicate), and rows are formatted from the returned arrays.

Index parsing and profile gathering (orf.py:122-182, detect_orfs.py:134-203) are
build-owned minimal host code here -- the "next" rows of SURVEY.md section 8(f).
"""

from __future__ import annotations

import sys
from typing import NamedTuple

import numpy as np

from .const import (
    CUTOFF,
    MINIMUM_DENSITY_OVER_ORF,
    MINIMUM_READS_PER_CODON,
    MINIMUM_VALID_CODONS,
    MINIMUM_VALID_CODONS_RATIO,
)
from ._lib import make_filter


def get_engine(device=None):
    """``engine.get_engine``, imported on first use: the engine needs torch, the cpu backend of this module does not."""
    from .engine import get_engine as _get_engine

    return _get_engine(device)

# detect_orfs.py:241-260 (the last name carries the newline, :259)
COLUMNS = [
    "ORF_ID",
    "ORF_type",
    "status",
    "phase_score",
    "read_count",
    "length",
    "valid_codons",
    "valid_codons_ratio",
    "read_density",
    "transcript_id",
    "transcript_type",
    "gene_id",
    "gene_name",
    "gene_type",
    "chrom",
    "strand",
    "start_codon",
    "profile",
]


class IndexRecord(NamedTuple):
    """One candidate ORF of a ribotricer index (the fields ORF.from_string keeps, orf.py:153-182)."""

    oid: str
    category: str
    tid: str
    ttype: str
    gid: str
    gname: str
    gtype: str
    chrom: str
    strand: str
    start_codon: object  # str, or None when the index field has < 3 characters (orf.py:117-118)
    intervals: tuple  # ((start, end), ...) 1-based closed, ascending


def parse_index_line(line: str) -> IndexRecord:
    """Parse one line of ``{prefix}_candidate_orfs.tsv`` (format: prepare_orfs.py:370-404).

    Error behaviour follows ORF.from_string (orf.py:143-152): a wrong column count
    terminates with the reference's message.
    """
    fields = line.split("\t")
    if len(fields) != 11:
        sys.exit(
            "{}\n{}".format(
                "Error: unexpected number of columns found for index file",
                "please run ribotricer prepare-orfs to regenerate",
            )
        )
    blocks = []
    for group in fields[10].split(","):
        start, end = group.split("-")
        blocks.append((int(start), int(end)))
    blocks.sort(key=lambda b: b[0])  # orf.py:100
    length = sum(e - s + 1 for s, e in blocks)
    tid = fields[2]
    oid = f"{tid}_{blocks[0][0]}_{blocks[-1][1]}_{length}"  # orf.py:103 (index column 0 is ignored)
    seq = fields[9]
    return IndexRecord(
        oid, fields[1], tid, fields[3], fields[4], fields[5], fields[6], fields[7], fields[8],
        seq[:3] if len(seq) >= 3 else None, tuple(blocks),
    )


def read_index(ribotricer_index: str) -> list:
    """All records of an index file, header skipped (detect_orfs.py:273-276)."""
    records = []
    with open(ribotricer_index) as anno:
        anno.readline()
        for line in anno:
            records.append(parse_index_line(line))
    return records


def orf_coverage(orf: IndexRecord, alignments, offset_5p: int = 0, offset_3p: int = 0) -> list:
    """Per-nucleotide P-site counts of one ORF, 5'->3' (detect_orfs.py:134-203).

    ``alignments[strand][(chrom, pos)]`` -> count; missing positions count 0; the
    profile is reversed for '-' strand ORFs.
    """
    strand = orf.strand
    if strand == "-":
        offset_5p, offset_3p = offset_3p, offset_5p
    table = alignments[strand] if strand in alignments else {}
    chrom = orf.chrom
    get = table.get
    first_start = orf.intervals[0][0]
    last_end = orf.intervals[-1][1]
    coverage = [get((chrom, pos), 0) for pos in range(first_start - offset_5p, first_start)]
    for start, end in orf.intervals:
        coverage.extend(get((chrom, pos), 0) for pos in range(start, end + 1))
    coverage.extend(get((chrom, pos), 0) for pos in range(last_end + 1, last_end + offset_3p + 1))
    if strand == "-":
        coverage.reverse()
    return coverage


def pack_profiles(records, merged_alignments):
    """CSR-pack the profiles of all records: ``(counts int32[sum L], offsets int64[n+1])``."""
    lengths = np.fromiter((sum(max(0, e - s + 1) for s, e in r.intervals) for r in records), np.int64, len(records))  # (end < start: no positions)
    offsets = np.zeros(len(records) + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    counts = np.zeros(int(offsets[-1]), np.int32)
    for r, o in zip(records, offsets[:-1]):
        cov = orf_coverage(r, merged_alignments)
        counts[o : o + len(cov)] = cov
    return counts, offsets


def gather_profiles(records, merged_alignments, device=None):
    """Profiles of all records, gathered on the device (ribotricer_amd/gather.py):
    ``(counts, offsets)`` as device tensors.  Same bytes as :func:`pack_profiles`."""
    from .gather import build_dense_coverage, build_interval_table, gather_profiles_device

    coverage, base = build_dense_coverage(merged_alignments, records)
    table = build_interval_table(records, base)
    return gather_profiles_device(coverage, table, device)


def gather_profiles_indexed(index, merged_alignments, device=None):
    """Same for a natively parsed index (``ribotricer_amd.index.NativeIndex``): the interval
    table comes from the parser's arrays, the dense coverage is accumulated on the device from
    the alignment columns (``alignments.build_coverage_device``) -- no per-ORF, per-position or
    per-key Python beyond the one conversion pass when ``merged_alignments`` still is the
    reference's ``strand -> Counter``."""
    from .alignments import build_coverage_device
    from .gather import gather_profiles_device, interval_table_from_index, make_gather_plan

    # (big={}: a gather has no fp32 limit -- a count beyond 2^24 - 1 is copied like any other; only callers that hand the
    # coverage to the fp32 scorers themselves need the strict contract)
    coverage, base = build_coverage_device(merged_alignments, index, device, big={})
    table = interval_table_from_index(index, base)
    return gather_profiles_device(coverage, table, device, plan=make_gather_plan(table, coverage.numel(), device))


def _devices_from_env():
    """``RIBOTRICER_AMD_DEVICES="0,1,2,3"``: the GPUs ``export_orf_coverages`` shards the index over."""
    import os

    spec = os.environ.get("RIBOTRICER_AMD_DEVICES", "").strip()
    return [int(x) for x in spec.split(",") if x.strip() != ""] if spec else None


def score_profiles(counts, offsets, phase_score_cutoff, min_valid_codons, min_reads_per_codon,
                   min_valid_codons_ratio, min_density_over_orf, device=None, devices=None, shards=None, pinned=None) -> dict:
    """One launch for the whole batch; host numpy arrays back (24 B + 2 B per ORF).  With
    ``devices`` (several GPUs of this node): nt-balanced ORF-index slices, one per GPU, host
    concat (``engine.score_sharded``).  Exact frame ties carry the reference's bits throughout
    (``engine.resolve_big_ties`` finishes the few the device cannot).  ``shards``: the caller's
    ``engine.CsrShards`` of this index and these devices (per-device plans kept across samples)."""
    thresholds = make_filter(
        phase_score_cutoff, min_valid_codons, min_reads_per_codon, min_valid_codons_ratio, min_density_over_orf
    )
    if devices is not None and len(devices) > 1:
        from .engine import score_sharded

        return score_sharded(counts, offsets, devices, thresholds=thresholds, shards=shards)
    eng = get_engine(device if devices is None or not len(devices) else devices[0])
    return eng.score_host(counts, offsets, thresholds=thresholds, pinned=pinned)


def format_rows(records, counts, offsets, res, report_all: bool):
    """Yield the TSV rows of detect_orfs.py:301-324 (same ``str.format('{}')`` renderings).

    Pure-Python rendering, kept as the readable statement of the row format and as the
    cross-check of the native renderer (``format_rows_native``) that the export path uses."""
    lengths = np.diff(offsets)
    n_codons = np.maximum(1, lengths // 3)  # detect_orfs.py:281
    formatter = "{}\t" * (len(COLUMNS) - 1) + "{}\n"
    for i, r in enumerate(records):
        translating = bool(res["status"][i])
        if not report_all and not translating:
            continue
        valid = int(res["valid"][i])
        count = int(res["read_count"][i])
        nc = int(n_codons[i])
        yield formatter.format(
            r.oid,
            r.category,
            "translating" if translating else "nontranslating",
            repr(float(res["phase"][i])),  # str(np.float64) == shortest round-trip repr
            count,
            int(lengths[i]),
            valid,
            valid / nc,  # detect_orfs.py:285
            repr(count / nc),  # detect_orfs.py:287 (np.float64 str == float repr)
            r.tid,
            r.ttype,
            r.gid,
            r.gname,
            r.gtype,
            r.chrom,
            r.strand,
            r.start_codon,
            counts[offsets[i] : offsets[i + 1]].tolist(),
        )


def format_rows_native(records, counts, offsets, res, report_all: bool, chunk_bytes: int = 64 << 20):
    """Yield the same bytes as ``format_rows`` in large chunks, rendered by the library
    (``rp_format_rows_host``: SURVEY.md 8(f) f2; detect_orfs.py:301-324)."""
    from . import tsv

    yield from tsv.format_rows_native(
        counts, offsets, res["phase"], res["valid"], res["read_count"], res["status"],
        tsv.record_tables(records), report_all, chunk_bytes=chunk_bytes,
    )


def export_orf_coverages(
    ribotricer_index: str,
    merged_alignments,
    prefix: str,
    phase_score_cutoff: float = CUTOFF,
    min_valid_codons: int = MINIMUM_VALID_CODONS,
    min_reads_per_codon: float = MINIMUM_READS_PER_CODON,
    min_valid_codons_ratio: float = MINIMUM_VALID_CODONS_RATIO,
    min_density_over_orf: float = MINIMUM_DENSITY_OVER_ORF,
    report_all: bool = False,
    devices=None,
    timings: dict | None = None,
) -> None:
    """Write ``{prefix}_translating_ORFs.tsv`` -- drop-in for detect_orfs.py:206-324.

    ``timings`` (beyond the reference's signature): a dict that receives the wall time of every
    stage in seconds (index parse, coverage build, interval table + plans, device scoring, profiles
    back to the host, TSV rendering + write) -- what scripts/bench_export_big.py reports.

    ``devices`` (beyond the reference's signature; default: ``RIBOTRICER_AMD_DEVICES`` or the
    current GPU): the GPUs to shard the candidate ORFs over -- contiguous nt-balanced slices of
    the index, one per GPU, results concatenated on the host; no collective (BASELINE configs[3]).

    ``merged_alignments``: what ``merge_read_lengths`` returns -- the reference's
    ``strand -> Counter{(chrom, pos): count}`` or this package's columnar
    ``alignments.MergedColumns`` (``alignments.merge_read_lengths``).

    index text -> ``rp_index_parse_host`` (f3) -> interval table + gather plan -> (f1, fused:
    :func:`score_index`) ``rp_phase_score_coverage_dev`` -> ``rp_format_rows_host`` (f2): no
    per-ORF Python.  ``RIBOTRICER_AMD_BACKEND=cpu`` (or ``auto`` without a visible HIP device): the same
    pipeline on the host through the library's ``*_host`` entry points (``backend.py``) -- the reference's
    bits in every column."""
    from . import tsv
    from .index import NativeIndex

    import os
    import time

    from . import backend

    if backend.selected() == "cpu":
        # the GPU-less backend (backend.py): parser -> host gather -> rp_phase_score_csr_host -> the same row formatter.
        # Chosen from what is visible BEFORE any work; a failing device call below is never retried here.
        t0 = time.perf_counter()
        index = _index_of(ribotricer_index)
        if timings is not None:
            timings["index_parse"] = time.perf_counter() - t0
        counts, offsets, res = backend.score_index_host(
            index, merged_alignments, make_filter(phase_score_cutoff, min_valid_codons, min_reads_per_codon, min_valid_codons_ratio,
                                                  min_density_over_orf), timings=timings)
        t0 = time.perf_counter()
        with open(f"{prefix}_translating_ORFs.tsv", "wb") as output:
            output.write(("\t".join(COLUMNS) + "\n").encode("utf-8"))
            for chunk in tsv.format_rows_native(counts, offsets, res["phase"], res["valid"], res["read_count"], res["status"],
                                                index.tables_native, report_all, recycle=True):
                output.write(chunk)
        if timings is not None:
            timings["tsv_render_write"] = time.perf_counter() - t0
            timings["backend"] = "cpu"
        return
    if devices is None:
        devices = _devices_from_env()
    if timings is not None:
        timings["backend"] = "hip"
    t0 = time.perf_counter()
    # the alignment columns go up to the device (and a Counter input becomes columns first) BESIDE the index parse:
    # neither needs the other (the parser is C++ without the GIL, the copies release it too)
    from concurrent.futures import ThreadPoolExecutor

    from .alignments import UploadedColumns, upload_columns

    sharded = devices is not None and len(devices) > 1
    with ThreadPoolExecutor(max_workers=1) as side:
        uploaded = None
        if sharded:  # every distinct device takes the columns up itself (shards.columns_on_devices), all beside the parse
            from .shards import ColumnsOnDevices, columns_on_devices

            if not isinstance(merged_alignments, (UploadedColumns, ColumnsOnDevices)):
                uploaded = side.submit(columns_on_devices, merged_alignments, devices)
        elif not isinstance(merged_alignments, UploadedColumns):
            uploaded = side.submit(upload_columns, merged_alignments, None if not devices else f"cuda:{int(devices[0])}")
        index = _index_of(ribotricer_index)  # parsed once per file: detect-orfs scores one index against many samples
        if timings is not None:
            timings["index_parse"] = time.perf_counter() - t0
        if uploaded is not None:
            merged_alignments = uploaded.result()
    if timings is not None:
        timings["index_parse_and_column_upload"] = time.perf_counter() - t0
    counts, offsets, res = score_index(
        index, merged_alignments, phase_score_cutoff, min_valid_codons, min_reads_per_codon,
        min_valid_codons_ratio, min_density_over_orf, report_all, devices, timings=timings, profiles_on_device=True,
        reuse_result_buffers=True,  # (the results are written out before this function returns)
        # default mode prints translating rows only (detect_orfs.py:301-302): too-close-to-call ORFs that cannot be
        # translating under ANY resolution are not re-walked (RP_FILTER_PRINTED_ONLY; same file, byte for byte)
        printed_only=not report_all and os.environ.get("RIBOTRICER_AMD_PRINTED_ONLY", "1") != "0",
    )
    if timings is not None and "flags" in res:
        from ._lib import FLAG_UNRESOLVED

        timings["unresolved_orfs"] = int(np.count_nonzero(np.asarray(res["flags"]) & FLAG_UNRESOLVED))
    t0 = time.perf_counter()
    tables = index.tables_native
    with open(f"{prefix}_translating_ORFs.tsv", "w+b") as output:  # (read-write: the writer maps the file)
        output.write(("\t".join(COLUMNS) + "\n").encode("utf-8"))
        output.flush()
        at = output.tell()
        # the profiles come back from the device in slices (pinned staging buffers, the copy of slice
        # k+1 behind the rendering of slice k): a human-sized `report_all` run prints 14 GB of them.  Every
        # render thread writes its ranges at their final file offsets (tsv.write_rows_native): neither the
        # rendering nor the copy into the page cache is serial.
        import os

        mapped = os.environ.get("RIBOTRICER_AMD_TSV_WRITER", TSV_WRITER) == "mapped"
        # (several devices: one part per slice of the index, each on the device that gathered it -- shards.IndexShards)
        parts = counts if isinstance(counts, ProfileParts) else ProfileParts([(counts, offsets, 0)])
        for part_counts, part_offsets, first in parts:
            for a, b, part in _profile_slices(part_counts, part_offsets):
                lo, hi = first + a, first + b
                sliced = (tables[0], tables[1][lo : hi + 1], tables[2], tables[3][lo : hi + 1])
                args = (part, part_offsets[a : b + 1] - part_offsets[a], res["phase"][lo:hi], res["valid"][lo:hi], res["read_count"][lo:hi],
                        res["status"][lo:hi], sliced, report_all)
                if mapped:
                    at += tsv.write_rows_native(output.fileno(), at, *args)
                else:  # render threads hand their chunks, in order, to this one writer
                    for chunk in tsv.format_rows_native(*args, recycle=True):  # (written before the next one is asked for)
                        output.write(chunk)
    if timings is not None:
        timings["profiles_d2h_tsv_render_write"] = time.perf_counter() - t0


class ProfileParts(list):
    """The profiles the TSV prints as ``(counts, offsets of the part's ORFs, first ORF)`` triples in index order -- one per
    slice of a sharded export, each ``counts`` on the device that gathered it (``score_index`` with several devices)."""


TSV_WRITER = "stream"  # "mapped": tsv.write_rows_native (parallel copies into a mapping of the file); RIBOTRICER_AMD_TSV_WRITER overrides


_INDEX_CACHE: dict = {}  # (real path, size, mtime_ns) -> NativeIndex; the interval table and gather plan hang on the index
_INDEX_CACHE_MAX = 2


def _index_of(path: str):
    """The parsed index of ``path``, kept across calls while the file is unchanged (RIBOTRICER_AMD_INDEX_CACHE=0
    disables it).  The reference re-reads the index for every sample (detect_orfs.py:263-278); a batch job that runs
    many samples against one index pays the parse (0.45 s for 11 M lines), the interval table and the gather plan once."""
    import os

    from .index import NativeIndex

    if os.environ.get("RIBOTRICER_AMD_INDEX_CACHE", "1") == "0":
        return NativeIndex.from_file(path)
    st = os.stat(path)
    key = (os.path.realpath(path), st.st_size, st.st_mtime_ns)
    hit = _INDEX_CACHE.pop(key, None)
    if hit is None:
        hit = NativeIndex.from_file(path)
        while len(_INDEX_CACHE) >= _INDEX_CACHE_MAX:
            _forget_index(_INDEX_CACHE.pop(next(iter(_INDEX_CACHE))))
    _INDEX_CACHE[key] = hit  # most recently used last
    return hit


def _forget_index(index) -> None:
    """An index leaves the cache: what hangs on it and holds device memory registered elsewhere -- the per-device shards,
    whose streams' workspaces and outputs live in the engines -- is released now, not when the garbage collector gets
    round to it.  (Tables, maps and plans of the one-GPU path are plain attributes: they go with the object.)"""
    for shards in index.__dict__.pop("_shard_cache", {}).values():
        if shards is not None:
            shards.release()


def forget_indexes() -> None:
    """Empty the index cache (tests; a long-lived process that is done with its indexes)."""
    while _INDEX_CACHE:
        _forget_index(_INDEX_CACHE.pop(next(iter(_INDEX_CACHE))))


def _layout_key(base, coverage_len: int, device):
    return (str(get_engine(device).device), int(coverage_len), tuple(sorted((k, tuple(int(x) for x in v)) for k, v in base.items())))


def _layout_cached(index, base, coverage_len: int, device) -> bool:
    return _layout_key(base, coverage_len, device) in index.__dict__.get("_layout_cache", {})


def _compact_coverage() -> bool:
    import os

    return os.environ.get("RIBOTRICER_AMD_COMPACT_COVERAGE", "1") != "0"


def _table_and_plan(index, base, coverage_len: int, device, table=None):
    """(interval table, gather plan, per-layout extras) of an index for a coverage layout, remembered on the index
    object: all depend on the index and the layout only (the layout on the index's group extents only), not on the
    sample.  ``coverage_len`` is the length of the DENSE layout.  With the compact coverage (default;
    ``RIBOTRICER_AMD_COMPACT_COVERAGE=0`` keeps the dense one) ``extras["coverage_map"]`` is the ``gather.CoverageMap``
    of the layout, the table is in compact coordinates and the plan is built for the compact length.  ``extras`` is a
    dict that lives and dies with the layout: the multi-GPU shards (``engine.CsrShards`` / ``engine.CoverageShards``:
    per-device windows, gather plans, tile plans) are kept there too."""
    from .gather import CoverageMap, interval_table_from_index, make_gather_plan

    key = _layout_key(base, coverage_len, device)
    cache = index.__dict__.setdefault("_layout_cache", {})
    if key not in cache:
        if table is None:
            table = interval_table_from_index(index, base)
        for old in cache.values():
            for kept in old[2].values():  # (the extras also hold plain records: sample count, placement report)
                if hasattr(kept, "release"):
                    kept.release()
        cache.clear()  # (one layout per index: another one replaces it)
        extras: dict = {}
        plan_len = coverage_len
        if _compact_coverage() and len(table.iv_len):
            from ._lib import ERR_INTERVALS, RibophaseError

            try:
                cmap = CoverageMap(table, coverage_len, device)
            except RibophaseError as e:  # (an interval off the layout: not mappable, not plannable either -- the dense path copes)
                if e.status != ERR_INTERVALS:
                    raise
            else:
                extras["coverage_map"] = cmap
                table, plan_len = cmap.table, cmap.compact_len
        cmap = extras.get("coverage_map")
        plan = make_gather_plan(table, plan_len, device, device_intervals=None if cmap is None else cmap.device_intervals)
        if cmap is not None:
            cmap.release_device_intervals()  # (0.3 GB of device memory the plan has digested)
        cache[key] = (table, plan, extras)
    return cache[key]


def _shards(extras: dict, kind: str, devices, build):
    """The layout's shards of ``kind`` for ``devices`` (made once per index, layout and device list)."""
    key = (kind, tuple(int(d) if isinstance(d, int) else str(d) for d in devices))
    if key not in extras:
        extras[key] = build()
    return extras[key]


def _profile_slices(counts, offsets, slice_nt: int = 64 << 20):
    """Yield ``(a, b, counts[offsets[a]:offsets[b]] as a host array)`` over consecutive ORF ranges of
    about ``slice_nt`` nucleotides.  ``counts``: a host array (sliced in place) or a device tensor
    (copied slice by slice through two pinned buffers, the next copy in flight while the caller works
    on the current one)."""
    import torch

    n = offsets.size - 1
    total = int(offsets[-1]) if n >= 0 else 0
    if n <= 0:
        return
    cuts = np.searchsorted(offsets, np.arange(slice_nt, total, slice_nt), side="right") - 1
    bounds = np.unique(np.concatenate(([0], cuts, [n]))).astype(np.int64)
    ranges = [(int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    if not isinstance(counts, torch.Tensor):
        for a, b in ranges:
            yield a, b, counts[int(offsets[a]) : int(offsets[b])]
        return
    widest = max(int(offsets[b] - offsets[a]) for a, b in ranges)
    side = torch.cuda.Stream(device=counts.device)
    side.wait_stream(torch.cuda.current_stream(counts.device))
    bufs = [torch.empty(max(1, widest), dtype=torch.int32, pin_memory=True) for _ in range(2)]
    done = [torch.cuda.Event(), torch.cuda.Event()]

    def fetch(k):
        a, b = ranges[k]
        lo, hi = int(offsets[a]), int(offsets[b])
        with torch.cuda.stream(side):
            bufs[k % 2][: hi - lo].copy_(counts[lo:hi], non_blocking=True)
            done[k % 2].record(side)

    fetch(0)
    for k, (a, b) in enumerate(ranges):
        done[k % 2].synchronize()
        if k + 1 < len(ranges):
            fetch(k + 1)
        yield a, b, bufs[k % 2][: int(offsets[b] - offsets[a])].numpy()


def score_index(index, merged_alignments, phase_score_cutoff, min_valid_codons, min_reads_per_codon,
                min_valid_codons_ratio, min_density_over_orf, report_all: bool, devices=None, timings=None,
                profiles_on_device: bool = False, reuse_result_buffers: bool = False, printed_only: bool = False):
    """Gather + score for a natively parsed index: ``(counts, offsets, results)`` as host arrays,
    ready for the row formatter (``profiles_on_device``: ``counts`` stays a device tensor, for
    ``_profile_slices``).

    ``report_all`` (every profile is printed, detect_orfs.py:301-324): the whole CSR counts array
    is gathered (tile kernel through the index's gather plan), scored, copied back.  Default mode
    (only the translating ORFs are printed): gather and score are FUSED -- the scorer stages its
    tiles straight from the dense coverage (``rp_phase_score_coverage_dev``), the profiles are
    never written to HBM -- and only the translating ORFs' profiles are gathered afterwards;
    ``offsets`` then gives every other ORF an empty range.  With several ``devices`` both modes
    shard the ORFs (``engine.score_sharded`` / ``engine.score_coverage_sharded``).

    ``printed_only`` (default mode only; ``export_orf_coverages`` sets it): the caller prints nothing but translating
    ORFs, so too-close-to-call ORFs that no resolution could make translating keep their fp32 result
    (``RP_FILTER_PRINTED_ONLY`` / ``FLAG_UNRESOLVED``, ``engine.make_filter``).  The translating ORFs' results, hence the
    file, are the same bytes; ``results`` of the others are NOT at full resolution then.  Off: every ORF is resolved."""
    import os

    import numpy as np
    import torch

    printed_only = bool(printed_only) and not report_all

    from .alignments import build_coverage_device
    from .engine import rescore_big_count_orfs, resolve_big_ties
    from .gather import coverage_profiles_of, gather_profiles_device, interval_table_from_index, make_gather_plan, orfs_touching, select_orfs

    import time

    def lap(name, since):
        if timings is not None:
            torch.cuda.synchronize()
            timings[name] = timings.get(name, 0.0) + time.perf_counter() - since
        return time.perf_counter()

    device = None if not devices else f"cuda:{int(devices[0])}"
    t = time.perf_counter()
    big: dict = {}  # positions whose count passes 2^24 - 1 (none on ordinary samples): their ORFs are finished in float64 below
    from .gather import coverage_layout

    base0, total0 = coverage_layout(index.extents)
    if devices is not None and len(devices) > 1 and os.environ.get("RIBOTRICER_AMD_SHARD_ON_FIRST_DEVICE", "0") != "1":
        # several devices: every device builds ITS slice's compact coverage from the sample's columns and does for the slice
        # what the code below does for the whole index (shards.IndexShards); nothing is built on devices[0] and shipped
        from ._lib import ERR_INTERVALS, RibophaseError
        from .shards import IndexShards

        cache = index.__dict__.setdefault("_shard_cache", {})
        key = (tuple(str(d) for d in devices), int(total0))
        try:
            if key in cache and cache[key] is None:  # (this index could not be sharded over these devices before: do not try again)
                raise RibophaseError(ERR_INTERVALS, "not plannable (remembered)")
            if key not in cache:
                for old in cache.values():
                    if old is not None:
                        old.release()
                cache.clear()
                cache[key] = IndexShards(index, interval_table_from_index(index, base0), total0, devices)
            t = lap("interval_table_slices", t)
            thresholds = make_filter(phase_score_cutoff, min_valid_codons, min_reads_per_codon, min_valid_codons_ratio, min_density_over_orf,
                                     printed_only=printed_only)
            res, parts = cache[key].score(merged_alignments, thresholds, report_all, timings=timings, reuse_result_buffers=reuse_result_buffers)
            lap("sharded_build_score_gather", t)
            if profiles_on_device:
                return ProfileParts(parts), None, res
            # (host arrays wanted: the parts are copied back and joined; they cover the index's ORFs in order)
            offsets = np.zeros(index.n_orfs + 1, np.int64)
            chunks, base = [], 0
            for c, off_k, first in parts:
                offsets[first : first + off_k.size] = off_k + base
                base += int(off_k[-1])
                chunks.append(c.cpu().numpy())
            counts = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
            return counts, offsets, res
        except RibophaseError as e:  # (a table that cannot be mapped / planned: the per-ORF kernels below cope)
            if e.status != ERR_INTERVALS:
                raise
            half_built = cache.get(key)
            if half_built is not None:
                half_built.release()  # (slices that were built hold streams, workspaces, coverage buffers)
            cache[key] = None  # later samples of this index go straight to the path below
    from .shards import ColumnsOnDevices

    if isinstance(merged_alignments, ColumnsOnDevices):  # (the whole-coverage-on-one-device flow below: that device's copy)
        merged_alignments = merged_alignments[get_engine(device).device]
    # everything that depends on the index alone first (kept on the cached index: table, coverage map, plans), then
    # this sample's coverage -- compact: only the blocks under an exon have a slot (gather.CoverageMap)
    table, plan, extras = _table_and_plan(index, base0, total0, device)
    t = lap("interval_table_gather_plan", t)
    coverage, base = build_coverage_device(merged_alignments, index, device, big=big, cmap=extras.get("coverage_map"))
    t = lap("coverage_build", t)
    if timings is not None:
        timings["coverage_positions"] = int(coverage.numel())
        timings["dense_layout_positions"] = int(total0)
    sharded = devices is not None and len(devices) > 1
    extras["samples"] = extras.get("samples", 0) + 1
    # (No placement search of the record workspace here.  engine.tune_workspace exists for callers whose job IS the device
    # step -- many samples against one CSR batch, bench.py -- where the best placement buys 3-10 % of a 2.5 ms kernel.  A
    # sample of an 11 M-line index takes 0.2 s in this function, 5 ms of it on the device: the search (0.07-0.5 s) would
    # need hundreds of samples of one index to pay for a thousandth of each.  Rounds 4-5 kept it as an opt-in knob
    # (RIBOTRICER_AMD_PLACE_WORKSPACE, profiles/archive/r05_placement_check.txt); round 6 removed the knob: DESIGN.md section 4.)
    if plan is None or report_all:
        d_counts, d_offsets = gather_profiles_device(coverage, table, device, plan=plan)
        shards = None
        if sharded:  # per-device offsets, tile plans, workspaces: once per index, not per sample
            from .engine import CsrShards

            shards = _shards(extras, "csr", devices, lambda: CsrShards(table.offsets, devices))
        res = score_profiles(
            d_counts, d_offsets, phase_score_cutoff, min_valid_codons, min_reads_per_codon,
            min_valid_codons_ratio, min_density_over_orf, device=device, devices=devices, shards=shards,
            pinned=extras.setdefault("pinned_results", {}) if reuse_result_buffers and not sharded else None,
        )
        # (counts beyond 2^24 - 1: score_profiles scans the gathered CSR counts itself -- engine.fix_big_counts_csr)
        t = lap("gather_score_results_d2h", t)
        if profiles_on_device:  # (the caller streams them back slice by slice: _profile_slices)
            return d_counts, d_offsets.cpu().numpy(), res
        out = d_counts.cpu().numpy(), d_offsets.cpu().numpy(), res
        lap("profiles_d2h", t)
        return out
    thresholds = make_filter(
        phase_score_cutoff, min_valid_codons, min_reads_per_codon, min_valid_codons_ratio, min_density_over_orf, printed_only=printed_only
    )
    if sharded:
        from .engine import CoverageShards

        # per-device windows, gather plans and tile plans: once per index, not per sample
        res = _shards(extras, "coverage", devices, lambda: CoverageShards(table, devices, coverage.numel())).score(coverage, thresholds)
    else:
        eng = get_engine(device)
        res = eng.score_coverage(coverage, plan, thresholds=thresholds)
        # (reuse_result_buffers: pinned staging tensors kept with the index's layout, overwritten by its next sample)
        res = res.cpu_numpy(extras.setdefault("pinned_results", {}) if reuse_result_buffers else None)
        torch.cuda.synchronize(eng.device)
        resolve_big_ties(res, coverage_profiles_of(coverage, table, device), thresholds)
    if big["positions"].size:
        rescore_big_count_orfs(res, orfs_touching(table, big["positions"]), coverage_profiles_of(coverage, table, device), thresholds, device)
    t = lap("fused_score_results_d2h", t)
    keep = res["status"] != 0
    if plan is not None and not sharded:  # through the plan's pieces: ids and prefix sums on the device (GatherPlan.gather_selected)
        if "orf_lengths" not in extras:
            extras["orf_lengths"] = np.diff(table.offsets)  # (11 M subtractions: once per index, not per sample)
        d_counts, offsets = plan.gather_selected(coverage, keep, lengths=extras["orf_lengths"], reuse_arrays=reuse_result_buffers)
    else:
        chosen = np.flatnonzero(keep)
        d_counts, _ = gather_profiles_device(coverage, select_orfs(table, chosen), device)
        offsets = np.zeros(index.n_orfs + 1, np.int64)
        np.cumsum(np.where(keep, np.diff(table.offsets), 0), out=offsets[1:])
    if profiles_on_device:
        lap("translating_profiles_gather", t)
        return d_counts, offsets, res
    out = d_counts.cpu().numpy(), offsets, res
    lap("translating_profiles_gather_d2h", t)
    return out


def export_wig(merged_alignments, prefix: str) -> None:
    """Drop-in for ``ribotricer.detect_orfs.export_wig`` (detect_orfs.py:327-352): the merged P-site
    counts as ``{prefix}_pos.wig`` / ``{prefix}_neg.wig``, variableStep, sorted by (chrom, pos).
    ``merged_alignments``: the reference's ``strand -> Counter`` or ``alignments.MergedColumns``
    (rows naming one position are added up first, vectorised)."""
    from .alignments import STRANDS, MergedColumns, OtherStrand

    import ctypes

    from . import _lib

    lib = _lib.load()
    is_mapping = not isinstance(merged_alignments, MergedColumns)
    if is_mapping and any(k not in STRANDS for k in merged_alignments):
        cols = None  # (a foreign strand key, even over an empty table, takes the general road below)
    else:
        cols = merged_alignments if not is_mapping else MergedColumns.from_counters(merged_alignments)
    if cols is None or any(isinstance(c, OtherStrand) for c in cols.chroms):
        # a strand key other than '+' / '-' (never out of split_bam): every strand that is not '+' goes to _neg.wig, one
        # after the other in the mapping's order, the last one staying -- an empty table leaves an empty file
        # (detect_orfs.py:338-352).  No fast path for that.
        by_strand = merged_alignments if is_mapping else cols.as_counters()
        from itertools import groupby
        from operator import itemgetter

        for strand, table in by_strand.items():
            text = []
            for chrom, keys in groupby(sorted(table), key=itemgetter(0)):  # one block per chromosome, positions ascending
                if chrom != "":  # (a chromosome named "" sorts first and gets no header: the reference's loop starts from "")
                    text.append(f"variableStep chrom={chrom}\n")
                text.extend(f"{pos}\t{table[(chrom, pos)]}\n" for _, pos in keys)
            with open(f"{prefix}_pos.wig" if strand == "+" else f"{prefix}_neg.wig", "w") as output:
                output.write("".join(text))
        return
    names = np.asarray(cols.chroms, dtype=object)
    rank = np.argsort(np.argsort(names)) if names.size else np.zeros(0, np.int64)  # sorted() orders by chromosome NAME
    inv = np.empty(names.size, np.int64)
    inv[rank] = np.arange(names.size)
    strand8 = np.ascontiguousarray(cols.strand, dtype=np.uint8)
    chrom32 = np.ascontiguousarray(cols.chrom, dtype=np.int32)
    pos64 = np.ascontiguousarray(cols.pos, dtype=np.int64)
    count64 = np.ascontiguousarray(cols.count, dtype=np.int64)
    rank32 = np.ascontiguousarray(rank, dtype=np.int32)
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    for code, strand in enumerate(STRANDS):
        n_strand = int(np.count_nonzero(strand8 == code))
        if n_strand == 0 and (isinstance(merged_alignments, MergedColumns) or strand not in merged_alignments):
            continue  # the reference writes a file only for strands that are keys of the mapping
        # (chromosome rank, position, count) packed into one UNSIGNED 64-bit word per row (rp_wig_pack_host, in threads:
        # strand filter included), a plain value sort, and the text straight from the sorted words (rp_wig_render_host, which
        # adds up the rows of one position) -- 1e8 rows: seconds of numpy passes (masks, gathers, shifts, reduceat) became tenths
        packed = np.empty(n_strand, np.uint64)
        n_packed = ctypes.c_int64(0)
        rc = lib.rp_wig_pack_host(ptr(strand8), ptr(chrom32), ptr(pos64), ptr(count64), strand8.size, code, ptr(rank32), int(names.size),
                                  ptr(packed), ctypes.byref(n_packed)) if n_strand else 0
        path = f"{prefix}_pos.wig" if strand == "+" else f"{prefix}_neg.wig"
        if rc == 0:
            packed.sort()
            # ranges of ~1 Mi rows, cut where the position changes, rendered side by side (rp_wig_render_host adds up the
            # rows of a position and puts a chromosome's header in front of its first one) and written in order
            step = 1 << 20
            cuts = [0]
            while cuts[-1] < n_strand:
                i = min(n_strand, cuts[-1] + step)
                while 0 < i < n_strand and (int(packed[i]) >> 22) == (int(packed[i - 1]) >> 22):
                    i += 1
                cuts.append(i)
            by_rank = [str(names[int(inv[r])]).encode("utf-8") for r in range(names.size)]
            name_off = np.zeros(names.size + 1, np.int64)
            np.cumsum([len(x) for x in by_rank], out=name_off[1:])
            name_buf = ctypes.create_string_buffer(b"".join(by_rank) + b"\0")
            room = 42 * (step + 64) + sum(24 + len(x) for x in by_rank)

            def render(k: int):
                lo, hi = cuts[k], cuts[k + 1]
                need = room if hi - lo <= step + 64 else 42 * (hi - lo) + room
                buf = np.empty(need, np.uint8)  # (filled by the C call, written as it is: no copy under the GIL)
                n = lib.rp_wig_render_host(ptr(packed), lo, hi, name_buf, ptr(name_off), ptr(buf))
                return memoryview(buf)[:n]

            from concurrent.futures import ThreadPoolExecutor

            with open(path, "wb") as output, ThreadPoolExecutor(max_workers=min(8, _lib.usable_cores())) as pool:
                for part in pool.map(render, range(len(cuts) - 1)):
                    output.write(part)
            continue
        if rc != -8:  # (RP_ERR_ARG: a row does not fit the packed key -- positions >= 2^32, counts >= 2^22, >= 1024 names)
            _lib.check(rc)
        keep = strand8 == code
        chrom, pos, count = chrom32[keep], pos64[keep], count64[keep]
        order = np.lexsort((pos, rank[chrom])) if pos.size else np.zeros(0, np.int64)
        chrom, pos, count = chrom[order], pos[order], count[order]
        new_key = np.ones(pos.size, bool)
        new_key[1:] = (chrom[1:] != chrom[:-1]) | (pos[1:] != pos[:-1])
        starts = np.nonzero(new_key)[0]
        totals = np.add.reduceat(count, starts) if starts.size else count[:0]
        key_chrom = chrom[starts]
        key_pos = np.ascontiguousarray(pos[starts], dtype=np.int64)
        totals = np.ascontiguousarray(totals, dtype=np.int64)
        # one "variableStep" header per chromosome, its "{pos}\t{count}\n" lines rendered natively (rp_format_wig_rows_host)
        block = np.flatnonzero(np.concatenate(([True], key_chrom[1:] != key_chrom[:-1]))) if key_chrom.size else np.zeros(0, np.int64)
        ends = np.concatenate((block[1:], [key_chrom.size])) if block.size else block
        with open(path, "wb") as output:
            for a, b in zip(block.tolist(), ends.tolist()):
                if cols.chroms[int(key_chrom[a])] != "":  # (no header for a chromosome named "": detect_orfs.py:340-345 starts from "")
                    output.write(f"variableStep chrom={cols.chroms[int(key_chrom[a])]}\n".encode("utf-8"))
                for lo in range(a, b, 1 << 22):  # 4 Mi positions per call: <= 176 MB of text buffer
                    hi = min(b, lo + (1 << 22))
                    buf = np.empty(42 * (hi - lo), np.uint8)
                    n = lib.rp_format_wig_rows_host(key_pos[lo:hi].ctypes.data, totals[lo:hi].ctypes.data, hi - lo, buf.ctypes.data)
                    output.write(memoryview(buf)[:n])


def detect_orfs(bam, ribotricer_index, prefix, protocol, read_lengths, psite_offsets, phase_score_cutoff=CUTOFF,
                min_valid_codons=MINIMUM_VALID_CODONS, min_reads_per_codon=MINIMUM_READS_PER_CODON,
                min_valid_codons_ratio=MINIMUM_VALID_CODONS_RATIO, min_density_over_orf=MINIMUM_DENSITY_OVER_ORF,
                report_all=False, meta_min_reads=100000) -> None:
    """The ``detect-orfs`` pipeline of the reference (detect_orfs.py:355-526) on this package's
    native pieces: BAM -> 5'-end columns (``alignments.split_bam``, no pysam) -> metagene profiles
    and P-site offsets on the device (``metagene``) -> merged P-site columns -> WIG ->
    ``export_orf_coverages``.  Same argument order, same output files -- except the two PDF plots
    (plotting is out of scope) and protocol inference: ``protocol`` ('forward' / 'reverse') must be
    given, as ``--stranded`` does on the reference's CLI."""
    import datetime
    import os

    from . import alignments as al
    from . import metagene as mg

    def banner(text: str, dots: str = "...") -> None:  # the stage lines of detect_orfs.py:401-525
        print("{} {} {}".format(datetime.datetime.now().strftime("%b %d %H:%M:%S"), dots, text))

    if protocol not in ("forward", "reverse"):
        sys.exit("Error: ribotricer_amd.detect_orfs needs protocol 'forward' or 'reverse' (protocol inference is not part of this engine)")
    banner("started ribotricer detect-orfs", ".....")
    banner("started parsing ribotricer index file")
    annotated = mg.annotated_records(ribotricer_index)
    parent = os.path.dirname(prefix)
    if parent:
        os.makedirs(parent, exist_ok=True)
    banner("started reading bam file")
    cols, read_length_counts = al.split_bam(bam, protocol, prefix, read_lengths)
    banner("started calculating metagene profiles. This may take a long time...")
    metagenes = mg.metagene_coverage(annotated, cols, read_length_counts, prefix, meta_min_reads=meta_min_reads)
    if psite_offsets is None:
        banner("started inferring P-site offsets")
        psite_offsets = mg.align_metagenes(metagenes, read_length_counts, prefix, phase_score_cutoff, read_lengths is None)
    banner("started shifting according to P-site offsets")
    merged = al.merge_read_lengths(cols, psite_offsets)
    banner("started exporting wig file of alignments after shifting")
    export_wig(merged, prefix)
    banner("started calculating phase scores for each ORF")
    export_orf_coverages(ribotricer_index, merged, prefix, phase_score_cutoff, min_valid_codons, min_reads_per_codon,
                         min_valid_codons_ratio, min_density_over_orf, report_all)
    banner("finished ribotricer detect-orfs")
