"""Native TSV row rendering: ``rp_format_rows_host``.

Replaces the per-ORF ``formatter.format(...)`` of the reference's output loop
(``ribotricer/detect_orfs.py:301-324``).  The index columns travel as two byte tables
(``ORF_ID\\tORF_type`` and ``transcript_id ... start_codon``); everything numeric and the
``profile`` list are rendered in C++, byte-identical to CPython's ``'{}'.format``.
"""

from __future__ import annotations

import ctypes
import os
import threading
from typing import Iterator, Sequence

import numpy as np

from . import _lib

RP_ERR_SIZE = -2
_tls = threading.local()


def double_repr(value: float) -> str:
    """``repr(float)`` as the library renders it (tests compare it with CPython's)."""
    buf = ctypes.create_string_buffer(32)
    n = _lib.load().rp_format_double_repr(float(value), ctypes.cast(buf, ctypes.c_void_p))
    return buf.raw[:n].decode("ascii")


def int_list_str(values: np.ndarray) -> str:
    """``str(list_of_int)`` as the library renders it."""
    v = np.ascontiguousarray(values, dtype=np.int32)
    buf = ctypes.create_string_buffer(2 + 13 * max(1, v.size))
    n = _lib.load().rp_format_int_list(v.ctypes.data, v.size, ctypes.cast(buf, ctypes.c_void_p))
    return buf.raw[:n].decode("ascii")


def string_table(rows: Sequence[str]):
    """Concatenate per-ORF strings into (bytes, int64 offsets[n + 1])."""
    enc = [r.encode("utf-8") for r in rows]
    off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    return b"".join(enc), off


def record_tables(records):
    """(head, head_off, tail, tail_off) for IndexRecord-like rows: the columns of
    detect_orfs.py:304-323 that come straight from the index."""
    head, head_off = string_table([f"{r.oid}\t{r.category}" for r in records])
    tail, tail_off = string_table(
        [f"{r.tid}\t{r.ttype}\t{r.gid}\t{r.gname}\t{r.gtype}\t{r.chrom}\t{r.strand}\t{r.start_codon}" for r in records]
    )
    return head, head_off, tail, tail_off


_SPARE: dict = {}  # size -> chunk buffers handed back by a consumer that has written them (format_rows_native(recycle=True))
_SPARE_LOCK = threading.Lock()
_SPARE_MAX = 96  # buffers kept, all sizes together
_REGULAR: set = set()  # the chunk sizes callers ask _take_buffer for: only those are worth keeping


def _take_buffer(cap: int) -> np.ndarray:
    """A chunk buffer of exactly ``cap`` bytes: a spare one of that size if the pool holds one (its pages are mapped
    already: a fresh 4 MB buffer costs a thousand page faults).  The pool is keyed by size, so callers with different
    ``chunk_bytes`` do not evict each other's spares; an over-long row's one-off buffer (allocated directly, a size nobody
    will ask for again) is never pooled."""
    with _SPARE_LOCK:
        _REGULAR.add(cap)
        spare = _SPARE.get(cap)
        if spare:
            return spare.pop()
    return np.empty(cap, np.uint8)


def _give_back(chunk) -> None:
    buf = getattr(chunk, "obj", None)
    if isinstance(buf, np.ndarray):
        with _SPARE_LOCK:
            if int(buf.size) in _REGULAR and sum(len(v) for v in _SPARE.values()) < _SPARE_MAX:
                _SPARE.setdefault(int(buf.size), []).append(buf)


def _format_range(arrays, tables_c, report_all, first, last, chunk_bytes) -> list:
    """Rows of ORFs [first, last) as a list of byte chunks (one C call per chunk)."""
    lib = _lib.load()
    counts, offsets, phase, valid, read_count, status = arrays
    head_buf, head_off, tail_buf, tail_off = tables_c
    cap = chunk_bytes
    # Every chunk gets a buffer of its own (numpy: no zero fill) that the C call fills and the writer takes as it is:
    # no copy of the text is made while the GIL is held.  (Until round 4's last day a reusable buffer was copied out
    # with `out.raw[:n]` -- two copies of every chunk under the GIL, which the one writer thread also needs: 12.8 GB of
    # report_all text took 2.43 s that way, 1.9 s with one copy, and less with none.)
    out = _take_buffer(cap)
    nxt = ctypes.c_int64(first)
    ln = ctypes.c_size_t(0)
    cur = first
    chunks = []
    while cur < last:
        rc = lib.rp_format_rows_host(
            counts.ctypes.data, offsets.ctypes.data, last, phase.ctypes.data, valid.ctypes.data,
            read_count.ctypes.data, status.ctypes.data, ctypes.cast(head_buf, ctypes.c_void_p), head_off.ctypes.data,
            ctypes.cast(tail_buf, ctypes.c_void_p), tail_off.ctypes.data, int(bool(report_all)), cur,
            out.ctypes.data, cap, ctypes.byref(nxt), ctypes.byref(ln),
        )
        if rc == RP_ERR_SIZE and ln.value > cap:
            cur = nxt.value      # (skipped rows before it are done with)
            if out.size == chunk_bytes:
                _give_back(memoryview(out))  # (untouched: back to the pool)
            cap = int(ln.value)  # one row longer than the chunk: a buffer of its own for THIS row, and retry
            out = np.empty(cap, np.uint8)
            continue
        _lib.check(rc)
        if ln.value:
            chunks.append(memoryview(out)[: ln.value])  # (a view: it keeps its buffer alive until it is written)
            if nxt.value < last:
                cap = chunk_bytes  # (after an over-long row: back to the normal chunk size and the pooled buffers)
                out = _take_buffer(cap)
        cur = nxt.value
    return chunks


def format_rows_native(
    counts: np.ndarray,
    offsets: np.ndarray,
    phase: np.ndarray,
    valid: np.ndarray,
    read_count: np.ndarray,
    status: np.ndarray,
    tables,
    report_all: bool,
    first: int = 0,
    last: int | None = None,
    chunk_bytes: int = 64 << 20,
    threads: int | None = None,
    recycle: bool = False,
) -> Iterator[bytes]:
    """Yield the TSV body of ORFs [first, last), in order, in chunks of at most
    ``chunk_bytes`` (a single longer row gets a buffer of its own).

    ``threads`` > 1 renders disjoint ORF ranges concurrently (the C call holds no shared
    state and ctypes drops the GIL); default: the usable cores (at most 32), one for small batches.
    A chunk is a view of a buffer of its own.  ``recycle``: the caller is done with a chunk when it asks for the next
    one (it wrote it out): the buffer goes back to a pool and later chunks reuse it -- not for callers that keep them."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    phase = np.ascontiguousarray(phase, dtype=np.float64)
    valid = np.ascontiguousarray(valid, dtype=np.int32)
    read_count = np.ascontiguousarray(read_count, dtype=np.int64)
    status = np.ascontiguousarray(status, dtype=np.uint8)
    head, head_off, tail, tail_off = tables
    n = offsets.size - 1
    last = n if last is None else last
    if not (phase.size == valid.size == read_count.size == status.size == n == head_off.size - 1 == tail_off.size - 1):
        raise ValueError("per-ORF arrays and string tables must all have n_orfs entries")
    def c_buffer(table):  # bytes are copied once; a ctypes view (NativeIndex.tables_native) is used as it is
        if isinstance(table, (bytes, bytearray)):
            return ctypes.create_string_buffer(bytes(table), len(table)) if table else ctypes.create_string_buffer(1)
        return table

    head_buf, tail_buf = c_buffer(head), c_buffer(tail)
    arrays = (counts, offsets, phase, valid, read_count, status)
    tables_c = (head_buf, np.ascontiguousarray(head_off, dtype=np.int64), tail_buf, np.ascontiguousarray(tail_off, dtype=np.int64))
    total_nt = int(offsets[last] - offsets[first]) if last > first else 0
    if threads is None:
        threads = min(32, _lib.usable_cores()) if total_nt > (4 << 20) else 1
    def hand_out(chunks):
        for chunk in chunks:
            yield chunk
            if recycle:
                _give_back(chunk)

    if threads <= 1 or last - first < 2:
        yield from hand_out(_format_range(arrays, tables_c, report_all, first, last, chunk_bytes))
        return
    # ranges of about chunk_bytes / 4 nucleotides (a row is ~3.2 bytes of text per nt), cut on
    # the prefix sum of lengths; a sliding window of futures keeps the output ordered and bounded
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    chunk_bytes = max(1 << 20, chunk_bytes // threads)  # per-thread buffer
    per_range = max(1, min(chunk_bytes // 4, -(-total_nt // threads)))
    cuts = np.searchsorted(offsets[first : last + 1], offsets[first] + np.arange(per_range, total_nt, per_range)) + first
    bounds = np.unique(np.concatenate(([first], cuts, [last])))
    window: deque = deque()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        for a, b in zip(bounds[:-1], bounds[1:]):
            window.append(pool.submit(_format_range, arrays, tables_c, report_all, int(a), int(b), chunk_bytes))
            if len(window) >= 2 * threads:
                yield from hand_out(window.popleft().result())
        while window:
            yield from hand_out(window.popleft().result())


class _OrderedOffsets:
    """File offsets for ranges that are rendered concurrently but must land in range order: range r learns where it
    starts once the ranges before it have declared their sizes (not finished writing)."""

    def __init__(self, base: int):
        self.cond = threading.Condition()
        self.next_range = 0
        self.at = base

    def claim(self, r: int, nbytes: int) -> int:
        with self.cond:
            while self.next_range != r:
                self.cond.wait()
            start = self.at
            self.at += nbytes
            self.next_range += 1
            self.cond.notify_all()
            return start


def write_rows_native(
    fd: int,
    file_offset: int,
    counts: np.ndarray,
    offsets: np.ndarray,
    phase: np.ndarray,
    valid: np.ndarray,
    read_count: np.ndarray,
    status: np.ndarray,
    tables,
    report_all: bool,
    threads: int | None = None,
    range_bytes: int = 16 << 20,
) -> int:
    """Render the TSV body of all ORFs of the arrays into the file behind ``fd`` (open for reading AND writing: it is
    mapped) from ``file_offset`` on; returns the
    number of bytes written (the file ends there afterwards).  The same bytes as ``b"".join(format_rows_native(...))``
    -- but every render thread copies its own ranges to their final offsets (ranges are sized as they are rendered,
    ``_OrderedOffsets``) through a shared MAPPING of the file, so neither the rendering nor the copy into the page cache
    is serial and no intermediate ``bytes`` object is made: a human-sized `report_all` export writes 12.8 GB
    (detect_orfs.py:301-324 writes them row by row).  (``pwrite`` from many threads was measured first: buffered writes
    to one file serialise on the inode lock -- 5.2 s against 2.5 s for a single writer, profiles/archive/r04_export_e2e_11M_pwrite.json.)
    The file is extended to an upper bound of the text first (sparse) and cut to the exact end afterwards."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    phase = np.ascontiguousarray(phase, dtype=np.float64)
    valid = np.ascontiguousarray(valid, dtype=np.int32)
    read_count = np.ascontiguousarray(read_count, dtype=np.int64)
    status = np.ascontiguousarray(status, dtype=np.uint8)
    head, head_off, tail, tail_off = tables
    n = offsets.size - 1
    if not (phase.size == valid.size == read_count.size == status.size == n == head_off.size - 1 == tail_off.size - 1):
        raise ValueError("per-ORF arrays and string tables must all have n_orfs entries")
    if n <= 0:
        return 0

    def c_buffer(table):
        if isinstance(table, (bytes, bytearray)):
            return ctypes.create_string_buffer(bytes(table), len(table)) if table else ctypes.create_string_buffer(1)
        return table

    arrays = (counts, offsets, phase, valid, read_count, status)
    tables_c = (c_buffer(head), np.ascontiguousarray(head_off, dtype=np.int64), c_buffer(tail), np.ascontiguousarray(tail_off, dtype=np.int64))
    total_nt = int(offsets[n] - offsets[0])
    # rows cost ~3.2 bytes of text per nucleotide + ~120 bytes of columns: ranges of about range_bytes of text
    weight = (offsets[1:] - offsets[0]) * 4 + np.arange(1, n + 1, dtype=np.int64) * 128
    per_range = max(1, range_bytes)
    cuts = np.searchsorted(weight, np.arange(per_range, int(weight[-1]), per_range), side="left") + 1
    bounds = np.unique(np.concatenate(([0], np.minimum(cuts, n), [n])))
    ranges = list(zip(bounds[:-1].tolist(), bounds[1:].tolist()))
    if threads is None:
        threads = min(32, _lib.usable_cores()) if total_nt > (4 << 20) or n > 100000 else 1
    threads = max(1, min(threads, len(ranges)))
    order = _OrderedOffsets(file_offset)
    chunk_bytes = 2 * range_bytes + (1 << 20)  # (a range nearly always fits one buffer: one claim per range)
    # upper bound of the text: <= 12 bytes per printed count ("-2147483648, "), the index columns as they are, < 200
    # bytes of numbers, tabs and brackets per row
    import mmap

    upper = 12 * total_nt + int(head_off[n] - head_off[0]) + int(tail_off[n] - tail_off[0]) + 200 * n + 4096
    gran = mmap.ALLOCATIONGRANULARITY
    map_base = file_offset // gran * gran
    os.ftruncate(fd, max(os.fstat(fd).st_size, file_offset + upper))
    mm = mmap.mmap(fd, file_offset - map_base + upper, offset=map_base)
    anchor = ctypes.c_char.from_buffer(mm)
    mm_addr = ctypes.addressof(anchor) - map_base  # address of file offset 0 in this mapping's coordinates

    lib = _lib.load()
    head_buf, head_off_c, tail_buf, tail_off_c = tables_c

    def pwrite_all(view, at: int) -> None:  # (a memcpy into the mapping; ctypes releases the GIL for its duration)
        if len(view):
            ctypes.memmove(mm_addr + at, (ctypes.c_char * len(view)).from_buffer(view) if not isinstance(view, bytes) else view, len(view))

    def work(r: int) -> int:
        claimed = [False]
        try:
            return render(r, claimed)
        except BaseException:
            if not claimed[0]:  # the ranges behind this one must not wait for a claim that will never come
                order.claim(r, 0)
            raise

    def render(r: int, claimed: list) -> int:
        a, b = ranges[r]
        cap = chunk_bytes
        out = getattr(_tls, "wout", None)  # one reusable render buffer per thread (first touch is the costly part)
        if out is None or len(out) < cap:
            out = _tls.wout = ctypes.create_string_buffer(cap)
        cap = len(out)
        nxt = ctypes.c_int64(a)
        ln = ctypes.c_size_t(0)
        cur = a
        pieces: list = []  # (rare: a range that did not fit one buffer is collected and written as one)
        while cur < b:
            rc = lib.rp_format_rows_host(
                counts.ctypes.data, offsets.ctypes.data, b, phase.ctypes.data, valid.ctypes.data, read_count.ctypes.data,
                status.ctypes.data, ctypes.cast(head_buf, ctypes.c_void_p), head_off_c.ctypes.data,
                ctypes.cast(tail_buf, ctypes.c_void_p), tail_off_c.ctypes.data, int(bool(report_all)), cur,
                ctypes.cast(out, ctypes.c_void_p), cap, ctypes.byref(nxt), ctypes.byref(ln),
            )
            if rc == RP_ERR_SIZE and ln.value > cap:  # one row longer than the buffer: give it room and retry
                cur = nxt.value
                cap = int(ln.value)
                out = _tls.wout = ctypes.create_string_buffer(cap)
                continue
            _lib.check(rc)
            cur = nxt.value
            if cur >= b and not pieces:  # the common case: the whole range in one buffer -> straight from it to the file
                start = order.claim(r, ln.value)
                claimed[0] = True
                if ln.value:
                    ctypes.memmove(mm_addr + start, out, ln.value)
                return int(ln.value)
            pieces.append(ctypes.string_at(out, ln.value))
        blob = b"".join(pieces)
        start = order.claim(r, len(blob))
        claimed[0] = True
        pwrite_all(blob, start)
        return len(blob)

    try:
        if threads == 1:
            written = sum(work(r) for r in range(len(ranges)))
        else:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=threads) as pool:  # (ranges are taken in order: a claim never waits for an unstarted range)
                written = sum(pool.map(work, range(len(ranges))))
    finally:
        del anchor  # (the mapping cannot be closed while a ctypes view of it lives)
        mm.close()
    os.ftruncate(fd, file_offset + written)
    return written

