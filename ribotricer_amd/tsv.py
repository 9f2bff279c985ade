"""Native TSV row rendering: ``rp_format_rows_host``.

Replaces the per-ORF ``formatter.format(...)`` of the reference's output loop
(``ribotricer/detect_orfs.py:301-324``).  The index columns travel as two byte tables
(``ORF_ID\\tORF_type`` and ``transcript_id ... start_codon``); everything numeric and the
``profile`` list are rendered in C++, byte-identical to CPython's ``'{}'.format``.
"""

from __future__ import annotations

import ctypes
from typing import Iterator, Sequence

import numpy as np

from . import _lib

RP_ERR_SIZE = -2


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
) -> Iterator[bytes]:
    """Yield the TSV body of ORFs [first, last) in chunks of at most ``chunk_bytes``
    (a single longer row gets a buffer of its own)."""
    lib = _lib.load()
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
    head_buf = ctypes.create_string_buffer(head, len(head)) if head else ctypes.create_string_buffer(1)
    tail_buf = ctypes.create_string_buffer(tail, len(tail)) if tail else ctypes.create_string_buffer(1)
    cap = chunk_bytes
    out = ctypes.create_string_buffer(cap)
    nxt = ctypes.c_int64(first)
    ln = ctypes.c_size_t(0)
    cur = first
    while cur < last:
        rc = lib.rp_format_rows_host(
            counts.ctypes.data, offsets.ctypes.data, last, phase.ctypes.data, valid.ctypes.data,
            read_count.ctypes.data, status.ctypes.data, ctypes.cast(head_buf, ctypes.c_void_p), head_off.ctypes.data,
            ctypes.cast(tail_buf, ctypes.c_void_p), tail_off.ctypes.data, int(bool(report_all)), cur,
            ctypes.cast(out, ctypes.c_void_p), cap, ctypes.byref(nxt), ctypes.byref(ln),
        )
        if rc == RP_ERR_SIZE and ln.value > cap:
            cur = nxt.value      # (skipped rows before it are done with)
            cap = int(ln.value)  # one row longer than the chunk: give it room and retry
            out = ctypes.create_string_buffer(cap)
            continue
        _lib.check(rc)
        if ln.value:
            yield out.raw[: ln.value]
        cur = nxt.value
