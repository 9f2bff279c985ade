"""Native ribotricer index parser: ``rp_index_parse_host``.

One C++ pass over ``{prefix}_candidate_orfs.tsv`` replaces ``parse_ribotricer_index``
(``ribotricer/detect_orfs.py:86-131``) and the per-line ``ORF.from_string`` calls of the
export loop (``detect_orfs.py:273-278``, ``orf.py:122-182``).  The result carries the
interval table for the device gather, the (strand, chrom) extents for the dense coverage
layout and the two string tables the native TSV writer prints.
"""

from __future__ import annotations

import ctypes
import sys

import numpy as np

from . import _lib

RP_ERR_INDEX_COLUMNS = -9
RP_ERR_INDEX_COORD = -10


class _View(ctypes.Structure):  # rp_index_view
    _fields_ = [(n, ctypes.c_int64) for n in ("n_orfs", "n_intervals", "n_groups")] + [
        (n, ctypes.c_void_p)
        for n in ("orf_iv", "length", "group", "reverse", "iv_start", "iv_end", "group_names", "group_off",
                  "group_lo", "group_hi", "head", "head_off", "tail", "tail_off")
    ]


def _array(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class _Parsed:
    """Owns one ``rp_index`` object of the library.  Every view handed out below refers to it (through the
    ctypes buffer at the bottom of numpy's ``base`` chain), so the C++ memory lives as long as any array
    derived from it does -- not only as long as the ``NativeIndex`` (no cycle: this object knows nobody)."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _lib.load().rp_index_free(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass


def _view(ptr, n, dtype, owner):
    """numpy view (no copy) of n items of the library's memory at ptr; keeps ``owner`` alive"""
    if n == 0:
        return np.zeros(0, dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    buf._owner = owner
    a = np.frombuffer(buf, dtype=dtype, count=n)
    a.flags.writeable = False
    return a


def _bytes(ptr, n):
    return ctypes.string_at(ptr, n) if n else b""


class NativeIndex:
    """Arrays of a parsed index.  The arrays are VIEWS of the C++ object's memory (an 11 M-line index
    is 2 GB of them: no second copy), which lives as long as this object does."""

    def __init__(self, text, skip_header: bool = True):
        """``text``: the index file as bytes, or anything with the buffer protocol (``from_file`` maps
        the file: the parser threads then read it straight out of the page cache)."""
        lib = _lib.load()
        handle = ctypes.c_void_p()
        bad_line = ctypes.c_int64(0)
        flat = None
        try:
            if isinstance(text, (bytes, bytearray)):
                src, n_text = bytes(text), len(text)
            else:
                flat = np.frombuffer(text, dtype=np.uint8)
                src, n_text = ctypes.c_void_p(flat.ctypes.data if flat.size else 0), int(flat.size)
            rc = lib.rp_index_parse_host(src, n_text, int(skip_header), ctypes.byref(handle), ctypes.byref(bad_line))
        finally:
            # the export of the caller's buffer ends HERE: an error raised below must not keep it pinned through the
            # traceback's frame (a mapped file could then not be closed, and that error would mask this one)
            del flat, text
        if rc == RP_ERR_INDEX_COLUMNS:  # what ORF.from_string does (orf.py:143-152)
            sys.exit(
                "{}\n{}".format(
                    "Error: unexpected number of columns found for index file",
                    "please run ribotricer prepare-orfs to regenerate",
                )
            )
        if rc == RP_ERR_INDEX_COORD:  # a ValueError in the reference (orf.py:165-168): the same here, with the library's message
            raise _lib.IndexCoordinateError(rc, lib.rp_last_error().decode("utf-8", "replace"))
        _lib.check(rc)
        self._parsed = own = _Parsed(handle)
        v = _View()
        _lib.check(lib.rp_index_view_host(handle, ctypes.byref(v)))
        n, m, g = v.n_orfs, v.n_intervals, v.n_groups
        self.n_orfs = int(n)
        self.orf_iv = _view(v.orf_iv, n + 1, np.int64, own)
        self.length = _view(v.length, n, np.int64, own)
        self.group = _view(v.group, n, np.int32, own)
        self.reverse = _view(v.reverse, n, np.uint8, own)
        self.iv_start = _view(v.iv_start, m, np.int64, own)
        self.iv_end = _view(v.iv_end, m, np.int64, own)
        group_off = _array(v.group_off, g + 1, np.int64)
        names = _bytes(v.group_names, int(group_off[-1]))
        self.group_keys = [
            tuple(names[group_off[k] : group_off[k + 1]].decode("utf-8").split("\t", 1)) for k in range(g)
        ]  # (strand, chrom)
        self.group_lo = _array(v.group_lo, g, np.int64)
        self.group_hi = _array(v.group_hi, g, np.int64)
        self.head_off = _view(v.head_off, n + 1, np.int64, own)
        self.tail_off = _view(v.tail_off, n + 1, np.int64, own)
        self._head_ptr, self._tail_ptr = v.head, v.tail
        self._head = self._tail = None

    @property
    def head(self) -> bytes:
        """``ORF_ID\\tORF_type`` of every ORF, concatenated (a copy: tests and small inputs)."""
        if self._head is None:
            self._head = _bytes(self._head_ptr, int(self.head_off[-1]) if self.n_orfs else 0)
        return self._head

    @property
    def tail(self) -> bytes:
        if self._tail is None:
            self._tail = _bytes(self._tail_ptr, int(self.tail_off[-1]) if self.n_orfs else 0)
        return self._tail

    @classmethod
    def from_file(cls, path: str) -> "NativeIndex":
        import mmap
        import os

        with open(path, "rb") as fh:
            if os.fstat(fh.fileno()).st_size == 0:
                return cls(b"", skip_header=True)
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            try:  # (not `with mm`: closing a mapping that an in-flight exception still exports raises BufferError)
                # The reference reads the file in text mode: a lone '\r' ends a line too (universal newlines).  The native
                # parser cuts lines at '\n' (a '\r' in front of it is blank space to its number parser: CRLF files are
                # fine); a file whose HEADER line already holds a lone '\r' -- classic Mac line ends -- is translated first,
                # as Python would.  (A lone '\r' inside a data line of an otherwise '\n'-ended file is not looked for.)
                first_nl = mm.find(b"\n")
                header = mm[: first_nl if first_nl >= 0 else len(mm)]
                if b"\r" in (header[:-1] if header.endswith(b"\r") else header):
                    return cls(mm[:].replace(b"\r\n", b"\n").replace(b"\r", b"\n"), skip_header=True)
                return cls(mm, skip_header=True)  # header line skipped: detect_orfs.py:273
            finally:
                try:
                    mm.close()
                except BufferError:  # pragma: no cover - the parse error on its way out is the one to report
                    pass

    @property
    def tables(self):
        """(head, head_off, tail, tail_off) as ``tsv.format_rows_native`` takes them (byte copies)."""
        return self.head, self.head_off, self.tail, self.tail_off

    @property
    def tables_native(self):
        """The same without copying the two string tables (700 MB for an 11 M-line index): ctypes
        views of the library's memory; they keep this object alive."""
        out = []
        for ptr, off in ((self._head_ptr, self.head_off), (self._tail_ptr, self.tail_off)):
            n = int(off[-1]) if self.n_orfs else 0
            buf = (ctypes.c_char * max(1, n)).from_address(ptr) if n else ctypes.create_string_buffer(1)
            buf._owner = self._parsed
            out += [buf, off]
        return tuple(out)

    @property
    def extents(self) -> dict:
        """``{(strand, chrom): (lo, hi)}`` over the ORFs of each group."""
        return {k: (int(lo), int(hi)) for k, lo, hi in zip(self.group_keys, self.group_lo, self.group_hi)}

    def records(self):
        """Materialise ``detect_orfs.IndexRecord`` rows (tests / small inputs only)."""
        from .detect_orfs import IndexRecord

        out = []
        for i in range(self.n_orfs):
            oid, category = self.head[self.head_off[i] : self.head_off[i + 1]].decode("utf-8").split("\t")
            t = self.tail[self.tail_off[i] : self.tail_off[i + 1]].decode("utf-8").split("\t")
            ivs = tuple(
                (int(s), int(e))
                for s, e in zip(self.iv_start[self.orf_iv[i] : self.orf_iv[i + 1]], self.iv_end[self.orf_iv[i] : self.orf_iv[i + 1]])
            )
            out.append(IndexRecord(oid, category, t[0], t[1], t[2], t[3], t[4], t[5], t[6], None if t[7] == "None" else t[7], ivs))
        return out
