"""Backend selection, and the GPU-less backend (SURVEY.md sections 2.1(iii), 5 and 8(b); BASELINE configs[0]).

``RIBOTRICER_AMD_BACKEND`` = ``hip`` | ``cpu`` | ``auto`` (default):

* ``hip``  the gfx950 kernels behind the ``*_dev`` entry points of libribophase.so.  A missing GPU raises
           ``RibophaseError``; nothing is retried anywhere else.
* ``cpu``  ``export_orf_coverages`` / ``phasescore`` entirely on the host, through the ``*_host`` entry points of the SAME
           library: index parser -> ``rp_gather_profiles_host`` (orf_coverage, detect_orfs.py:134-203) ->
           ``rp_phase_score_csr_host`` (the loop body of detect_orfs.py:274-299 in the reference's own float64 operation
           sequence, csrc/rp_replay.hpp: every ORF carries the reference's bits, ties or not) -> ``rp_format_rows_host``.
           No HIP call, and torch is not imported (tests/test_host_backend_cpu.py runs it with torch made unimportable).
           What the host still needs: numpy, and the ROCm runtime LIBRARIES on the loader path -- libribophase.so is one
           library for both backends and links libamdhip64 -- but no GPU and no driver.
           Byte-identical TSVs to the reference's, phase column included
           (tests/test_host_backend_cpu.py against the reference's own outputs G6 / G10).
* ``auto`` ``hip`` when a HIP device is visible (``torch.cuda.device_count() > 0``: counting devices does not
           initialise one), else ``cpu`` with one line on stderr.  The choice is made from what is VISIBLE, once per call,
           before any work: a device call that fails is an error under ``auto`` exactly as under ``hip`` -- there is no
           fallback behind a failure.

Nothing here touches ``oracle/`` (test infrastructure).
"""

from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

from . import _lib

_CHOICES = ("auto", "hip", "cpu")
_told = [False]
POS_BITS = 40  # (group << 40) | position: the key of rp_gather_profiles_host


def selected() -> str:
    """``"hip"`` or ``"cpu"`` for this call (module docstring)."""
    want = os.environ.get("RIBOTRICER_AMD_BACKEND", "auto").strip().lower() or "auto"
    if want not in _CHOICES:
        raise ValueError(f"RIBOTRICER_AMD_BACKEND must be one of {_CHOICES}, got {want!r}")
    if want != "auto":
        return want
    try:
        import torch

        if torch.cuda.device_count() > 0:
            return "hip"
    except ImportError:  # (a host without torch has no hip backend to offer)
        pass
    if not _told[0]:
        _told[0] = True
        print("ribotricer_amd: no HIP device visible -- using the CPU backend (the reference's float64 arithmetic in C++, "
              "libribophase *_host entry points); RIBOTRICER_AMD_BACKEND=hip insists on the GPU", file=sys.stderr)
    return "cpu"


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data if a.size else 0)


def merged_keys(merged, index):
    """The merged P-site counts (``alignments.MergedColumns`` or the reference's ``strand -> Counter``) as the sorted table
    ``rp_gather_profiles_host`` looks positions up in: ``(keys int64, vals int64)``, key = (group of the index << 40) |
    position, rows of one position added up (several read lengths can land on one P-site: detect_orfs.py:76-82 adds them
    in the Counter).  Rows on a (strand, chrom) without candidate ORFs, or outside its ORFs' extent, are never looked up
    by the reference (detect_orfs.py:176-187) and are dropped here."""
    from .alignments import MergedColumns

    cols = merged if isinstance(merged, MergedColumns) else MergedColumns.from_counters(merged)
    if cols.pos.size == 0 or not index.group_keys:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    group = cols.group_codes(index.group_keys).astype(np.int64)
    pos = np.asarray(cols.pos, np.int64)
    lo = np.asarray(index.group_lo, np.int64)
    hi = np.minimum(np.asarray(index.group_hi, np.int64), (1 << POS_BITS) - 1)
    g = np.maximum(group, 0)
    keep = (group >= 0) & (pos >= np.maximum(lo[g], 0)) & (pos <= hi[g])
    keys = (group[keep] << POS_BITS) | pos[keep]
    vals = np.asarray(cols.count, np.int64)[keep]
    if keys.size == 0:
        return keys, vals
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], vals[order]
    first = np.flatnonzero(np.concatenate(([True], keys[1:] != keys[:-1])))
    return np.ascontiguousarray(keys[first]), np.ascontiguousarray(np.add.reduceat(vals, first))


def gather_profiles_host(index, merged, n_threads: int = 0):
    """``orf_coverage`` (detect_orfs.py:134-203) for every ORF of a natively parsed index on the host:
    ``(counts int32[sum L], offsets int64[n + 1])`` -- the same bytes as the device gathers."""
    keys, vals = merged_keys(merged, index)
    n = int(index.n_orfs)
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(np.asarray(index.length, np.int64), out=offsets[1:])
    counts = np.empty(int(offsets[-1]), np.int32)
    arrays = [np.ascontiguousarray(a, dtype=t) for a, t in (
        (index.iv_start, np.int64), (index.iv_end, np.int64), (index.orf_iv, np.int64), (index.group, np.int32), (index.reverse, np.uint8))]
    _lib.check(_lib.load().rp_gather_profiles_host(_ptr(keys), _ptr(vals), keys.size, *[_ptr(a) for a in arrays], _ptr(offsets), n,
                                                  _ptr(counts), int(n_threads)))
    return counts, offsets


def score_index_host(index, merged, thresholds, n_threads: int = 0, timings=None):
    """Gather + score of ``detect_orfs.score_index`` on the host: ``(counts, offsets, results)`` with every ORF's profile
    (the row formatter skips the non-translating ones itself unless ``report_all``)."""
    import time

    t0 = time.perf_counter()
    counts, offsets = gather_profiles_host(index, merged, n_threads)
    t1 = time.perf_counter()
    res = _lib.phase_score_csr_host(counts, offsets, thresholds, n_threads=n_threads)
    if timings is not None:
        timings["host_gather"] = t1 - t0
        timings["host_score"] = time.perf_counter() - t1
    return counts, offsets, res


def phasescore_batch_host(arrays):
    """``statistics.phasescore_batch`` on the host: ``(phase, valid, flags)``.  Integer profiles whose counts fit int32 go
    through ``rp_tie_replay_host``, everything else through ``rp_tie_replay_f64_host`` -- both are statistics.py:48-115
    operation for operation (integers are converted exactly as Python's float arithmetic sees them)."""
    lengths = np.array([a.size for a in arrays], np.int64)
    offsets = np.zeros(len(arrays) + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    flat = np.concatenate(arrays) if arrays and offsets[-1] else np.zeros(0)
    integral = flat.dtype.kind in "iub" and (flat.size == 0 or (int(flat.min()) >= 0 and int(flat.max()) <= np.iinfo(np.int32).max))
    values = flat.astype(np.int32) if integral else flat.astype(np.float64)
    phase, valid = _lib.tie_replay_host(values, offsets)
    return phase, valid, np.zeros(len(arrays), np.uint8)
