"""ctypes loader for oracle/phase_oracle.c -- TEST INFRASTRUCTURE ONLY.

Built by ``make -C oracle`` (also by ``__graft_entry__.build()``) into
oracle/_build/.  Not imported by the product path.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import NamedTuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

FLAG_TIE = 1


class OracleResult(NamedTuple):
    phase: np.ndarray  # float64 [n]
    valid: np.ndarray  # int32   [n]
    read_count: np.ndarray  # int64 [n]
    min_codon_cov: np.ndarray  # int32 [n]
    flags: np.ndarray  # uint8 [n]
    frame_score: np.ndarray  # float64 [n,3]
    frame_n: np.ndarray  # int32 [n,3]
    frame_m: np.ndarray  # int32 [n,3]


def build(force: bool = False) -> None:
    """Compile the C restatement with gcc (a few hundred ms)."""
    srcs = [os.path.join(_HERE, "phase_oracle.c"), os.path.join(_HERE, "scipy_replay.c")]
    lib = os.path.join(_BUILD, "libphase_oracle.so")
    if (
        not force
        and os.path.exists(lib)
        and os.path.getmtime(lib) >= max(os.path.getmtime(s) for s in srcs)
        and os.path.exists(os.path.join(_BUILD, "libphase_oracle_omp.so"))
    ):
        return
    subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


_libs: dict = {}


def _load(openmp: bool) -> ctypes.CDLL:
    key = bool(openmp)
    if key in _libs:
        return _libs[key]
    name = "libphase_oracle_omp.so" if openmp else "libphase_oracle.so"
    path = os.path.join(_BUILD, name)
    if not os.path.exists(path):
        build()
    lib = ctypes.CDLL(path)
    p = ctypes.c_void_p
    lib.rp_oracle_phase_score_csr.restype = ctypes.c_int
    lib.rp_oracle_phase_score_csr.argtypes = [p, p, ctypes.c_int64, p, p, p, p, p, p, p, p, ctypes.c_int]
    lib.rp_oracle_replay_csr.restype = ctypes.c_int
    lib.rp_oracle_replay_csr.argtypes = [p, p, ctypes.c_int64, p, p, p, p]
    lib.rp_oracle_phase_score_f64.restype = ctypes.c_int
    lib.rp_oracle_phase_score_f64.argtypes = [p, ctypes.c_int64, p, p, p, p, p, p]
    _libs[key] = lib
    return lib


def _ptr(a: np.ndarray) -> ctypes.c_void_p:
    return ctypes.c_void_p(a.ctypes.data)


def phase_score_csr(counts, offsets, n_threads: int = 1) -> OracleResult:
    """Closed-form float64 phase score of every ORF of a CSR batch (C, host)."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    if n < 0:
        raise ValueError("offsets must have n_orfs+1 entries")
    res = OracleResult(
        np.empty(n, np.float64),
        np.empty(n, np.int32),
        np.empty(n, np.int64),
        np.empty(n, np.int32),
        np.empty(n, np.uint8),
        np.empty((n, 3), np.float64),
        np.empty((n, 3), np.int32),
        np.empty((n, 3), np.int32),
    )
    lib = _load(openmp=n_threads > 1)
    rc = lib.rp_oracle_phase_score_csr(
        _ptr(counts), _ptr(offsets), n, *[_ptr(a) for a in res], int(n_threads)
    )
    if rc != 0:
        raise ValueError(f"rp_oracle_phase_score_csr failed with code {rc}")
    return res


def phasescore_f64(values):
    """Single float profile -> (phase, valid, flags, frame_score[3], frame_n[3], frame_m[3])."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    phase = np.empty(1, np.float64)
    valid = np.empty(1, np.int32)
    flags = np.empty(1, np.uint8)
    fs = np.empty(3, np.float64)
    fn = np.empty(3, np.int32)
    fm = np.empty(3, np.int32)
    rc = _load(False).rp_oracle_phase_score_f64(
        _ptr(v), v.size, _ptr(phase), _ptr(valid), _ptr(flags), _ptr(fs), _ptr(fn), _ptr(fm)
    )
    if rc != 0:
        raise ValueError(f"rp_oracle_phase_score_f64 failed with code {rc}")
    return float(phase[0]), int(valid[0]), int(flags[0]), fs, fn, fm


class ReplayResult(NamedTuple):
    phase: np.ndarray  # float64 [n]  the reference's own bits (np.sqrt of the winning Cxy)
    valid: np.ndarray  # int32 [n]
    frame_score: np.ndarray  # float64 [n,3]  Cxy[f=1/3] per frame (NaN for an empty frame)
    frame_n: np.ndarray  # int32 [n,3]


def replay_csr(counts, offsets) -> ReplayResult:
    """Bit-for-bit replay of the reference's numpy/scipy arithmetic (oracle/scipy_replay.c)."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    res = ReplayResult(np.empty(n, np.float64), np.empty(n, np.int32), np.empty((n, 3), np.float64), np.empty((n, 3), np.int32))
    rc = _load(False).rp_oracle_replay_csr(_ptr(counts), _ptr(offsets), n, _ptr(res.phase), _ptr(res.valid),
                                           _ptr(res.frame_score), _ptr(res.frame_n))
    if rc != 0:
        raise ValueError(f"rp_oracle_replay_csr failed with status {rc}")
    return res
