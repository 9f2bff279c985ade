"""Batch phase scoring on one MI355X through the C ABI (include/ribophase.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every number
is produced by the hand-written gfx950 kernels in ``csrc/``.  There is no CPU
path -- without a GPU (or without the built library) the calls raise.

The batch call replaces the per-ORF body of the reference's hot loop
(ribotricer/detect_orfs.py:274-299): ``sum(cov)``, ``phasescore(cov)``
(statistics.py:48-115), ``collapse_coverage_to_codon`` (common.py:164-180) and the
status predicate, for all ORFs of a CSR-packed batch at once.
"""

from __future__ import annotations

import ctypes
from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _lib
from ._lib import FilterParams, RibophaseError
from .const import (
    CUTOFF,
    MINIMUM_DENSITY_OVER_ORF,
    MINIMUM_READS_PER_CODON,
    MINIMUM_VALID_CODONS,
    MINIMUM_VALID_CODONS_RATIO,
)


class PhaseScores(NamedTuple):
    """Per-ORF results, device tensors of length n_orfs."""

    phase: torch.Tensor  # float64  np.sqrt(coh), statistics.py:115
    valid: torch.Tensor  # int32    valid codons of the winning frame
    read_count: torch.Tensor  # int64    sum(cov), detect_orfs.py:278
    min_codon_cov: torch.Tensor  # int32    min codon sum (MIN_CODON_COV_EMPTY if L == 0)
    flags: torch.Tensor  # uint8    FLAG_TIE | FLAG_RECHECK64 | FLAG_SPLIT
    status: Optional[torch.Tensor]  # uint8    1 = translating (None when no thresholds given)

    def cpu_numpy(self) -> dict:
        return {k: (None if v is None else v.cpu().numpy()) for k, v in self._asdict().items()}


class FrameDiagnostics(NamedTuple):
    score: torch.Tensor  # float64 [n,3]  (NaN when M == 0 < N, 0 when N == 0)
    n: torch.Tensor  # int32 [n,3]
    m: torch.Tensor  # int32 [n,3]


def make_filter(
    phase_score_cutoff: float = CUTOFF,
    min_valid_codons: int = MINIMUM_VALID_CODONS,
    min_reads_per_codon: float = MINIMUM_READS_PER_CODON,
    min_valid_codons_ratio: float = MINIMUM_VALID_CODONS_RATIO,
    min_density_over_orf: float = MINIMUM_DENSITY_OVER_ORF,
) -> FilterParams:
    """Thresholds in the argument order of export_orf_coverages (detect_orfs.py:206-216)."""
    fp = FilterParams()
    fp.phase_score_cutoff = float(phase_score_cutoff)
    fp.min_valid_codons = int(min_valid_codons)
    fp.min_reads_per_codon = float(min_reads_per_codon)
    fp.min_valid_codons_ratio = float(min_valid_codons_ratio)
    fp.min_density_over_orf = float(min_density_over_orf)
    return fp


def _require_gpu() -> None:
    if not torch.cuda.is_available():
        # let the library produce its own diagnostic; it has no CPU path either
        _lib.device_count()
        raise RibophaseError(-6, "no HIP device visible to PyTorch; ribotricer_amd has no CPU fallback")


def _as_device(x, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    elif not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if x.dtype != dtype:
        x = x.to(dtype)
    if x.device != device:
        x = x.to(device, non_blocking=True)
    return x.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None or t.numel() == 0 else t.data_ptr())


def _alloc_outputs(dev: torch.device, n: int, with_status: bool) -> PhaseScores:
    return PhaseScores(
        torch.empty(n, dtype=torch.float64, device=dev),
        torch.empty(n, dtype=torch.int32, device=dev),
        torch.empty(n, dtype=torch.int64, device=dev),
        torch.empty(n, dtype=torch.int32, device=dev),
        torch.empty(n, dtype=torch.uint8, device=dev),
        torch.empty(n, dtype=torch.uint8, device=dev) if with_status else None,
    )


class PhaseScoreEngine:
    """Owns the reusable device buffers (outputs + workspace) for one GPU."""

    def __init__(self, device=None):
        _lib.load()
        _require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("PhaseScoreEngine needs a cuda (HIP) device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._workspace: Optional[torch.Tensor] = None
        self._out: Optional[PhaseScores] = None

    # -- buffers -----------------------------------------------------------------
    def _get_workspace(self, nbytes: int) -> Optional[torch.Tensor]:
        if nbytes == 0:
            return None
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._workspace

    def _get_outputs(self, n: int, with_status: bool) -> PhaseScores:
        o = self._out
        if o is None or o.phase.numel() != n or (with_status and o.status is None):
            o = _alloc_outputs(self.device, n, with_status)
            self._out = o
        if not with_status and o.status is not None:
            o = o._replace(status=None)
        return o

    # -- the hot path --------------------------------------------------------------
    def score(
        self,
        counts,
        offsets,
        thresholds: Optional[FilterParams] = None,
        algo: str = "auto",
        reuse_outputs: bool = False,
        timings: Optional[list] = None,
    ) -> PhaseScores:
        """Score every ORF of a CSR batch (counts int32, offsets int64, offsets[-1] == len(counts)).

        Asynchronous on the current torch stream unless ``timings`` is a list, in
        which case the call blocks and appends [index_ms, main_ms, finish_ms, total_ms].
        """
        dev = self.device
        counts = _as_device(counts, torch.int32, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        if offsets.dim() != 1 or offsets.numel() < 1 or counts.dim() != 1:
            raise ValueError("counts and offsets must be 1-D; offsets needs n_orfs+1 entries")
        n = offsets.numel() - 1
        total_nt = counts.numel()
        algo_id = _lib.ALGOS[algo]
        ws = self._get_workspace(_lib.workspace_bytes(n, total_nt, algo_id))
        with_status = thresholds is not None
        out = self._get_outputs(n, with_status) if reuse_outputs else _alloc_outputs(dev, n, with_status)
        lib = _lib.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        args = [
            dev.index,
            _ptr(counts),
            _ptr(offsets),
            n,
            total_nt,
            _ptr(out.phase),
            _ptr(out.valid),
            _ptr(out.read_count),
            _ptr(out.min_codon_cov),
            _ptr(out.flags),
            _ptr(out.status),
            ctypes.byref(thresholds) if thresholds is not None else None,
            _ptr(ws),
            0 if ws is None else ws.numel(),
            algo_id,
            stream,
        ]
        if timings is None:
            _lib.check(lib.rp_phase_score_csr_dev(*args))
        else:
            ms = (ctypes.c_float * 4)()
            _lib.check(lib.rp_phase_score_csr_dev_timed(*args, ctypes.byref(ms)))
            timings.append([float(x) for x in ms])
        return out

    def validate(self, counts, offsets) -> None:
        """Synchronous input check; raises RibophaseError on bad offsets / counts."""
        dev = self.device
        counts = _as_device(counts, torch.int32, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(
            _lib.load().rp_validate_csr_dev(
                dev.index, _ptr(counts), _ptr(offsets), offsets.numel() - 1, counts.numel(), stream
            )
        )

    def frames(self, counts, offsets) -> FrameDiagnostics:
        """Per-frame float64 (score_f, N_f, M_f) -- what statistics.py:67-108 computes per frame."""
        dev = self.device
        counts = _as_device(counts, torch.int32, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        n = offsets.numel() - 1
        d = FrameDiagnostics(
            torch.empty((n, 3), dtype=torch.float64, device=dev),
            torch.empty((n, 3), dtype=torch.int32, device=dev),
            torch.empty((n, 3), dtype=torch.int32, device=dev),
        )
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(
            _lib.load().rp_phase_score_frames_dev(
                dev.index, _ptr(counts), _ptr(offsets), n, _ptr(d.score), _ptr(d.n), _ptr(d.m), stream
            )
        )
        return d

    def score_float_profiles(self, values, offsets):
        """float64 profiles (metagene.py:243-244 -> statistics.py:48): (phase, valid, flags)."""
        dev = self.device
        values = _as_device(values, torch.float64, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        n = offsets.numel() - 1
        phase = torch.empty(n, dtype=torch.float64, device=dev)
        valid = torch.empty(n, dtype=torch.int32, device=dev)
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(
            _lib.load().rp_phase_score_f64_csr_dev(
                dev.index, _ptr(values), _ptr(offsets), n, _ptr(phase), _ptr(valid), _ptr(flags), stream
            )
        )
        return phase, valid, flags


_engines: dict = {}


def get_engine(device=None) -> PhaseScoreEngine:
    """Process-wide engine per device."""
    _require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if dev not in _engines:
        _engines[dev] = PhaseScoreEngine(dev)
    return _engines[dev]


def phase_score_csr(counts, offsets, thresholds: Optional[FilterParams] = None, algo: str = "auto", device=None) -> PhaseScores:
    """Functional form of :meth:`PhaseScoreEngine.score`."""
    return get_engine(device).score(counts, offsets, thresholds=thresholds, algo=algo)
