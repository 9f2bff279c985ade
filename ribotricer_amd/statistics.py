"""Host-side mirror of ``ribotricer.statistics`` for the phase-score path.

``phasescore`` keeps the reference's signature and return convention
(ribotricer/statistics.py:48-115: ``(np.float64 phase_score, int valid_codons)``)
but every number comes from the gfx950 kernels behind libribophase.so.  Callers in
the reference: detect_orfs.py:280 (int profiles), metagene.py:243-244 (float
profiles), utils.py:227-229.  ``RIBOTRICER_AMD_BACKEND=cpu`` (or ``auto`` on a machine
without a HIP device) takes the library's host entry points instead -- the reference's own
float64 operation sequence in C++ (``backend.py``); a failing device call is never retried there.
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np

from . import _lib, backend


def _is_integral(values: np.ndarray) -> bool:
    if values.dtype.kind in "iub":
        return True
    if values.dtype.kind == "f":
        return bool(np.all(np.isfinite(values)) and np.all(values == np.rint(values)))
    return False


def phasescore_batch(profiles: Sequence[Sequence[float]], device=None):
    """Phase score of many profiles in one launch: ``(phase float64[n], valid int32[n], flags uint8[n])``."""
    arrays = [np.asarray(list(p) if not isinstance(p, np.ndarray) else p) for p in profiles]
    arrays = [a.astype(np.float64) if a.size == 0 else a for a in arrays]
    if backend.selected() == "cpu":  # (before anything that needs torch: the cpu backend does not)
        return backend.phasescore_batch_host(arrays)
    import torch

    from .engine import get_engine

    lengths = np.array([a.size for a in arrays], np.int64)
    offsets = np.zeros(len(arrays) + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    eng = get_engine(device)
    integral = all(_is_integral(a) for a in arrays)
    if integral:
        flat = np.concatenate(arrays) if arrays and offsets[-1] else np.zeros(0)
        if flat.size and (flat.min() < 0 or flat.max() > _lib.MAX_COUNT):
            integral = False
    if integral:
        res = eng.score_host(flat.astype(np.int32), offsets)
        return res["phase"], res["valid"], res["flags"]
    flat = np.concatenate([a.astype(np.float64) for a in arrays]) if offsets[-1] else np.zeros(0)
    phase, valid, flags = eng.score_float_profiles(flat, offsets)
    torch.cuda.synchronize(eng.device)
    phase, valid, flags = phase.cpu().numpy(), valid.cpu().numpy(), flags.cpu().numpy()
    # Exact frame ties of float profiles (a few metagene profiles per sample, metagene.py:243-244):
    # the reference's strict `>` is decided by the last bits of its own float64 arithmetic, libm
    # pow() included -- replayed on the host for the flagged profiles (rp_tie_replay_f64_host).
    tied = np.flatnonzero(flags & _lib.FLAG_TIE)
    if tied.size:
        sub_off = np.zeros(tied.size + 1, np.int64)
        np.cumsum(lengths[tied], out=sub_off[1:])
        sub = np.concatenate([flat[offsets[i] : offsets[i + 1]] for i in tied]) if sub_off[-1] else np.zeros(0)
        phase[tied], valid[tied] = _lib.tie_replay_host(sub.astype(np.float64), sub_off)
        flags[tied] |= _lib.FLAG_REPLAY
    return phase, valid, flags


def phasescore(original_values: Sequence[float]) -> tuple[np.float64, int]:
    """Phase score of one coverage profile (same contract as the reference's ``phasescore``).

    Returns ``(periodicity_score, valid_codons)``: the square root of the maximal
    1/3-frequency coherence over the three reading frames and the number of
    non-all-zero codons of the winning frame.
    """
    phase, valid, _ = phasescore_batch([original_values])
    return np.float64(phase[0]), int(valid[0])
