"""Literal CPU restatement of ribotricer's ``phasescore`` -- TEST INFRASTRUCTURE.

Follows /root/reference/ribotricer/statistics.py:48-115 operation by operation
(same floating-point expression order, same ``scipy.signal.coherence`` call) but
is written independently and split into per-frame pieces so that tests can look
at the per-frame (score_f, N_f) pairs the reference never exposes.

Pinned: tests/test_oracle_golden.py checks this function bit-for-bit against
outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_golden.py in the build container where /root/reference is
importable).  Third-party arithmetic on the path: ``scipy.signal.coherence``
(scipy 1.15.3 here; reference requires scipy>=1.7.0, pyproject.toml:48) and
numpy 2.2.6 -- un-vendored, so the tie-class outputs recorded in the fixtures
encode those versions' rounding.

Not used by the product path.
"""

from __future__ import annotations

import warnings
from math import cos, pi, sin, sqrt

import numpy as np
from scipy import signal

# statistics.py:75-84 evaluates these four constants inline for every codon.
_COS1 = cos(2 * pi / 3)
_COS2 = cos(4 * pi / 3)
_SIN1 = sin(2 * pi / 3)
_SIN2 = sin(4 * pi / 3)

_WINDOW = np.array([1.0, 1.0, 1.0])


def frame_normalized_triplets(values: list, frame: int) -> list:
    """Normalised codon triplets of one reading frame.

    statistics.py:68-91: walk ``values[frame:]`` three at a time while a full
    triplet remains, skip all-zero triplets, divide the others by the modulus of
    ``a + b*e^{2*pi*i/3} + c*e^{4*pi*i/3}`` (modulus 0 -> 1).
    """
    flat: list = []
    n = len(values)
    for i in range(frame, n - 2, 3):
        a = values[i]
        b = values[i + 1]
        c = values[i + 2]
        if a == b == c == 0:  # statistics.py:72
            continue
        real = a + b * _COS1 + c * _COS2  # statistics.py:75-79
        image = b * _SIN1 + c * _SIN2  # statistics.py:80-82
        norm = sqrt(real**2 + image**2)  # statistics.py:83
        if norm == 0:
            norm = 1  # statistics.py:84-85
        flat.append(a / norm)
        flat.append(b / norm)
        flat.append(c / norm)
    return flat


def frame_coherence(flat: list) -> float:
    """The reference's coherence call for one frame (statistics.py:97-108).

    ``flat`` must be non-empty with a length that is a multiple of 3.
    """
    n_seg = len(flat) // 3
    x = np.array(flat[: n_seg * 3])
    y = np.array([1, 0, 0] * n_seg)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f, cxy = signal.coherence(x, y, window=_WINDOW, nperseg=3, noverlap=0)
    return cxy[np.argwhere(np.isclose(f, 1 / 3.0))[0]][0]


def phasescore_frames(original_values) -> list:
    """Per-frame ``(score_f, N_f)``; ``score_f`` is None for an empty frame."""
    values = list(original_values)
    frames = []
    for frame in (0, 1, 2):
        flat = frame_normalized_triplets(values, frame)
        n_seg = len(flat) // 3
        if n_seg == 0:
            frames.append((None, 0))
        else:
            frames.append((frame_coherence(flat), n_seg))
    return frames


def combine_frames(frames) -> tuple:
    """Frame state machine of statistics.py:64-66,94-95,109-115.

    An empty frame RESETS (coh, valid) to (0.0, 0); a frame wins only with a
    strictly greater score (NaN never wins); ``valid`` falls back to the first
    non-empty frame's N while it is still -1.
    """
    coh = 0.0
    valid = -1
    for score, n_seg in frames:
        if n_seg == 0:
            coh, valid = 0.0, 0
            continue
        if score > coh:
            coh = score
            valid = n_seg
        if valid == -1:
            valid = n_seg
    return np.sqrt(coh), valid


def phasescore_literal(original_values) -> tuple:
    """Same signature and result as the reference's ``phasescore``."""
    return combine_frames(phasescore_frames(original_values))
