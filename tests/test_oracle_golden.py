"""The oracle against outputs of the reference itself (tests/golden/).

These pin the two CPU restatements under oracle/ -- they are what every GPU
parity test is then measured against.
  * literal (scipy) restatement: bit-for-bit on phase and valid, every vector;
  * C closed form: phase <= 1e-12 abs, per-frame N bit-exact, valid bit-exact on
    every ORF whose frames are not tied (flag bit 0).
"""

import numpy as np
import pytest

from conftest import split_csr
from oracle import c_oracle
from oracle.phasescore_literal import phasescore_literal

PHASE_TOL = 1e-12


def _check_c_oracle(g):
    r = c_oracle.phase_score_csr(g["counts"], g["offsets"])
    tie = (r.flags & c_oracle.FLAG_TIE) != 0
    assert np.abs(r.phase - g["phase"]).max() <= PHASE_TOL
    assert np.array_equal(r.frame_n, g["frame_n"])
    both = ~np.isnan(g["frame_score"]) & ~np.isnan(r.frame_score)
    assert np.abs(g["frame_score"] - r.frame_score)[both].max() <= PHASE_TOL
    # NaN (M == 0) frames agree; the fixture stores NaN for empty frames too
    assert np.array_equal(np.isnan(g["frame_score"]), np.isnan(r.frame_score) | (r.frame_n == 0))
    mismatch = r.valid != g["valid"]
    assert not (mismatch & ~tie).any(), "valid_codons differs on an ORF that is not a frame tie"
    # on tie-flagged ORFs the reference's pick must still be one of the frames' N
    for i in np.nonzero(mismatch)[0]:
        assert g["valid"][i] in set(r.frame_n[i]) | {0}
    return r, tie


def test_c_oracle_known_answers(g1):
    for row in g1:
        c = np.array(row["input"], np.int32)
        r = c_oracle.phase_score_csr(c, np.array([0, c.size], np.int64))
        assert abs(r.phase[0] - row["phase"]) <= PHASE_TOL
        assert list(r.frame_n[0]) == row["frame_n"]
        if not r.flags[0] & c_oracle.FLAG_TIE:
            assert r.valid[0] == row["valid"], row["input"]


def test_c_oracle_poisson(g2):
    r, tie = _check_c_oracle(g2)
    assert tie.mean() < 0.02  # ties are a sparse-coverage corner, not the bulk
    # no ties once coverage is reasonable (0/1500 at lambda >= 0.3)
    lens = np.diff(g2["offsets"])
    assert not tie[(g2["lam"] >= 1.0) & (lens >= 60)].any()


def test_c_oracle_adversarial(g3):
    _check_c_oracle(g3)


def test_c_oracle_long(g4):
    r, tie = _check_c_oracle(g4)
    assert not tie.any()
    assert np.array_equal(r.valid, g4["valid"])


def test_c_oracle_float_profiles(g5):
    for row in g5:
        phase, valid, flags, *_ = c_oracle.phasescore_f64(row["input"])
        assert abs(phase - row["phase"]) <= PHASE_TOL
        if not flags & c_oracle.FLAG_TIE:
            assert valid == row["valid"]


def test_c_oracle_counts_and_codon_min():
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 40, size=300)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    counts = rng.poisson(0.7, size=int(offsets[-1])).astype(np.int32)
    r = c_oracle.phase_score_csr(counts, offsets)
    for i, v in enumerate(split_csr(counts, offsets)):
        assert r.read_count[i] == int(v.sum())  # detect_orfs.py:278
        codons = [int(v[k : k + 3].sum()) for k in range(0, len(v), 3)]  # common.py:164-180
        assert r.min_codon_cov[i] == (min(codons) if codons else np.iinfo(np.int32).max)


def test_c_oracle_openmp_matches_serial(g2):
    a = c_oracle.phase_score_csr(g2["counts"], g2["offsets"], n_threads=1)
    b = c_oracle.phase_score_csr(g2["counts"], g2["offsets"], n_threads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


def test_c_oracle_rejects_bad_offsets():
    with pytest.raises(ValueError):
        c_oracle.phase_score_csr(np.zeros(4, np.int32), np.array([1, 4], np.int64))
    with pytest.raises(ValueError):
        c_oracle.phase_score_csr(np.zeros(4, np.int32), np.array([0, 3, 2, 4], np.int64))


def test_literal_restatement_bit_exact_known_answers(g1):
    for row in g1:
        p, v = phasescore_literal(row["input"])
        assert repr(float(p)) == row["phase_repr"]
        assert v == row["valid"]


def test_literal_restatement_bit_exact_sample(g2, g3, g5):
    # a 300-vector sample per set keeps the CPU suite short; make_golden.py asserted all of them
    for g in (g2, g3):
        vecs = split_csr(g["counts"], g["offsets"])
        for i in range(0, len(vecs), max(1, len(vecs) // 300)):
            p, v = phasescore_literal(vecs[i].tolist())
            assert p == g["phase"][i] and v == g["valid"][i]
    for row in g5:
        p, v = phasescore_literal(row["input"])
        assert p == row["phase"] and v == row["valid"]


# ------------------------------------------------------------------ the scipy replay (oracle/scipy_replay.c)
@pytest.mark.parametrize("name", ["g2", "g3", "g4", "g10"])
def test_scipy_replay_is_the_reference_bit_for_bit(request, name):
    """The C replay of the reference's numpy/scipy arithmetic reproduces the reference's
    phase score (every bit), valid_codons and per-frame coherence on EVERY golden ORF, the
    exact frame ties included -- this is what the device-side tie replay is checked against."""
    g = request.getfixturevalue(name)
    r = c_oracle.replay_csr(g["counts"], g["offsets"])
    assert np.array_equal(r.valid, g["valid"])
    assert np.array_equal(r.phase, g["phase"])
    assert np.array_equal(r.frame_n, g["frame_n"])
    live = g["frame_n"] > 0
    assert np.array_equal(r.frame_score[live], g["frame_score"][live], equal_nan=True)


def test_scipy_replay_known_answers(g1):
    for row in g1:
        c = np.array(row["input"], np.int32)
        r = c_oracle.replay_csr(c, np.array([0, c.size], np.int64))
        assert r.phase[0] == row["phase"] and r.valid[0] == row["valid"], row["input"]


def test_c_oracle_counts_beyond_fp32(g10):
    """G10: counts 2^24 .. 2^30 (the reference has no limit).  The float64 closed form stays within 1e-9 of the
    reference, valid_codons equal off the exact ties, the integer sums are int64 and the int32 codon minimum
    saturates below the 'empty' sentinel when a codon sum passes int32."""
    r = c_oracle.phase_score_csr(g10["counts"], g10["offsets"])
    assert np.abs(r.phase - g10["phase"]).max() <= 1e-9
    tie = (r.flags & 1) != 0
    assert np.array_equal(r.valid[~tie], g10["valid"][~tie]) and 0 < tie.sum() < tie.size
    counts, offsets = g10["counts"].astype(np.int64), g10["offsets"]
    assert int(counts.max()) >= 1 << 30 and int(counts.max()) > 16777215
    sat = 0
    for i in range(offsets.size - 1):
        prof = counts[offsets[i] : offsets[i + 1]]
        assert r.read_count[i] == prof.sum()
        mn = np.pad(prof, (0, -prof.size % 3)).reshape(-1, 3).sum(axis=1).min()
        assert r.min_codon_cov[i] == min(mn, 2**31 - 2)
        sat += mn > 2**31 - 2
    assert sat > 0  # (kind 2 of the generator: every codon of the ORF past 2^31)
