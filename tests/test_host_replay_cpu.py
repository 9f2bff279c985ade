"""The product's HOST tie resolver (rp_tie_replay_host / rp_tie_replay_f64_host, csrc/rp_replay.hpp)
against outputs of the reference itself: it restates statistics.py:48-115 float64 operation for
float64 operation (libm pow() included), so it must return the reference's (phase, valid_codons)
BIT FOR BIT on every golden vector -- ties or not, counts >= 16 or not, integer or float.
No GPU involved: the resolver is host code of libribophase.so."""

import numpy as np
import pytest

from ribotricer_amd import _lib


def _csr_of(vectors, dtype):
    lens = np.array([len(v) for v in vectors], np.int64)
    off = np.zeros(len(vectors) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.concatenate([np.asarray(v, dtype) for v in vectors]) if off[-1] else np.zeros(0, dtype)
    return flat, off


@pytest.mark.parametrize("name", ["g2", "g3", "g4", "g8", "g10"])
def test_int_profiles_carry_the_references_bits(request, name):
    g = request.getfixturevalue(name)
    phase, valid = _lib.tie_replay_host(g["counts"], g["offsets"])
    assert np.array_equal(valid, g["valid"])
    assert np.array_equal(phase, g["phase"])  # bitwise (no NaN: the reference returns sqrt(coh), coh >= 0)


def test_known_answers(g1):
    ints = [r for r in g1 if all(float(x).is_integer() for x in r["input"])]
    flat, off = _csr_of([r["input"] for r in ints], np.int32)
    phase, valid = _lib.tie_replay_host(flat, off)
    for r, p, v in zip(ints, phase, valid):
        assert (p, v) == (r["phase"], r["valid"]), r["input"]


@pytest.mark.parametrize("name", ["g5", "g8f"])
def test_float_profiles_carry_the_references_bits(request, name):
    rows = request.getfixturevalue(name)
    flat, off = _csr_of([r["input"] for r in rows], np.float64)
    phase, valid = _lib.tie_replay_host(flat, off)
    assert [int(v) for v in valid] == [r["valid"] for r in rows]
    assert [float(p) for p in phase] == [r["phase"] for r in rows]


def test_g8_really_holds_big_ties(g8):
    """The fixture must exercise what it is for: exact frame ties whose codons hold counts >= 16,
    among them ORFs where squaring with x*x instead of libm pow() changes the outcome's bits."""
    from oracle import c_oracle

    o = c_oracle.phase_score_csr(g8["counts"], g8["offsets"])
    tie = (o.flags & 1) != 0
    big = np.array([g8["counts"][a:b].max(initial=0) >= 16 for a, b in zip(g8["offsets"][:-1], g8["offsets"][1:])])
    assert (tie & big).sum() > 300
    assert ((o.valid != g8["valid"]) & ~tie).sum() == 0  # the closed form is right wherever no tie is flagged


def test_bad_arguments():
    with pytest.raises(_lib.RibophaseError):
        _lib.tie_replay_host(np.zeros(3, np.int32), np.array([0, 3, 1], np.int64))  # decreasing offsets
    phase, valid = _lib.tie_replay_host(np.zeros(0, np.int32), np.zeros(1, np.int64))
    assert phase.size == 0 and valid.size == 0


@pytest.mark.parametrize("name", ["g2", "g3", "g4", "g8", "g10"])
def test_host_batch_entry_point_equals_the_reference(request, name):
    """rp_phase_score_csr_host: phase and valid_codons are the reference's on EVERY ORF of
    the golden sets (bitwise), the integer results equal the C oracle's, status equals the predicate."""
    from helpers import reference_status
    from oracle import c_oracle
    from ribotricer_amd.engine import make_filter

    g = request.getfixturevalue(name)
    th = make_filter(phase_score_cutoff=0.3, min_valid_codons=3, min_reads_per_codon=0, min_valid_codons_ratio=0.05, min_density_over_orf=0.1)
    one = _lib.phase_score_csr_host(g["counts"], g["offsets"], thresholds=th, n_threads=1)
    many = _lib.phase_score_csr_host(g["counts"], g["offsets"], thresholds=th, n_threads=5)
    for k in one:
        assert np.array_equal(one[k], many[k]), k
    assert np.array_equal(one["phase"], g["phase"]) and np.array_equal(one["valid"], g["valid"])
    o = c_oracle.phase_score_csr(g["counts"], g["offsets"])
    assert np.array_equal(one["read_count"], o.read_count) and np.array_equal(one["min_codon_cov"], o.min_codon_cov)
    assert not one["flags"].any()
    want = reference_status(one["phase"], one["valid"], o.read_count, o.min_codon_cov, np.diff(g["offsets"]), cutoff=0.3, min_valid=3,
                            min_reads=0, min_ratio=0.05, min_density=0.1)
    assert np.array_equal(one["status"], want)
    no_status = _lib.phase_score_csr_host(g["counts"][:0], np.zeros(3, np.int64))
    assert no_status["status"] is None and np.all(no_status["phase"] == 0) and np.all(no_status["min_codon_cov"] == 2147483647)
