"""Helpers shared by CPU and GPU tests."""

import numpy as np

from oracle import c_oracle

INT32_MAX = np.iinfo(np.int32).max


def reference_status(phase, valid, read_count, min_codon_cov, lengths, cutoff=0.428571428571, min_valid=5,
                     min_reads=0, min_ratio=0, min_density=0.0):
    """Status predicate of ribotricer/detect_orfs.py:281,285-299 on arrays (numpy restatement)."""
    n_codons = np.maximum(1, lengths // 3)
    ratio = valid / n_codons
    density = read_count / n_codons
    ok = (
        (phase >= cutoff)
        & (valid >= min_valid)
        & (min_codon_cov >= min_reads)
        & (ratio >= min_ratio)
        & (density >= min_density)
    )
    return ok.astype(np.uint8)


def assert_matches_oracle(res, counts, offsets, phase_tol=1e-6, oracle=None, check_flags=True):
    """res: dict of numpy arrays from the HIP path; compares with the C oracle on the same bytes."""
    o = oracle if oracle is not None else c_oracle.phase_score_csr(counts, offsets, n_threads=4)
    assert np.array_equal(res["read_count"], o.read_count), "read_count must be bit-exact"
    assert np.array_equal(res["min_codon_cov"], o.min_codon_cov), "min_codon_cov must be bit-exact"
    dphase = np.abs(res["phase"] - o.phase)
    assert dphase.max(initial=0.0) <= phase_tol, f"phase differs by {dphase.max()}"
    tie_gpu = (res["flags"] & 1) != 0
    tie_cpu = (o.flags & 1) != 0
    bad = (res["valid"] != o.valid) & ~(tie_gpu | tie_cpu)
    assert not bad.any(), f"valid_codons differs on {bad.sum()} non-tie ORFs, first {np.nonzero(bad)[0][:5]}"
    # the exact-arithmetic tie rule is deterministic, so tie ORFs must agree as well
    assert np.array_equal(res["valid"], o.valid), "valid_codons differs on tie-flagged ORFs"
    if check_flags:
        assert np.array_equal(tie_gpu, tie_cpu), "tie flags differ from the oracle's"
    return o


def assert_matches_fixture(res, g, phase_tol=1e-6):
    """res vs outputs of the reference itself (tests/golden)."""
    assert np.abs(res["phase"] - g["phase"]).max(initial=0.0) <= phase_tol
    tie = (res["flags"] & 1) != 0
    bad = (res["valid"] != g["valid"]) & ~tie
    assert not bad.any(), f"valid_codons differs from the reference on {bad.sum()} unflagged ORFs"
    return tie
