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


def _sub_csr(counts, offsets, idx):
    lens = (offsets[1:] - offsets[:-1])[idx]
    off = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    parts = [counts[offsets[i] : offsets[i + 1]] for i in idx]
    return (np.concatenate(parts) if parts and off[-1] else np.zeros(0, np.int32)), off


def check_tie_replay(res, counts, offsets):
    """Tie-flagged ORFs: the engine replays the reference's own float64 arithmetic on the device
    (RP_FLAG_REPLAY); its phase score and valid_codons must equal oracle/scipy_replay.c -- which
    is bit-identical to the reference on every golden vector -- BIT FOR BIT.  The only licence:
    a codon with a count >= 16 takes x*x instead of the tabulated libm pow() on the device; ORFs
    holding such a codon may differ and are returned as a count."""
    counts = np.asarray(counts)
    offsets = np.asarray(offsets)
    tie = (res["flags"] & 1) != 0
    assert np.array_equal(tie, (res["flags"] & 8) != 0), "every tie-flagged ORF (and no other) must be replayed"
    idx = np.nonzero(tie)[0]
    if idx.size == 0:
        return 0
    c, o = _sub_csr(counts, offsets, idx)
    rep = c_oracle.replay_csr(c, o)
    big = np.array([counts[offsets[i] : offsets[i + 1]].max(initial=0) >= 16 for i in idx])
    same = (res["valid"][idx] == rep.valid) & (res["phase"][idx] == rep.phase)
    assert same[~big].all(), f"device replay differs from the reference's bits on {(~same[~big]).sum()} tie ORFs"
    return int((~same).sum())


def assert_matches_oracle(res, counts, offsets, phase_tol=1e-6, oracle=None, check_flags=True):
    """res: dict of numpy arrays from the HIP path; compares with the C oracle on the same bytes:
    closed form (phase_oracle.c) everywhere, the scipy replay (scipy_replay.c) on frame ties."""
    o = oracle if oracle is not None else c_oracle.phase_score_csr(counts, offsets, n_threads=4)
    assert np.array_equal(res["read_count"], o.read_count), "read_count must be bit-exact"
    assert np.array_equal(res["min_codon_cov"], o.min_codon_cov), "min_codon_cov must be bit-exact"
    dphase = np.abs(res["phase"] - o.phase)
    assert dphase.max(initial=0.0) <= phase_tol, f"phase differs by {dphase.max()}"
    tie_gpu = (res["flags"] & 1) != 0
    tie_cpu = (o.flags & 1) != 0
    bad = (res["valid"] != o.valid) & ~(tie_gpu | tie_cpu)
    assert not bad.any(), f"valid_codons differs on {bad.sum()} non-tie ORFs, first {np.nonzero(bad)[0][:5]}"
    if check_flags:
        assert np.array_equal(tie_gpu, tie_cpu), "tie flags differ from the oracle's"
    check_tie_replay(res, counts, offsets)
    return o


def assert_matches_fixture(res, g, phase_tol=1e-6):
    """res vs outputs of the reference itself (tests/golden): phase within tolerance everywhere,
    valid_codons identical on EVERY ORF -- the tie-flagged ones included, and those bit-exact
    in phase as well (the fixtures hold no count >= 16 inside a tie)."""
    assert np.abs(res["phase"] - g["phase"]).max(initial=0.0) <= phase_tol
    tie = (res["flags"] & 1) != 0
    bad = res["valid"] != g["valid"]
    assert not bad.any(), f"valid_codons differs from the reference on {bad.sum()} ORFs ({(bad & tie).sum()} of them tie-flagged)"
    assert np.array_equal(res["phase"][tie], g["phase"][tie]), "replayed phase scores must be the reference's bits"
    return tie
