"""Helpers shared by CPU and GPU tests."""

import numpy as np

from oracle import c_oracle, verify

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
    return verify.sub_csr(counts, offsets, idx)


def resolve_like_the_product(res, counts, offsets, thresholds=None):
    """Raw device results may hold ORFs flagged RP_FLAG_BIGTIE (an exact frame tie involving a
    count >= 16: the device stands on x*x where the reference has libm pow()).  The product's
    Python layer finishes those on the host (engine.resolve_big_ties -> rp_tie_replay_host);
    tests that read raw device tensors apply the same step here, so that what is compared with
    the oracle is what a caller of the package gets.  Returns how many ORFs that was."""
    from ribotricer_amd.engine import csr_profiles_of, resolve_big_ties

    return resolve_big_ties(res, csr_profiles_of(np.asarray(counts), np.asarray(offsets)), thresholds)


def check_tie_replay(res, counts, offsets):
    """Tie-flagged ORFs carry the reference's own bits: phase score and valid_codons equal
    oracle/scipy_replay.c -- bit-identical to the reference on every golden vector -- BIT FOR
    BIT, whatever the counts (round 2 licensed ORFs holding a count >= 16 to differ)."""
    resolve_like_the_product(res, counts, offsets)
    tie = (res["flags"] & 1) != 0
    assert np.array_equal(tie, (res["flags"] & 8) != 0), "every tie-flagged ORF (and no other) must be replayed"
    idx = np.nonzero(tie)[0]
    if idx.size == 0:
        return 0
    c, o = _sub_csr(counts, offsets, idx)
    rep = c_oracle.replay_csr(c, o)
    same = (res["valid"][idx] == rep.valid) & (res["phase"][idx] == rep.phase)
    assert same.all(), f"tie replay differs from the reference's bits on {(~same).sum()} tie ORFs"
    return int(idx.size)


def assert_matches_oracle(res, counts, offsets, phase_tol=1e-6, oracle=None, check_flags=True):
    """res: dict of numpy arrays from the HIP path; compares with the C oracle on the same bytes:
    closed form (phase_oracle.c) everywhere, the scipy replay (scipy_replay.c) on frame ties
    (oracle/verify.py holds the bars)."""
    resolve_like_the_product(res, counts, offsets)
    return verify.check_slice(res, counts, offsets, phase_tol=phase_tol, check_flags=check_flags, oracle=oracle)["oracle"]


def assert_matches_fixture(res, g, phase_tol=1e-6):
    """res vs outputs of the reference itself (tests/golden): phase within tolerance everywhere,
    valid_codons identical on EVERY ORF -- the tie-flagged ones included, and those bit-exact
    in phase as well."""
    resolve_like_the_product(res, g["counts"], g["offsets"])
    assert np.abs(res["phase"] - g["phase"]).max(initial=0.0) <= phase_tol
    tie = (res["flags"] & 1) != 0
    bad = res["valid"] != g["valid"]
    assert not bad.any(), f"valid_codons differs from the reference on {bad.sum()} ORFs ({(bad & tie).sum()} of them tie-flagged)"
    assert np.array_equal(res["phase"][tie], g["phase"][tie]), "replayed phase scores must be the reference's bits"
    return tie


def g12_alignments():
    """The merged alignments of fixture G12 (tests/golden/make_golden.py::g12): strand -> Counter, a '.' strand included."""
    import os
    import sys

    from conftest import GOLDEN

    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    from random_index import read_alignments

    return read_alignments(os.path.join(GOLDEN, "g12_alignments.tsv.gz"))


def g12_expected(name: str) -> bytes:
    """The TSV the REFERENCE wrote for G12 under parameter set `name` (random_index.PARAM_SETS)."""
    import gzip
    import os

    from conftest import GOLDEN

    with gzip.open(os.path.join(GOLDEN, f"g12_expected_{name}.tsv.gz"), "rb") as fh:
        return fh.read()


def g12_params(name: str) -> dict:
    import sys

    from conftest import GOLDEN

    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    from random_index import PARAM_SETS

    return dict(PARAM_SETS[name])
