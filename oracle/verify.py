"""The checker: results of the HIP path against the CPU oracle on the same bytes -- TEST
INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), and the `verify` block bench.py runs AFTER its
timed region).  Nothing here is measured or shipped, and nothing here imports the product.

Bars (BASELINE.json north_star): integer outputs bit-exact; phase score within 1e-6 of the float64
closed form (oracle/phase_oracle.c); on exact frame ties phase AND valid_codons
equal to the replay of the reference's own float64 arithmetic (oracle/scipy_replay.c, itself
bit-identical to the reference on every golden vector) BIT FOR BIT -- no licence for large counts
any more: the product finishes those on the host (RP_FLAG_BIGTIE, rp_tie_replay_host).
"""

from __future__ import annotations

import numpy as np

from . import c_oracle

FLAG_TIE, FLAG_REPLAY = 0x01, 0x08


def sub_csr(counts, offsets, idx):
    """CSR batch of the profiles ``idx`` of a CSR batch."""
    counts = np.asarray(counts)
    offsets = np.asarray(offsets)
    lens = (offsets[1:] - offsets[:-1])[idx]
    off = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    parts = [counts[offsets[i] : offsets[i + 1]] for i in idx]
    return (np.concatenate(parts) if parts and off[-1] else np.zeros(0, np.int32)), off


def check_slice(res: dict, counts, offsets, phase_tol: float = 1e-6, check_flags: bool = True, n_threads: int = 4, oracle=None) -> dict:
    """``res``: dict of host numpy arrays (phase, valid, read_count, min_codon_cov, flags) for the
    ORFs of the CSR batch ``(counts, offsets)``.  Raises AssertionError on the first broken bar;
    returns ``{"orfs", "max_abs_dphase", "ties", "oracle"}``."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    o = oracle if oracle is not None else c_oracle.phase_score_csr(counts, offsets, n_threads=n_threads)
    for key, want in (("read_count", o.read_count), ("min_codon_cov", o.min_codon_cov)):
        if not np.array_equal(res[key], want):  # (say WHERE: a once-in-many-runs failure must leave something to go on)
            bad = np.flatnonzero(np.asarray(res[key]) != want)
            i = int(bad[0])
            raise AssertionError(f"{key} must be bit-exact: {bad.size} of {want.size} ORFs differ, first at {i} (length {int(offsets[i + 1] - offsets[i])}): "
                                 f"got {int(res[key][i])}, oracle {int(want[i])}; last at {int(bad[-1])}")
    dphase = np.abs(res["phase"] - o.phase)
    worst = float(dphase.max(initial=0.0))
    assert worst <= phase_tol, f"phase differs by {worst}"
    tie_gpu = (res["flags"] & FLAG_TIE) != 0
    tie_cpu = (o.flags & 1) != 0
    bad = (res["valid"] != o.valid) & ~(tie_gpu | tie_cpu)
    assert not bad.any(), f"valid_codons differs on {bad.sum()} non-tie ORFs, first {np.nonzero(bad)[0][:5]}"
    if check_flags:
        assert np.array_equal(tie_gpu, tie_cpu), "tie flags differ from the oracle's"
    assert np.array_equal(tie_gpu, (res["flags"] & FLAG_REPLAY) != 0), "every tie-flagged ORF (and no other) must be replayed"
    idx = np.nonzero(tie_gpu)[0]
    if idx.size:
        c, off = sub_csr(counts, offsets, idx)
        rep = c_oracle.replay_csr(c, off)
        same = (res["valid"][idx] == rep.valid) & (res["phase"][idx] == rep.phase)
        assert same.all(), f"tie replay differs from the reference's bits on {(~same).sum()} of {idx.size} tie ORFs, first {idx[~same][:5]}"
    return {"orfs": int(offsets.size - 1), "max_abs_dphase": worst, "ties": int(idx.size), "oracle": o}
