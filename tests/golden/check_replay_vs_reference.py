"""Container-only check (imports /root/reference): oracle/scipy_replay.c AND the product's host tie
resolver (rp_tie_replay_host / rp_tie_replay_f64_host of libribophase.so, no GPU involved) against
the reference's own phasescore on a fresh batch of tie-rich profiles -- sparse ones, ones scaled up
to counts of 16..1000, and float-valued ones -- phase score bit for bit, valid_codons exact.
usage: python tests/golden/check_replay_vs_reference.py [n] [seed]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from ribotricer.statistics import phasescore  # noqa: E402

from oracle import c_oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
lens = rng.integers(0, 400, n)
lam = rng.choice([0.003, 0.01, 0.03, 0.1, 0.3, 1.0, 5.0, 40.0], n)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
counts = rng.poisson(np.repeat(lam, lens)).astype(np.int32)
rep = c_oracle.replay_csr(counts, offsets)
cf = c_oracle.phase_score_csr(counts, offsets)
bad_phase = bad_valid = ties = tie_disagree_closed_form = 0
for i in range(n):
    p, v = phasescore(counts[offsets[i] : offsets[i + 1]].tolist())
    bad_phase += float(p) != rep.phase[i]
    bad_valid += int(v) != rep.valid[i]
    if cf.flags[i] & 1:
        ties += 1
        tie_disagree_closed_form += int(v) != cf.valid[i]
# the product's host resolver on the same profiles, on a scaled-up copy (counts 16..1000 inside the ties) and on floats
from ribotricer_amd import _lib  # noqa: E402

host_bad = 0
ph, va = _lib.tie_replay_host(counts, offsets)
host_bad += int((ph != rep.phase).sum() + (va != rep.valid).sum())
scale = np.repeat(rng.integers(16, 1001, n), lens)
big = (counts.astype(np.int64) * np.where(np.repeat(lam, lens) < 0.2, scale, 1)).astype(np.int32)
ph, va = _lib.tie_replay_host(big, offsets)
fl = big.astype(np.float64) * np.repeat(rng.choice([0.5, 0.125, 1.75, 3.3], n), lens)
phf, vaf = _lib.tie_replay_host(fl, offsets)
for i in range(n):
    p, v = phasescore(big[offsets[i] : offsets[i + 1]].tolist())
    host_bad += (float(p) != ph[i]) + (int(v) != va[i])
    p, v = phasescore(fl[offsets[i] : offsets[i + 1]].tolist())
    host_bad += (float(p) != phf[i]) + (int(v) != vaf[i])
print(f"host resolver (rp_tie_replay_host, libm pow): {host_bad} mismatches over {3 * n} profiles (plain, counts x16..1000, float)")
bad_phase += host_bad
print(f"{n} profiles, {ties} exact frame ties ({tie_disagree_closed_form} of them decided differently by the closed form's "
      f"earlier-frame rule): replay phase mismatches {bad_phase}, valid_codons mismatches {bad_valid}")
sys.exit(1 if bad_phase or bad_valid else 0)
