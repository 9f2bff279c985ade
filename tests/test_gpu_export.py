"""End-to-end drop-in on the GPU: export_orf_coverages (same signature as
detect_orfs.py:206-216) against the TSVs the reference wrote for the same index and
alignments; plus the phasescore() mirror on single profiles."""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_export_cpu import load_alignments, read_tsv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["default", "report_all", "strict"])
def test_export_orf_coverages_matches_reference(tmp_path, name):
    from ribotricer_amd.detect_orfs import export_orf_coverages

    params = json.load(open(os.path.join(GOLDEN, "g6_params.json")))["param_sets"][name]
    prefix = str(tmp_path / "out")
    export_orf_coverages(os.path.join(GOLDEN, "g6_index.tsv"), load_alignments(), prefix, **params)
    with open(prefix + "_translating_ORFs.tsv") as fh:
        header = fh.readline().rstrip("\n").split("\t")
        got = [line.rstrip("\n").split("\t") for line in fh]
    eh, expect = read_tsv(f"g6_expected_{name}.tsv")
    assert header == eh
    assert len(got) == len(expect)
    for g, e in zip(got, expect):
        assert g[:3] == e[:3]
        assert abs(float(g[3]) - float(e[3])) <= 1e-6  # BASELINE.json tolerance
        assert g[4:6] == e[4:6]
        assert g[8:] == e[8:]
    # valid_codons may differ only on exact frame ties (flagged by the engine)
    diff = [i for i, (g, e) in enumerate(zip(got, expect)) if g[6:8] != e[6:8]]
    assert len(diff) <= 0.03 * len(got)


def test_phasescore_mirror(g1, g5):
    from ribotricer_amd.statistics import phasescore, phasescore_batch

    for row in g1:
        p, v = phasescore(row["input"])
        assert isinstance(p, np.float64) and isinstance(v, int)
        assert abs(p - row["phase"]) <= 1e-6
    phase, valid, flags = phasescore_batch([r["input"] for r in g5])
    for i, r in enumerate(g5):
        assert abs(phase[i] - r["phase"]) <= 1e-9
        if not flags[i] & 1:
            assert valid[i] == r["valid"]
