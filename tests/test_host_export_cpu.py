"""Host side of the drop-in (no GPU): index parsing, profile gathering and TSV row
formatting, checked against files produced by the reference's own
export_orf_coverages (tests/golden/g6_*).  Scores fed to the formatter come from the
C oracle here; the GPU twin (test_gpu_export.py) runs the real path."""

import json
import os
from collections import Counter, defaultdict

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import reference_status
from oracle import c_oracle

# importing the host mirror must not need a GPU (it only needs the built library)
from ribotricer_amd import detect_orfs as d


def load_alignments():
    align = defaultdict(Counter)
    with open(os.path.join(GOLDEN, "g6_alignments.tsv")) as fh:
        fh.readline()
        for line in fh:
            strand, chrom, pos, count = line.rstrip("\n").split("\t")
            align[strand][(chrom, int(pos))] = int(count)
    return align


def read_tsv(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        header = fh.readline().rstrip("\n").split("\t")
        rows = [line.rstrip("\n").split("\t") for line in fh]
    return header, rows


@pytest.fixture(scope="module")
def packed():
    records = d.read_index(os.path.join(GOLDEN, "g6_index.tsv"))
    counts, offsets = d.pack_profiles(records, load_alignments())
    return records, counts, offsets


def test_index_records_follow_orf_from_string(packed):
    records, _, _ = packed
    header, rows = read_tsv("g6_expected_report_all.tsv")
    assert header == d.COLUMNS
    assert len(rows) == len(records) == 220
    for rec, row in zip(records, rows):
        assert rec.oid == row[0] and rec.category == row[1]  # orf.py:103 recomputed ORF_ID
        assert [rec.tid, rec.ttype, rec.gid, rec.gname, rec.gtype, rec.chrom, rec.strand] == row[9:16]
        assert str(rec.start_codon) == row[16]  # 'None' when the index field is < 3 chars
    assert any(r.start_codon is None for r in records)
    assert all(r.intervals == tuple(sorted(r.intervals)) for r in records)


def test_wrong_column_count_exits_like_the_reference():
    with pytest.raises(SystemExit) as e:
        d.parse_index_line("a\tb\tc\n")
    assert "unexpected number of columns" in str(e.value)  # orf.py:145-152


def test_packed_profiles_equal_reference_profile_column(packed):
    records, counts, offsets = packed
    _, rows = read_tsv("g6_expected_report_all.tsv")
    for i, row in enumerate(rows):
        expect = json.loads(row[17])  # the reference prints the python list it scored
        assert counts[offsets[i] : offsets[i + 1]].tolist() == expect
        assert int(row[5]) == len(expect)
    minus = [i for i, r in enumerate(records) if r.strand == "-"]
    assert minus, "fixture must exercise strand reversal (detect_orfs.py:201-202)"


@pytest.mark.parametrize("name", ["default", "report_all", "strict"])
def test_formatted_rows_match_reference_tsv(packed, name):
    records, counts, offsets = packed
    params = json.load(open(os.path.join(GOLDEN, "g6_params.json")))["param_sets"][name]
    o = c_oracle.phase_score_csr(counts, offsets)
    lengths = np.diff(offsets)
    status = reference_status(
        o.phase, o.valid, o.read_count, o.min_codon_cov, lengths,
        cutoff=params.get("phase_score_cutoff", 0.428571428571), min_valid=params.get("min_valid_codons", 5),
        min_reads=params.get("min_reads_per_codon", 0), min_ratio=params.get("min_valid_codons_ratio", 0),
        min_density=params.get("min_density_over_orf", 0.0),
    )
    res = dict(phase=o.phase, valid=o.valid, read_count=o.read_count, min_codon_cov=o.min_codon_cov, flags=o.flags, status=status)
    got = [r.rstrip("\n").split("\t") for r in d.format_rows(records, counts, offsets, res, params.get("report_all", False))]
    _, expect = read_tsv(f"g6_expected_{name}.tsv")
    tie = {rec.oid for rec, f in zip(records, o.flags) if f & 1}
    assert len(got) == len(expect)
    exact_phase = 0
    for g, e in zip(got, expect):
        assert g[:3] == e[:3], (g[:3], e[:3])  # ORF_ID, ORF_type, status
        assert abs(float(g[3]) - float(e[3])) <= 1e-12
        exact_phase += g[3] == e[3]
        assert g[4:6] == e[4:6]  # read_count, length
        if g[0] not in tie:
            assert g[6:8] == e[6:8]  # valid_codons, valid_codons_ratio (text identical)
        assert g[8:] == e[8:]  # read_density ... profile, byte for byte
    # (the closed form and scipy's FFT route differ in the last ulps, so only some phase
    #  scores print identically: a float64 repr is 17 digits)
    assert 0 <= exact_phase <= len(got)


def test_dense_coverage_and_interval_table_reproduce_the_packer(packed):
    """Host half of the device gather (gather.py): dense coverage + interval table, replayed
    with numpy here, must give the same bytes as the per-nucleotide packer."""
    from ribotricer_amd.gather import build_dense_coverage, build_interval_table

    records, counts, offsets = packed
    coverage, base = build_dense_coverage(load_alignments(), records)
    table = build_interval_table(records, base)
    assert np.array_equal(table.offsets, offsets)
    assert table.orf_iv[-1] == sum(len(r.intervals) for r in records)
    for i in range(len(records)):
        parts = [coverage[s : s + n] for s, n in zip(table.iv_start[table.orf_iv[i] : table.orf_iv[i + 1]],
                                                     table.iv_len[table.orf_iv[i] : table.orf_iv[i + 1]])]
        prof = np.concatenate(parts)
        if table.reverse[i]:
            prof = prof[::-1]
        assert np.array_equal(prof, counts[offsets[i] : offsets[i + 1]]), records[i].oid
