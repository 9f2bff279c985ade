"""The GPU-less backend (RIBOTRICER_AMD_BACKEND=cpu; ribotricer_amd/backend.py): export_orf_coverages and phasescore
entirely through the library's *_host entry points -- BASELINE configs[0], "CPU path, plumbing" -- against outputs of
the reference itself.  It is the reference's own float64 operation sequence, so the bar here is BYTES: every column of
the G6 / G10 TSVs (phase score included, where the GPU path is held to 1e-6) and every bit of the G1 / G5 vectors.
Nothing from oracle/ is involved."""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_export_cpu import load_alignments


@pytest.fixture()
def cpu_backend(monkeypatch):
    monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "cpu")
    monkeypatch.setenv("RIBOTRICER_AMD_INDEX_CACHE", "0")


def load_g10_alignments():
    from collections import Counter, defaultdict

    align = defaultdict(Counter)
    with open(os.path.join(GOLDEN, "g10_alignments.tsv")) as fh:
        fh.readline()
        for line in fh:
            strand, chrom, pos, count = line.rstrip("\n").split("\t")
            align[strand][(chrom, int(pos))] = int(count)
    return align


@pytest.mark.parametrize("fixture", ["g6", "g10"])
@pytest.mark.parametrize("name", ["default", "report_all", "strict"])
def test_cpu_backend_writes_the_reference_bytes(tmp_path, cpu_backend, fixture, name):
    from ribotricer_amd.detect_orfs import export_orf_coverages

    params = json.load(open(os.path.join(GOLDEN, "g6_params.json")))["param_sets"][name]
    prefix = str(tmp_path / "out")
    merged = load_alignments() if fixture == "g6" else load_g10_alignments()
    timings = {}
    export_orf_coverages(os.path.join(GOLDEN, "g6_index.tsv"), merged, prefix, timings=timings, **params)
    assert timings["backend"] == "cpu"
    with open(prefix + "_translating_ORFs.tsv", "rb") as fh:
        got = fh.read()
    with open(os.path.join(GOLDEN, f"{fixture}_expected_{name}.tsv"), "rb") as fh:
        assert got == fh.read()


@pytest.mark.parametrize("name", ["default", "report_all", "strict"])
def test_cpu_backend_on_the_corner_index_g12(tmp_path, cpu_backend, name):
    """G12: 1 500 lines from tests/golden/random_index.py -- overlapping / duplicated / abutting / 1-nt exons, shuffled
    lists, a '.' strand with reads of its own, blocks with end < start (alone: a row of length 0 with the raw sum in its
    id, orf.py:103; among others: no positions from it), numbers spelled '+12' / '0012' / ' 12' -- byte for byte what the
    reference wrote."""
    from helpers import g12_alignments, g12_expected, g12_params
    from ribotricer_amd.detect_orfs import export_orf_coverages

    prefix = str(tmp_path / "out")
    export_orf_coverages(os.path.join(GOLDEN, "g12_index.tsv"), g12_alignments(), prefix, **g12_params(name))
    with open(prefix + "_translating_ORFs.tsv", "rb") as fh:
        got = fh.read()
    assert got == g12_expected(name)
    if name == "report_all":
        rows = [r.split(b"\t") for r in got.split(b"\n")[1:-1]]
        empty = [r for r in rows if r[5] == b"0"]  # nothing but end < start blocks
        assert len(empty) >= 20 and all(r[17] == b"[]" and r[2] == b"nontranslating" and int(r[0].rsplit(b"_", 1)[1]) <= 0 for r in empty)


def test_cpu_backend_line_with_end_below_start(tmp_path, cpu_backend):
    """`10-5`: the reference writes tx_10_5_-4 / nontranslating / 0.0 / length 0 / [] (orf.py:100-103, detect_orfs.py:176-187:
    range(10, 6) is empty); beside a real block the bad one only shows in the id's sum.  Expected rows: what the reference
    printed for these three lines (run in the build container)."""
    from collections import Counter, defaultdict

    from ribotricer_amd.detect_orfs import export_orf_coverages

    header = "ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n"
    index = tmp_path / "bad_candidate_orfs.tsv"
    index.write_text(header + "x\tannotated\ttx\tpc\tg\tn\tpc\tchr1\t+\tATG\t10-5\n"
                     + "y\tnovel\tty\tpc\tg\tn\tpc\tchr9\t-\tATG\t40-31\n"  # its (strand, chrom) holds nothing else
                     + "z\tuORF\ttz\tpc\tg\tn\tpc\tchr1\t+\tATG\t30-25,100-105\n")
    align = defaultdict(Counter)
    align["+"][("chr1", 100)] = 4
    align["+"][("chr1", 7)] = 2
    export_orf_coverages(str(index), align, str(tmp_path / "o"), report_all=True)
    rows = [r.split("\t") for r in (tmp_path / "o_translating_ORFs.tsv").read_text().splitlines()[1:]]
    assert rows[0][:9] == ["tx_10_5_-4", "annotated", "nontranslating", "0.0", "0", "0", "0", "0.0", "0.0"] and rows[0][17] == "[]"
    assert rows[1][:9] == ["ty_40_31_-8", "novel", "nontranslating", "0.0", "0", "0", "0", "0.0", "0.0"] and rows[1][17] == "[]"
    assert rows[2][:9] == ["tz_30_105_2", "uORF", "nontranslating", "0.0", "4", "6", "0", "0.0", "2.0"] and rows[2][17] == "[4, 0, 0, 0, 0, 0]"


def test_cpu_backend_takes_columns_as_well(tmp_path, cpu_backend):
    """The package's own hand-over format (alignments.MergedColumns) with rows of several read lengths on one position."""
    from ribotricer_amd.alignments import MergedColumns
    from ribotricer_amd.detect_orfs import export_orf_coverages

    cols = MergedColumns.from_counters(load_alignments())
    half = cols.count // 2  # every position twice: the halves add up again
    doubled = MergedColumns(np.concatenate([cols.strand, cols.strand]), np.concatenate([cols.chrom, cols.chrom]),
                            np.concatenate([cols.pos, cols.pos]), np.concatenate([half, cols.count - half]), cols.chroms)
    prefix = str(tmp_path / "out")
    export_orf_coverages(os.path.join(GOLDEN, "g6_index.tsv"), doubled, prefix, report_all=True)
    with open(prefix + "_translating_ORFs.tsv", "rb") as a, open(os.path.join(GOLDEN, "g6_expected_report_all.tsv"), "rb") as b:
        assert a.read() == b.read()


def test_host_gather_equals_the_python_loop(cpu_backend):
    """rp_gather_profiles_host against the per-nucleotide dict lookups of orf_coverage (detect_orfs.py:134-203 as
    restated in ribotricer_amd.detect_orfs.orf_coverage), both strands, spliced ORFs."""
    from ribotricer_amd import backend
    from ribotricer_amd import detect_orfs as d
    from ribotricer_amd.index import NativeIndex

    path = os.path.join(GOLDEN, "g6_index.tsv")
    merged = load_alignments()
    counts, offsets = backend.gather_profiles_host(NativeIndex.from_file(path), merged, n_threads=3)
    want_counts, want_offsets = d.pack_profiles(d.read_index(path), merged)
    assert np.array_equal(offsets, want_offsets)
    assert np.array_equal(counts, want_counts)


def test_cpu_backend_phasescore_known_answers(cpu_backend, g1, g5, g8f):
    from ribotricer_amd.statistics import phasescore

    for v in g1:
        phase, valid = phasescore(v["input"])
        assert isinstance(phase, np.float64) and isinstance(valid, int)
        want = v["phase"]
        assert (np.isnan(phase) and want is None) or phase == np.float64(want), v
        assert valid == v["valid"], v
    for v in list(g5) + list(g8f):  # float profiles (metagene.py:243-244): the reference's bits
        phase, valid = phasescore(v["input"])
        assert phase == np.float64(v["phase"]) and valid == v["valid"], v


def test_cpu_backend_rejects_counts_beyond_int32(cpu_backend, tmp_path):
    from collections import Counter

    from ribotricer_amd._lib import RibophaseError
    from ribotricer_amd.detect_orfs import export_orf_coverages

    merged = load_alignments()
    strand = next(iter(merged))
    table = Counter(merged[strand])
    key = next(iter(table))
    table[key] = 1 << 31
    merged = dict(merged)
    merged[strand] = table
    with pytest.raises(RibophaseError) as e:
        export_orf_coverages(os.path.join(GOLDEN, "g6_index.tsv"), merged, str(tmp_path / "x"), report_all=True)
    assert e.value.status == -7


def test_backend_selection(monkeypatch):
    from ribotricer_amd import backend

    monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "hip")
    assert backend.selected() == "hip"
    monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "cpu")
    assert backend.selected() == "cpu"
    monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "cuda")
    with pytest.raises(ValueError):
        backend.selected()
    monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "auto")
    import torch

    assert backend.selected() == ("hip" if torch.cuda.device_count() > 0 else "cpu")


def test_hip_backend_never_falls_back(monkeypatch, tmp_path):
    """With the hip backend insisted on and no device around, the export fails loudly -- it does not quietly take the
    host path."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a HIP device is visible")
    from ribotricer_amd._lib import RibophaseError
    from ribotricer_amd.detect_orfs import export_orf_coverages

    monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "hip")
    with pytest.raises(RibophaseError):
        export_orf_coverages(os.path.join(GOLDEN, "g6_index.tsv"), load_alignments(), str(tmp_path / "x"))
    assert not os.path.exists(str(tmp_path / "x") + "_translating_ORFs.tsv")


def test_cpu_backend_whole_pipeline_from_a_bam(tmp_path, cpu_backend, capsys):
    """BASELINE configs[0] in miniature, no GPU: BAM -> columns -> metagene (rp_gather_profiles_host + rp_metagene_host)
    -> offsets -> merge -> WIG -> TSV on a BAM holding the G7 reads.  Every output file equals the reference's, BYTE FOR
    BYTE -- the metagene phase scores and the TSV's phase_score column included (the hip backend is held to 1e-9 /
    1e-6 there)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from bamwriter import write_bam
    from test_host_frontend_cpu import g7_params, load_g7_alignments

    INDEX = os.path.join(GOLDEN, "g6_index.tsv")

    from ribotricer_amd.detect_orfs import detect_orfs

    nested = load_g7_alignments()
    refs = [("chrI", 400000), ("chrII", 400000), ("chrM", 400000)]
    reads = []
    for length in sorted(nested):
        for strand in nested[length]:
            for (chrom, pos), count in nested[length][strand].items():
                for _ in range(count):  # forward protocol: '+' reads start at pos, '-' reads END at pos
                    start0 = pos - 1 if strand == "+" else pos - length
                    reads.append(dict(name="r", chrom=chrom, pos=start0, flag=0 if strand == "+" else 16, mapq=255, cigar=[("M", length)]))
    lead = []  # lengths are met in the order the reference's generator met them (its dict order decides the offsets report)
    for length in [int(k) for k in g7_params()["psite_offsets"]]:
        k = next(i for i, r in enumerate(reads) if r["cigar"][0][1] == length)
        lead.append(reads.pop(k))
    bam = str(tmp_path / "g7.bam")
    write_bam(bam, refs, lead + reads)
    prefix = str(tmp_path / "out" / "g7")
    params = g7_params()
    detect_orfs(bam, INDEX, prefix, "forward", None, None, report_all=True, meta_min_reads=params["meta_min_reads"])
    capsys.readouterr()
    for name in ("psite_offsets.txt", "pos.wig", "neg.wig", "metagene_profiles_5p.tsv", "metagene_profiles_3p.tsv", "translating_ORFs.tsv"):
        with open(f"{prefix}_{name}", "rb") as a, open(os.path.join(GOLDEN, f"g7_expected_{name}"), "rb") as b:
            assert a.read() == b.read(), name


@pytest.mark.parametrize("backend", ["cpu", "auto"])
def test_cpu_backend_runs_where_torch_cannot_be_imported(tmp_path, backend):
    """A GPU-less host WITHOUT torch: export_orf_coverages, phasescore and the offsets step through the cpu backend, in a
    fresh interpreter where `import torch` raises ImportError (`auto` then has no hip backend to offer and says so once).
    Same bytes as the reference's G6 file.  (libribophase.so itself still needs the ROCm runtime libraries on the loader
    path -- one library carries both backends -- but no device and no driver.)"""
    import subprocess
    import sys

    code = f"""
import sys
sys.modules['torch'] = None            # any `import torch` now raises ImportError
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
import json, os
from test_host_export_cpu import load_alignments
from ribotricer_amd.detect_orfs import export_orf_coverages
from ribotricer_amd.statistics import phasescore
timings = {{}}
export_orf_coverages({os.path.join(GOLDEN, "g6_index.tsv")!r}, load_alignments(), {str(tmp_path / "out")!r}, timings=timings)
assert timings['backend'] == 'cpu'
assert phasescore([7, 0, 0, 7, 2, 0, 1, 0, 0, 5, 0, 0, 9, 0, 0, 8, 0, 0, 7, 0, 0, 5, 0, 0, 8, 0, 0, 6, 0, 0]) == (0.996462947685208, 10)
assert 'torch' not in [m for m, v in sys.modules.items() if v is not None and m == 'torch']
print('ok')
"""
    env = dict(os.environ, RIBOTRICER_AMD_BACKEND=backend, RIBOTRICER_AMD_INDEX_CACHE="0")
    env.pop("HIP_VISIBLE_DEVICES", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
    if backend == "auto":
        assert "using the CPU backend" in out.stderr
    with open(str(tmp_path / "out") + "_translating_ORFs.tsv", "rb") as got, open(os.path.join(GOLDEN, "g6_expected_default.tsv"), "rb") as want:
        assert got.read() == want.read()


@pytest.mark.parametrize("seed", range(8))
def test_native_export_equals_the_python_restatement_on_corner_indexes(tmp_path, cpu_backend, seed):
    """Two roads through the package that share only the scorer: the native road (rp_index_parse_host ->
    rp_gather_profiles_host -> rp_phase_score_csr_host -> rp_format_rows_host) against the readable one
    (detect_orfs.parse_index_line -> orf_coverage's per-position dict lookups -> pack_profiles -> format_rows: Python
    restatements of orf.py:122-182 and detect_orfs.py:134-203,301-324) on fresh corner-case indexes -- blocks with
    end < start, a '.' strand, nested / overlapping / duplicated exons, dressed numbers, CRLF, no final newline.  EXPORTS are
    compared, not parser arrays: the round-5 crash on `10-5` lived between the two."""
    import sys

    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    from random_index import random_index

    from ribotricer_amd import _lib
    from ribotricer_amd import detect_orfs as d

    shape = [dict(), dict(malformed=0.2), dict(dressed=0.25), dict(crlf=True), dict(final_newline=False, malformed=0.1),
             dict(malformed=0.3, dressed=0.3), dict(malformed=1.0), dict(dressed=1.0, crlf=True)][seed]
    text, merged = random_index(400, 900 + seed, **shape)
    index = tmp_path / "c_candidate_orfs.tsv"
    with open(index, "w", newline="") as fh:
        fh.write(text)
    records = d.read_index(str(index))
    counts, offsets = d.pack_profiles(records, merged)
    res = _lib.phase_score_csr_host(counts, offsets, d.make_filter(), n_threads=2)
    for report_all in (False, True):
        d.export_orf_coverages(str(index), merged, str(tmp_path / "n"), report_all=report_all)
        want = "\t".join(d.COLUMNS) + "\n" + "".join(d.format_rows(records, counts, offsets, res, report_all))
        assert open(str(tmp_path / "n") + "_translating_ORFs.tsv", newline="").read() == want


def test_classic_mac_line_ends_are_lines(tmp_path, cpu_backend):
    """The reference opens the index in text mode: '\\r' alone ends a line (universal newlines).  A '\\r'-ended file gives the
    same rows as its '\\n'-ended twin (it used to parse as ONE header line: an empty result without an error)."""
    from collections import Counter, defaultdict

    from ribotricer_amd.detect_orfs import export_orf_coverages

    header = "ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n"
    line = "x\tannotated\tt1\tpc\tg\tn\tpc\tchr1\t+\tATG\t10-39\n"
    align = defaultdict(Counter)
    align["+"][("chr1", 10)] = 5
    align["+"][("chr1", 13)] = 4
    outs = {}
    for tag, text in (("lf", header + line + line), ("cr", (header + line + line).replace("\n", "\r")), ("crlf", (header + line + line).replace("\n", "\r\n")),
                      ("cr_no_final", (header + line + line).replace("\n", "\r")[:-1])):
        with open(tmp_path / f"{tag}.tsv", "w", newline="") as fh:
            fh.write(text)
        export_orf_coverages(str(tmp_path / f"{tag}.tsv"), align, str(tmp_path / tag), report_all=True)
        outs[tag] = (tmp_path / f"{tag}_translating_ORFs.tsv").read_bytes()
    assert outs["lf"].count(b"\n") == 3 and outs["cr"] == outs["lf"] and outs["crlf"] == outs["lf"] and outs["cr_no_final"] == outs["lf"]
