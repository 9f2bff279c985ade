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
    # valid_codons / valid_codons_ratio: identical to the reference's on every row, the exact frame
    # ties (RP_FLAG_TIE) included -- those are decided by the on-device replay of
    # the reference's float64 arithmetic.  Census: (rows, flagged rows, flagged rows that differ)
    import torch

    from ribotricer_amd import detect_orfs as d
    from ribotricer_amd.engine import get_engine

    records = d.read_index(os.path.join(GOLDEN, "g6_index.tsv"))
    counts, offsets = d.pack_profiles(records, load_alignments())
    flags = get_engine("cuda:0").score(counts, offsets).flags
    torch.cuda.synchronize()
    tie = {r.oid: bool(f & 1) for r, f in zip(records, flags.cpu().numpy())}
    diff = [g[0] for g, e in zip(got, expect) if g[6:8] != e[6:8]]
    assert all(tie[oid] for oid in diff), "valid_codons differs from the reference on an unflagged ORF"
    census = (len(got), sum(tie[g[0]] for g in got), len(diff))
    assert census == {"default": (63, 0, 0), "report_all": (220, 1, 0), "strict": (220, 1, 0)}[name], census


def load_g10_alignments():
    from collections import Counter, defaultdict

    align = defaultdict(Counter)
    with open(os.path.join(GOLDEN, "g10_alignments.tsv")) as fh:
        fh.readline()
        for line in fh:
            strand, chrom, pos, count = line.rstrip("\n").split("\t")
            align[strand][(chrom, int(pos))] = int(count)
    return align


@pytest.mark.parametrize("devices", [None, [0, 0]])
@pytest.mark.parametrize("name", ["default", "report_all", "strict"])
def test_export_with_counts_beyond_fp32_matches_reference(tmp_path, name, devices):
    """G10 end to end: the g6 index with counts 2^24 .. 2^30 piled on 16 ORFs (32 positions).  Round 3 aborted this
    export (RP_ERR_COUNTS); the reference has no limit (detect_orfs.py:176-187, 278-280).  Every column of every row
    equals the reference's TSV -- phase within 1e-6, the rest as text (read_count beyond 2^32, valid_codons on the
    saturated ORFs, the status column, the printed profiles)."""
    from ribotricer_amd.detect_orfs import export_orf_coverages

    params = json.load(open(os.path.join(GOLDEN, "g6_params.json")))["param_sets"][name]
    prefix = str(tmp_path / "out")
    export_orf_coverages(os.path.join(GOLDEN, "g6_index.tsv"), load_g10_alignments(), prefix, devices=devices, **params)
    with open(prefix + "_translating_ORFs.tsv") as fh:
        header = fh.readline().rstrip("\n").split("\t")
        got = [line.rstrip("\n").split("\t") for line in fh]
    eh, expect = read_tsv(f"g10_expected_{name}.tsv")
    assert header == eh and len(got) == len(expect)
    saturated = 0
    for g, e in zip(got, expect):
        assert g[:3] == e[:3] and g[4:] == e[4:], (g[0], g[:9], e[:9])
        assert abs(float(g[3]) - float(e[3])) <= 1e-6
        saturated += int(g[4]) > 16777215
    assert saturated >= (10 if name != "default" else 1)


@pytest.mark.parametrize("report_all", [False, True])
def test_compact_and_dense_coverage_write_the_same_file(tmp_path, report_all):
    """The export builds a COMPACT coverage by default (only the blocks under an exon keep a slot: gather.CoverageMap);
    RIBOTRICER_AMD_COMPACT_COVERAGE=0 keeps the dense one.  Same bytes either way, one GPU and sharded, counts beyond
    2^24 - 1 included (their positions are named in the coordinates of whichever coverage is in use)."""
    from ribotricer_amd import detect_orfs as d

    index = os.path.join(GOLDEN, "g6_index.tsv")
    outs = {}
    for mode in ("1", "0"):
        os.environ["RIBOTRICER_AMD_COMPACT_COVERAGE"] = mode
        d.forget_indexes()
        try:
            for tag, align, devices in (("plain", load_alignments(), None), ("big", load_g10_alignments(), None), ("sharded", load_g10_alignments(), [0, 0, 0])):
                prefix = str(tmp_path / f"{tag}{mode}")
                d.export_orf_coverages(index, align, prefix, report_all=report_all, devices=devices)
                outs[(tag, mode)] = open(prefix + "_translating_ORFs.tsv", "rb").read()
            cached = next(reversed(d._INDEX_CACHE.values()))
            (table, plan, extras), = cached.__dict__["_layout_cache"].values()
            assert ("coverage_map" in extras) == (mode == "1")
            if mode == "1":
                assert extras["coverage_map"].compact_len < extras["coverage_map"].dense_len
        finally:
            del os.environ["RIBOTRICER_AMD_COMPACT_COVERAGE"]
    d.forget_indexes()
    for tag in ("plain", "big", "sharded"):
        assert outs[(tag, "1")] == outs[(tag, "0")], tag
    # one GPU vs three slices: another tiling may move an unflagged phase by <= 2e-7; every other column
    # -- the integer results, the status, the saturated ORFs' float64 fix-ups, the printed profiles -- is the same text
    one, three = (outs[(tag, "1")].decode().splitlines() for tag in ("big", "sharded"))
    assert len(one) == len(three) and one[0] == three[0]
    for a, b in zip(one[1:], three[1:]):
        a, b = a.split("\t"), b.split("\t")
        assert a[:3] == b[:3] and a[4:] == b[4:] and abs(float(a[3]) - float(b[3])) <= 1e-6


def test_phasescore_mirror(g1, g5):
    from ribotricer_amd.statistics import phasescore, phasescore_batch

    for row in g1:
        p, v = phasescore(row["input"])
        assert isinstance(p, np.float64) and isinstance(v, int)
        assert abs(p - row["phase"]) <= 1e-6
        assert v == row["valid"], row["input"]  # ties included: replayed with the reference's arithmetic
    phase, valid, flags = phasescore_batch([r["input"] for r in g5])
    for i, r in enumerate(g5):
        assert abs(phase[i] - r["phase"]) <= 1e-9
        assert valid[i] == r["valid"]


def test_device_gather_equals_host_packer():
    """rp_gather_profiles_dev vs the per-nucleotide host packer (the reference's orf_coverage
    semantics, pinned by test_host_export_cpu) on the G6 fixture."""
    import torch

    from ribotricer_amd import detect_orfs as d

    records = d.read_index(os.path.join(GOLDEN, "g6_index.tsv"))
    align = load_alignments()
    counts, offsets = d.pack_profiles(records, align)
    dc, do = d.gather_profiles(records, align)
    torch.cuda.synchronize()
    assert np.array_equal(do.cpu().numpy(), offsets)
    assert np.array_equal(dc.cpu().numpy(), counts)


def test_device_gather_random_intervals():
    import torch

    from ribotricer_amd.gather import IntervalTable, gather_profiles_device

    rng = np.random.default_rng(8)
    cov = rng.poisson(0.6, size=200000).astype(np.int32)
    n = 3000
    n_iv = rng.integers(1, 6, size=n)
    n_iv[[5, 6, 7, 64, 2999]] = [150, 64, 65, 130, 70]  # around and beyond the 64 descriptors a wave holds
    orf_iv = np.concatenate([[0], np.cumsum(n_iv)]).astype(np.int64)
    iv_len = rng.integers(1, 400, size=int(orf_iv[-1])).astype(np.int32)
    iv_start = rng.integers(-50, cov.size - 100, size=int(orf_iv[-1])).astype(np.int64)  # some hang off both ends
    reverse = rng.integers(0, 2, size=n).astype(np.uint8)
    lengths = np.add.reduceat(iv_len.astype(np.int64), orf_iv[:-1])
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    dc, _ = gather_profiles_device(cov, IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets))
    torch.cuda.synchronize()
    got = dc.cpu().numpy()
    for i in range(n):
        parts = []
        for k in range(orf_iv[i], orf_iv[i + 1]):
            idx = iv_start[k] + np.arange(iv_len[k])
            ok = (idx >= 0) & (idx < cov.size)
            parts.append(np.where(ok, cov[np.clip(idx, 0, cov.size - 1)], 0))
        want = np.concatenate(parts)
        if reverse[i]:
            want = want[::-1]
        assert np.array_equal(got[offsets[i] : offsets[i + 1]], want), i


def test_export_edge_indexes(tmp_path):
    """Header-only index, alignments on a strand / chromosome no ORF uses, an ORF with no
    reads at all: same files as the reference's loop would write."""
    from collections import Counter, defaultdict

    from ribotricer_amd.detect_orfs import COLUMNS, export_orf_coverages

    header = "\t".join(["ORF_ID", "ORF_type", "transcript_id", "transcript_type", "gene_id", "gene_name", "gene_type",
                        "chrom", "strand", "start_codon", "coordinate"]) + "\n"
    empty = tmp_path / "empty_candidate_orfs.tsv"
    empty.write_text(header)
    align = defaultdict(Counter)
    align["+"][("chr1", 100)] = 3
    export_orf_coverages(str(empty), align, str(tmp_path / "e"))
    assert (tmp_path / "e_translating_ORFs.tsv").read_text() == "\t".join(COLUMNS) + "\n"

    one = tmp_path / "one_candidate_orfs.tsv"
    one.write_text(header + "x\tannotated\ttx\tpc\tg\tn\tpc\tchr2\t-\tATG\t10-18,30-38\n")
    align["-"][("chr2", 12)] = 5
    align["-"][("chr2", 500)] = 7     # outside every ORF
    align["+"][("chr2", 12)] = 9      # other strand
    export_orf_coverages(str(one), align, str(tmp_path / "o"), report_all=True)
    rows = (tmp_path / "o_translating_ORFs.tsv").read_text().splitlines()
    assert len(rows) == 2
    f = rows[1].split("\t")
    assert f[0] == "tx_10_38_18" and f[2] == "nontranslating" and f[4:6] == ["5", "18"]
    # '-' strand: the profile is reversed (detect_orfs.py:201-202); position 12 is the 3rd of 10-18
    assert f[17] == str([0] * 9 + [0, 0, 0, 0, 0, 0, 5, 0, 0])


def test_export_line_with_end_below_start(tmp_path):
    """`10-5` on the HIP path, one GPU and three slices: the rows the reference prints (orf.py:100-103: the id keeps the raw
    sum; detect_orfs.py:176-187: no positions from such a block) -- tests/test_host_backend_cpu.py holds the same rows."""
    from collections import Counter, defaultdict

    from ribotricer_amd import detect_orfs as d

    header = "ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n"
    index = tmp_path / "bad_candidate_orfs.tsv"
    index.write_text(header + "x\tannotated\ttx\tpc\tg\tn\tpc\tchr1\t+\tATG\t10-5\n"
                     + "y\tnovel\tty\tpc\tg\tn\tpc\tchr9\t-\tATG\t40-31\n"
                     + "z\tuORF\ttz\tpc\tg\tn\tpc\tchr1\t+\tATG\t30-25,100-105\n"
                     + "w\tuORF\ttw\tpc\tg\tn\tpc\tchr1\t+\tATG\t" + ",".join(f"{200 + 9 * k}-{205 + 9 * k}" for k in range(12)) + ",9-3\n")
    align = defaultdict(Counter)
    align["+"][("chr1", 100)] = 4
    align["+"][("chr1", 7)] = 2
    for k in range(12):
        align["+"][("chr1", 200 + 9 * k)] = 3 + k
        align["+"][("chr1", 203 + 9 * k)] = 2
    for tag, devices, report_all in (("one", None, True), ("three", [0, 0, 0], True), ("one_d", None, False), ("three_d", [0, 0, 0], False)):
        d.forget_indexes()
        d.export_orf_coverages(str(index), align, str(tmp_path / tag), report_all=report_all, devices=devices)
        rows = [r.split("\t") for r in (tmp_path / f"{tag}_translating_ORFs.tsv").read_text().splitlines()[1:]]
        if report_all:
            assert rows[0][:9] == ["tx_10_5_-4", "annotated", "nontranslating", "0.0", "0", "0", "0", "0.0", "0.0"] and rows[0][17] == "[]"
            assert rows[1][:9] == ["ty_40_31_-8", "novel", "nontranslating", "0.0", "0", "0", "0", "0.0", "0.0"] and rows[1][17] == "[]"
            assert rows[2][:9] == ["tz_30_105_2", "uORF", "nontranslating", "0.0", "4", "6", "0", "0.0", "2.0"] and rows[2][17] == "[4, 0, 0, 0, 0, 0]"
        assert len(rows) == (4 if report_all else 1)
        w = rows[-1]  # sorted by start the bad block comes first: id tw_9_304_{72 - 5}; a frame tie (valid 23, not 24) as the reference decides it
        assert w[:3] == ["tw_9_304_67", "uORF", "translating"] and w[4:9] == ["126", "72", "23", "0.9583333333333334", "5.25"]
        assert abs(float(w[3]) - 1.0000000000000007) <= 1e-6
    d.forget_indexes()


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
@pytest.mark.parametrize("name", ["default", "report_all", "strict"])
def test_export_of_the_corner_index_g12_matches_reference(tmp_path, name, devices):
    """G12 on the HIP path (one GPU; three slices of it): 1 500 lines from tests/golden/random_index.py -- overlapping /
    nested / duplicated / abutting / 1-nt exons, shuffled lists, a '.' strand with reads of its own, blocks with end < start,
    dressed numbers -- against the three TSVs the REFERENCE wrote: every column the same text, the phase score within 1e-6."""
    from helpers import g12_alignments, g12_expected, g12_params
    from ribotricer_amd import detect_orfs as d

    d.forget_indexes()
    prefix = str(tmp_path / "out")
    timings = {}
    d.export_orf_coverages(os.path.join(GOLDEN, "g12_index.tsv"), g12_alignments(), prefix, devices=devices, timings=timings, **g12_params(name))
    assert timings["backend"] == "hip"
    same_rows(g12_expected(name).decode(), open(prefix + "_translating_ORFs.tsv").read())
    d.forget_indexes()


def same_rows(a_text: str, b_text: str, tol: float = 1e-6) -> None:
    a_rows, b_rows = a_text.splitlines(), b_text.splitlines()
    assert len(a_rows) == len(b_rows) and a_rows[0] == b_rows[0]
    for a, b in zip(a_rows[1:], b_rows[1:]):
        a, b = a.split("\t"), b.split("\t")
        assert a[:3] == b[:3] and a[4:] == b[4:] and abs(float(a[3]) - float(b[3])) <= tol, (a[:9], b[:9])


def test_export_sharded_over_two_slices_equals_single(tmp_path):
    """configs[3] through the product entry point: export_orf_coverages(devices=[0, 0]) cuts the
    index into two nt-balanced slices (two streams of the one GPU here) and writes the same file."""
    from ribotricer_amd.detect_orfs import export_orf_coverages

    index = os.path.join(GOLDEN, "g6_index.tsv")
    export_orf_coverages(index, load_alignments(), str(tmp_path / "one"), report_all=True)
    export_orf_coverages(index, load_alignments(), str(tmp_path / "two"), report_all=True, devices=[0, 0])
    os.environ["RIBOTRICER_AMD_DEVICES"] = "0,0,0"
    try:
        export_orf_coverages(index, load_alignments(), str(tmp_path / "env"), report_all=True)
    finally:
        del os.environ["RIBOTRICER_AMD_DEVICES"]
    one = open(str(tmp_path / "one_translating_ORFs.tsv")).read()
    # (every column the same text except the phase score: a slice is tiled on its own, which may move an unflagged phase
    # by <= 2e-7 -- DESIGN.md section 5; the small one-GPU batch even takes the wave kernel)
    same_rows(one, open(str(tmp_path / "two_translating_ORFs.tsv")).read())
    same_rows(one, open(str(tmp_path / "env_translating_ORFs.tsv")).read())
    # default mode: the fused gather + score, sharded the same way (engine.score_coverage_sharded)
    export_orf_coverages(index, load_alignments(), str(tmp_path / "d1"))
    export_orf_coverages(index, load_alignments(), str(tmp_path / "d3"), devices=[0, 0, 0])
    d1 = [line.split("\t") for line in open(str(tmp_path / "d1_translating_ORFs.tsv"))]
    d3 = [line.split("\t") for line in open(str(tmp_path / "d3_translating_ORFs.tsv"))]
    assert len(d1) == len(d3) == 64  # header + the 63 translating ORFs of the fixture
    assert d1[0] == d3[0]
    for a, b in zip(d1[1:], d3[1:]):  # the tile path: another slicing moves the fp32 partial sums by <= 2e-7
        assert a[:3] == b[:3] and a[4:] == b[4:] and abs(float(a[3]) - float(b[3])) <= 1e-6
    # a second sample over the same devices: the shards (windows, gather plans, tile plans per slice) hang on the cached index
    from ribotricer_amd import detect_orfs as d

    export_orf_coverages(index, load_alignments(), str(tmp_path / "d3b"), devices=[0, 0, 0])
    assert open(str(tmp_path / "d3b_translating_ORFs.tsv")).read() == open(str(tmp_path / "d3_translating_ORFs.tsv")).read()
    cached = next(reversed(d._INDEX_CACHE.values()))
    (shards,) = cached.__dict__["_shard_cache"].values()  # (shards.IndexShards: every slice builds its own map, plans, coverage buffer)
    assert len(shards.devices) == 3 and shards.plans_built == 3  # three samples over these devices, one build per slice
    # the round-3/4 flow (whole coverage on devices[0], windows shipped out) is still there behind a switch, same bytes
    os.environ["RIBOTRICER_AMD_SHARD_ON_FIRST_DEVICE"] = "1"
    try:
        export_orf_coverages(index, load_alignments(), str(tmp_path / "old3"), devices=[0, 0, 0])
        export_orf_coverages(index, load_alignments(), str(tmp_path / "old2"), report_all=True, devices=[0, 0])
    finally:
        del os.environ["RIBOTRICER_AMD_SHARD_ON_FIRST_DEVICE"]
    same_rows(open(str(tmp_path / "old3_translating_ORFs.tsv")).read(), open(str(tmp_path / "d3_translating_ORFs.tsv")).read())
    same_rows(open(str(tmp_path / "old2_translating_ORFs.tsv")).read(), one)


def test_index_is_parsed_once_per_file_and_reparsed_when_it_changes(tmp_path):
    """One index, many samples: the parsed index, its interval table and its gather plan are kept across
    export_orf_coverages calls while the file is unchanged; a rewritten file is read again; results are the same
    either way."""
    import shutil

    from ribotricer_amd import detect_orfs as d

    params = json.load(open(os.path.join(GOLDEN, "g6_params.json")))["param_sets"]["default"]
    index_path = str(tmp_path / "index.tsv")
    shutil.copy(os.path.join(GOLDEN, "g6_index.tsv"), index_path)
    d.forget_indexes()
    outs = []
    for k in range(2):
        prefix = str(tmp_path / f"out{k}")
        d.export_orf_coverages(index_path, load_alignments(), prefix, **params)
        outs.append(open(prefix + "_translating_ORFs.tsv", "rb").read())
    assert outs[0] == outs[1]
    assert len(d._INDEX_CACHE) == 1
    first = next(iter(d._INDEX_CACHE.values()))
    assert len(first.__dict__.get("_layout_cache", {})) == 1  # table + gather plan remembered on the index
    # the same path with other content (one ORF fewer): parsed again, not served from the cache
    lines = open(index_path).read().splitlines(keepends=True)
    with open(index_path, "w") as fh:
        fh.writelines(lines[:-1])
    os.utime(index_path, ns=(os.stat(index_path).st_atime_ns, os.stat(index_path).st_mtime_ns + 1_000_000))
    prefix = str(tmp_path / "out2")
    d.export_orf_coverages(index_path, load_alignments(), prefix, report_all=True)
    rows = open(prefix + "_translating_ORFs.tsv").read().splitlines()
    assert len(rows) - 1 == len(lines) - 2  # header line of the index + the dropped ORF
    assert next(reversed(d._INDEX_CACHE.values())) is not first
    d.forget_indexes()



def _random_index_and_columns(path, seed, n_orfs=40_000):
    """A candidate index with overlapping / nested ORFs on both strands of a few chromosomes, spliced (1-5 exons, abutting
    exons included), lengths not always multiples of 3, written as the text prepare-orfs writes -- and merged P-site columns
    dense enough for clear winners, sparse enough for exact frame ties (some positions named by several rows)."""
    from ribotricer_amd.alignments import MergedColumns

    rng = np.random.default_rng(seed)
    chroms = [f"chr{k}" for k in range(1, 6)]
    lines = ["ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n"]
    rows = []
    for i in range(n_orfs):
        chrom, strand = int(rng.integers(0, len(chroms))), int(rng.integers(0, 2))
        k = int(rng.integers(1, 6))
        at = int(rng.integers(1, 3_000_000))
        ivs = []
        for _ in range(k):
            n = int(rng.integers(20, 400))
            ivs.append((at, at + n - 1))
            at += n + int(rng.choice([0, 0, 1, 50, 700]))  # abutting exons and introns
        coord = ",".join(f"{s}-{e}" for s, e in ivs)
        lines.append(f"o{i}\t{'annotated' if i % 7 == 0 else 'uORF'}\ttx{i}\tprotein_coding\tg{i}\tn{i}\tprotein_coding\t{chroms[chrom]}\t{'+-'[strand]}\tATG\t{coord}\n")
        dense = rng.random() < 0.5
        for s, e in ivs:
            m = int((e - s + 1) * (0.6 if dense else 0.03)) + 1
            pos = rng.integers(s, e + 1, size=m)
            if dense:
                pos = pos - (pos - s) % 3  # in-frame pile-ups: periodic profiles
            rows.append(np.stack([np.full(m, strand), np.full(m, chrom), pos, rng.integers(1, 6, size=m)], axis=1))
    with open(path, "w") as fh:
        fh.writelines(lines)
    r = np.concatenate(rows)
    cols = MergedColumns(r[:, 0].astype(np.uint8), r[:, 1].astype(np.int32), r[:, 2].astype(np.int64), r[:, 3].astype(np.int64), chroms)
    return cols


@pytest.mark.parametrize("report_all", [False, True])
def test_hip_cpu_and_sharded_exports_of_a_random_index_agree(tmp_path, report_all, monkeypatch):
    """Three roads, one file: the hip backend on one GPU, the hip backend over three slices (shards.IndexShards) and the cpu
    backend (the reference's own float64 arithmetic through the library's host entry points -- itself byte-identical to
    the reference on the golden fixtures) export a random 40 000-ORF index.  Same rows, every column the same text, the
    phase score within 1e-6 of the cpu backend's -- and on the exact frame ties (the reference's bits, replayed) equal."""
    from ribotricer_amd import detect_orfs as d

    index = str(tmp_path / "rnd_candidate_orfs.tsv")
    cols = _random_index_and_columns(index, seed=11 + int(report_all))
    d.forget_indexes()
    outs = {}
    for tag, backend, devices in (("hip", "hip", None), ("shards", "hip", [0, 0, 0]), ("cpu", "cpu", None)):
        monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", backend)
        prefix = str(tmp_path / tag)
        d.export_orf_coverages(index, cols, prefix, report_all=report_all, devices=devices)
        outs[tag] = open(prefix + "_translating_ORFs.tsv").read()
    d.forget_indexes()
    n_rows = outs["cpu"].count("\n") - 1
    assert n_rows == 40_000 if report_all else 1000 < n_rows < 40_000
    same_rows(outs["cpu"], outs["hip"])
    same_rows(outs["cpu"], outs["shards"])
    exact = sum(a.split("\t")[3] == b.split("\t")[3] for a, b in zip(outs["cpu"].splitlines()[1:], outs["hip"].splitlines()[1:]))
    assert exact >= 0.01 * n_rows  # (the replayed ties, phase 0 / 1 ORFs: the same bits on both roads)


def test_shards_come_and_go_without_leaving_anything_behind(tmp_path):
    """The lifetime rules of shards.IndexShards, rehearsed on the one GPU (devices=[0, 0, 0, 0]: four slices, four streams):
    samples of two different indexes alternate, default mode and report_all, with the shards released in between the three
    ways the product releases them -- the index leaving the cache (detect_orfs._forget_index), another device list for the
    same index, an explicit release().  Every file equals the one-GPU run's; a released shard's buffers are never touched
    again (they are poisoned and checked); scoring through a released object raises; and the device memory held when it is
    all over is what was held before."""
    import gc

    import torch

    from ribotricer_amd import detect_orfs as d

    paths, cols = {}, {}
    for tag, seed in (("a", 301), ("b", 302)):
        paths[tag] = str(tmp_path / f"{tag}_candidate_orfs.tsv")
        cols[tag] = _random_index_and_columns(paths[tag], seed=seed, n_orfs=12_000)
    d.forget_indexes()

    def export(tag, report_all, devices, name):
        d.export_orf_coverages(paths[tag], cols[tag], str(tmp_path / name), report_all=report_all, devices=devices)
        return open(str(tmp_path / name) + "_translating_ORFs.tsv").read()

    one = {(tag, ra): export(tag, ra, None, f"one_{tag}_{int(ra)}") for tag in "ab" for ra in (False, True)}
    d.forget_indexes()
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    reserved_before = torch.cuda.memory_reserved("cuda:0")

    def shards_of(tag):
        index = next(v for k, v in d._INDEX_CACHE.items() if k[0] == os.path.realpath(paths[tag]))
        (shards,) = index.__dict__["_shard_cache"].values()
        return shards

    poisoned = []  # buffers of released shards, filled with 0x7F bytes: nobody may write them again

    def buffers_of(shards):
        return [part["built"]["coverage"] for part in shards.parts if part["built"] is not None] + list(shards._host.values())

    def retire(shards_list, how):
        bufs = [t for sh in shards_list for t in buffers_of(sh)]
        assert bufs
        how()
        for sh in shards_list:
            assert sh.released and all(p["built"] is None for p in sh.parts) and not sh._host
        for t in bufs:
            t.view(torch.uint8).fill_(0x7F)
        torch.cuda.synchronize()
        poisoned.extend(bufs)

    four = [0, 0, 0, 0]
    for round_ in range(3):
        for tag in "ab":
            for ra in (False, True):
                same_rows(one[(tag, ra)], export(tag, ra, four, f"four_{tag}_{int(ra)}_{round_}"))
            assert shards_of(tag).plans_built == 4  # (built by the round's first sample of the index, reused by its second)
        sa, sb = shards_of("a"), shards_of("b")
        if round_ == 1:
            # another device list for the same index replaces its shards; an explicit release(); a released object refuses
            retire([sa], lambda: export("a", False, [0, 0], "two_a"))
            same_rows(one[("a", False)], open(str(tmp_path / "two_a_translating_ORFs.tsv")).read())
            assert shards_of("a") is not sa and len(shards_of("a").devices) == 2
            retire([sb], sb.release)
            with pytest.raises(RuntimeError):
                sb.score(cols["b"], None, True)
            d.forget_indexes()
        else:  # the indexes leave the cache (detect_orfs._forget_index)
            retire([sa, sb], d.forget_indexes)
    for t in poisoned:  # nothing wrote into a released buffer
        assert bool((t.view(torch.uint8) == 0x7F).all())
    del poisoned, sa, sb
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert torch.cuda.memory_reserved("cuda:0") <= reserved_before + (8 << 20), (torch.cuda.memory_reserved("cuda:0"), reserved_before)


def test_export_over_two_real_devices_equals_the_cpu_backend(tmp_path, monkeypatch):
    """Two DISTINCT devices (skipped on a one-GPU box): per-device column uploads, a thread, a stream and pinned staging per
    device -- against the cpu backend (the reference's own arithmetic) on a random index, default mode and report_all.  The
    phase text may differ between device counts by <= 1e-6 (every slice is tiled on its own: DESIGN.md section 5);
    every other column is the same text."""
    import torch

    from ribotricer_amd import detect_orfs as d

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    index = str(tmp_path / "rnd_candidate_orfs.tsv")
    cols = _random_index_and_columns(index, seed=77, n_orfs=30_000)
    for report_all in (False, True):
        d.forget_indexes()
        monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "hip")
        d.export_orf_coverages(index, cols, str(tmp_path / "two"), report_all=report_all, devices=[0, 1])
        monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "cpu")
        d.export_orf_coverages(index, cols, str(tmp_path / "cpu"), report_all=report_all)
        same_rows(open(str(tmp_path / "cpu_translating_ORFs.tsv")).read(), open(str(tmp_path / "two_translating_ORFs.tsv")).read())
    d.forget_indexes()


@pytest.mark.parametrize("seed", [41, 42, 43, 44, 45, 46])
def test_corner_indexes_hip_equals_cpu_backend(tmp_path, seed, monkeypatch):
    """Fresh corner-case indexes (tests/golden/random_index.py: overlapping / nested / duplicated / 1-nt exons, shuffled
    lists, a '.' strand, blocks with end < start, dressed numbers, CRLF, no final newline) through the HIP export -- one GPU
    and three slices, default mode and report_all -- against the cpu backend, whose bytes equal the reference's on this
    generator's indexes (tests/golden/check_export_vs_reference.py; G12).  Zero-interval ORFs, ORFs whose exons overlap
    inside one gather plan, groups without a single position: every column the same text, the phase within 1e-6."""
    import sys

    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    from random_index import random_index

    from ribotricer_amd import detect_orfs as d

    shape = [dict(), dict(malformed=0.15), dict(dressed=0.2), dict(crlf=True), dict(final_newline=False, malformed=0.1),
             dict(malformed=0.3, dressed=0.3)][seed % 6]
    text, merged = random_index(2500, seed, **shape)
    index = str(tmp_path / "corner_candidate_orfs.tsv")
    with open(index, "w", newline="") as fh:
        fh.write(text)
    for report_all in (False, True):
        outs = {}
        for tag, backend, devices in (("cpu", "cpu", None), ("hip", "hip", None), ("three", "hip", [0, 0, 0])):
            monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", backend)
            d.forget_indexes()
            d.export_orf_coverages(index, merged, str(tmp_path / tag), report_all=report_all, devices=devices)
            outs[tag] = open(str(tmp_path / tag) + "_translating_ORFs.tsv", newline="").read()
        assert outs["cpu"].count("\n") - 1 == (2500 if report_all else outs["cpu"].count("\ttranslating\t"))
        same_rows(outs["cpu"], outs["hip"])
        same_rows(outs["cpu"], outs["three"])
    d.forget_indexes()
