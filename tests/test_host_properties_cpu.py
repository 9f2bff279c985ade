"""Hypothesis properties of the native host ends (no GPU): whatever the index text or the
per-ORF numbers are, the C++ parser / renderer agree with the readable Python statements
(`detect_orfs.parse_index_line`, `detect_orfs.format_rows`)."""

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from ribotricer_amd import detect_orfs as d
from ribotricer_amd import tsv
from ribotricer_amd.index import NativeIndex

field = st.text(alphabet=st.characters(blacklist_characters="\t\n\r", blacklist_categories=("Cs",)), max_size=12)
interval = st.tuples(st.integers(1, 10**9), st.integers(0, 5000)).map(lambda t: (t[0], t[0] + t[1]))
line = st.tuples(
    st.lists(field, min_size=8, max_size=8), st.sampled_from(["+", "-", ".", ""]), st.sampled_from(["", "A", "AT", "ATG", "ATGC", "éééé"]),
    st.lists(interval, min_size=1, max_size=7), st.sampled_from(["\n", "\r\n", ""]),
)


def render(fields, strand, codon, ivs, _eol):
    cols = fields[:8] + [strand, codon, ",".join(f"{s}-{e}" for s, e in ivs)]
    return "\t".join(cols)


@settings(max_examples=150, deadline=None)
@given(st.lists(line, min_size=0, max_size=6))
def test_native_index_equals_python_parser_on_random_lines(lines):
    body = "".join(render(*ln) + (ln[4] if k == len(lines) - 1 else "\n") for k, ln in enumerate(lines))
    ni = NativeIndex(("header\n" + body).encode("utf-8"))
    want = [d.parse_index_line(render(*ln)) for ln in lines]
    assert ni.records() == [w._replace(start_codon=None if w.start_codon in (None, "None") else w.start_codon) for w in want]
    assert ni.length.tolist() == [sum(e - s + 1 for s, e in w.intervals) for w in want]
    assert ni.reverse.tolist() == [1 if w.strand == "-" else 0 for w in want]


@settings(max_examples=100, deadline=None)
@given(
    st.lists(st.lists(st.integers(0, 2**24 - 1), max_size=40), min_size=0, max_size=8),
    st.randoms(use_true_random=False),
)
def test_native_rows_equal_python_rows_on_random_numbers(profiles, rnd):
    n = len(profiles)
    counts = np.array([v for p in profiles for v in p], np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(p) for p in profiles])]).astype(np.int64)
    res = dict(
        phase=np.array([rnd.choice([0.0, 1.0, rnd.random(), rnd.random() * 1e-7, 1 / 3]) for _ in range(n)], np.float64),
        valid=np.array([rnd.randrange(0, 10**6) for _ in range(n)], np.int32),
        read_count=np.array([rnd.randrange(0, 2**53) for _ in range(n)], np.int64),
        status=np.array([rnd.randrange(2) for _ in range(n)], np.uint8),
    )
    records = [d.IndexRecord(f"id{i}", "t", f"tx{i}", "a", "g", "n", "b", "chr", "+", None if i % 2 else "ATG", ()) for i in range(n)]
    for report_all in (False, True):
        want = "".join(d.format_rows(records, counts, offsets, res, report_all)).encode("utf-8")
        got = b"".join(tsv.format_rows_native(counts, offsets, res["phase"], res["valid"], res["read_count"], res["status"],
                                              tsv.record_tables(records), report_all, chunk_bytes=256, threads=1))
        assert got == want


def test_profile_slices_cover_every_orf_once_in_order():
    """detect_orfs._profile_slices (host arrays): consecutive ORF ranges of about slice_nt nucleotides,
    every profile exactly once, empty profiles and ORFs longer than a slice included."""
    from ribotricer_amd.detect_orfs import _profile_slices

    rng = np.random.default_rng(0)
    lens = rng.integers(0, 500, 5000)
    lens[[17, 2500]] = [30_000, 0]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    counts = rng.integers(0, 9, int(off[-1])).astype(np.int32)
    last, parts = 0, []
    for a, b, part in _profile_slices(counts, off, slice_nt=20_000):
        assert a == last and b > a and part.size == off[b] - off[a]
        last = b
        parts.append(part)
    assert last == 5000 and np.array_equal(np.concatenate(parts), counts)
    assert list(_profile_slices(counts[:0], np.zeros(1, np.int64))) == []
    only_empty = list(_profile_slices(counts[:0], np.zeros(4, np.int64)))
    assert [(a, b) for a, b, _ in only_empty] == [(0, 3)]


def test_status_host_is_the_reference_predicate():
    """engine.status_host (used when a tie is resolved on the host) == detect_orfs.py:281,285-299."""
    from helpers import reference_status
    from ribotricer_amd.engine import make_filter, status_host

    rng = np.random.default_rng(2)
    n = 4000
    phase = rng.random(n)
    lengths = rng.integers(0, 900, n)
    valid = rng.integers(0, 40, n).astype(np.int32)
    read_count = rng.integers(0, 500, n)
    min_cov = rng.integers(0, 3, n).astype(np.int32)
    for kw in (dict(), dict(phase_score_cutoff=0.3, min_valid_codons=8, min_reads_per_codon=1, min_valid_codons_ratio=0.25, min_density_over_orf=0.5)):
        got = status_host(make_filter(**kw), phase, valid, read_count, min_cov, lengths)
        want = reference_status(phase, valid, read_count, min_cov, lengths, cutoff=kw.get("phase_score_cutoff", 0.428571428571),
                                min_valid=kw.get("min_valid_codons", 5), min_reads=kw.get("min_reads_per_codon", 0),
                                min_ratio=kw.get("min_valid_codons_ratio", 0), min_density=kw.get("min_density_over_orf", 0.0))
        assert np.array_equal(got, want)


def test_resolve_big_ties_patches_only_flagged_orfs(g8):
    """engine.resolve_big_ties on host data: ORFs flagged RP_FLAG_BIGTIE get the host replay's phase /
    valid_codons (the reference's bits, fixture G8) and a recomputed status; nothing else is touched."""
    from ribotricer_amd import _lib
    from ribotricer_amd.engine import csr_profiles_of, make_filter, resolve_big_ties

    counts, offsets = g8["counts"], g8["offsets"]
    n = offsets.size - 1
    flagged = np.zeros(n, bool)
    flagged[::7] = True
    res = {
        "phase": np.full(n, 0.5), "valid": np.full(n, 3, np.int32), "read_count": np.ones(n, np.int64),
        "min_codon_cov": np.zeros(n, np.int32), "flags": np.where(flagged, _lib.FLAG_TIE | _lib.FLAG_REPLAY | _lib.FLAG_BIGTIE, 0).astype(np.uint8),
        "status": np.zeros(n, np.uint8),
    }
    th = make_filter(phase_score_cutoff=0.9, min_valid_codons=1)
    assert resolve_big_ties(res, csr_profiles_of(counts, offsets), th) == int(flagged.sum())
    assert np.array_equal(res["phase"][flagged], g8["phase"][flagged]) and np.array_equal(res["valid"][flagged], g8["valid"][flagged])
    assert np.all(res["phase"][~flagged] == 0.5) and np.all(res["valid"][~flagged] == 3) and np.all(res["status"][~flagged] == 0)
    assert np.array_equal(res["status"][flagged], ((g8["phase"][flagged] >= 0.9) & (g8["valid"][flagged] >= 1)).astype(np.uint8))


def test_usable_cores_is_positive():
    from ribotricer_amd import _lib

    assert 1 <= _lib.usable_cores() <= 4096
