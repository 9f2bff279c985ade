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
