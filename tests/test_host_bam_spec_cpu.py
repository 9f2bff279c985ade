"""The native BAM reader (rp_bam_split_host, csrc/rp_bam.hpp) against a BAM file assembled field by
field from the SAM/BAM specification (tests/golden/make_spec_bam.py: the worked example of SAMv1
section 1.1 plus one read per rule of the reference's decision list) with EXPECTED TABLES DERIVED BY
HAND from ribotricer/bam.py:73-135 and common.py:33-70 (tests/golden/g9_spec_expected.json).
Neither the bytes nor the expectations come from the repo's own test writer / restatement
(tests/tools/bamwriter.py), which the other front-end tests use.  pysam is not installable here, so
this is as close to the reference's own reader as the container allows."""

import gzip
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN

BAM = os.path.join(GOLDEN, "g9_spec.bam")
with open(os.path.join(GOLDEN, "g9_spec_expected.json")) as _fh:
    EXPECT = json.load(_fh)


def _table(rows):
    out = {}
    for length, strand, chrom, pos, count in rows:
        out[(length, strand, chrom, pos)] = count
    return out


def _nested_as_table(nested):
    return {(ln, s, c, p): n for ln, by_strand in nested.items() for s, counter in by_strand.items() for (c, p), n in counter.items()}


def test_the_committed_bytes_are_a_bgzf_bam_for_an_independent_decoder():
    """Python's gzip module (multi-member gzip = BGZF) inflates the file; the BAM magic, the header
    text and the reference list of the specification's example come out."""
    raw = gzip.open(BAM, "rb").read()
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<I", raw, 4)[0]
    text = raw[8 : 8 + l_text].decode()
    assert "@SQ\tSN:ref\tLN:45" in text and "@SQ\tSN:ref2\tLN:40" in text
    assert struct.unpack_from("<I", raw, 8 + l_text)[0] == 2
    data = open(BAM, "rb").read()
    assert data.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # the EOF marker of section 4.1.2
    assert data.count(b"\x1f\x8b\x08\x04") >= 10  # many small blocks: records straddle them


@pytest.mark.parametrize("protocol", ["forward", "reverse"])
def test_split_bam_matches_the_hand_derived_table(tmp_path, protocol):
    from ribotricer_amd.alignments import split_bam

    prefix = str(tmp_path / "s")
    cols, counts = split_bam(BAM, protocol, prefix)
    assert _nested_as_table(cols.as_nested()) == _table(EXPECT[protocol]["rows"])
    assert list(counts.items()) == [tuple(x) for x in EXPECT[protocol]["read_length_counts"]]  # first-met order (the reference's dict)
    st = EXPECT["stats"]
    summary = open(prefix + "_bam_summary.txt").read()
    assert summary.startswith(
        f"summary:\n\ttotal_reads: {st['total']}\n\tunique_mapped: {st['valid']}\n\tqcfail: {st['qcfail']}\n\tduplicate: {st['duplicate']}\n"
        f"\tsecondary: {st['secondary']}\n\tunmapped:{st['unmapped']}\n\tmulti:{st['multi']}\n\nlength dist:\n"
    )
    for length, n in sorted(tuple(x) for x in EXPECT[protocol]["read_length_counts"]):
        assert f"\t{length}: {n}\n" in summary
    assert st["valid"] + st["qcfail"] + st["duplicate"] + st["secondary"] + st["unmapped"] + st["multi"] == st["total"]


def test_read_length_filter(tmp_path):
    from ribotricer_amd.alignments import split_bam

    e = EXPECT["forward_lengths_28_9"]
    cols, counts = split_bam(BAM, "forward", str(tmp_path / "f"), read_lengths=e["read_lengths"])
    assert _nested_as_table(cols.as_nested()) == _table(e["rows"])
    assert list(counts.items()) == [tuple(x) for x in e["read_length_counts"]]
    assert f"unique_mapped: {e['valid']}\n" in open(str(tmp_path / "f") + "_bam_summary.txt").read()
    assert int(np.asarray(cols.count).sum()) == e["valid"]
