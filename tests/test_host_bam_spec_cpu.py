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


# ------------------------------------------------------------------------------------------------------------
# g11_spec_hard.bam (round 4): corners of the format -- a CIGAR of 65 537 operations behind the CG:B:I tag, B arrays /
# Z / H tags in front of NH, NH as a float, NH twice, array bytes that spell a tag, l_read_name 255 and 1, reads of
# 32 769 aligned bases -- in an UNCOMPRESSED BAM (`samtools view -u`: stored deflate blocks).  Expectations by hand.
HARD = os.path.join(GOLDEN, "g11_spec_hard.bam")
with open(os.path.join(GOLDEN, "g11_spec_hard_expected.json")) as _fh:
    EXPECT_HARD = json.load(_fh)


def test_the_hard_file_is_an_uncompressed_bam_for_an_independent_decoder():
    raw = gzip.open(HARD, "rb").read()
    assert raw[:4] == b"BAM\x01"
    data = open(HARD, "rb").read()
    assert len(data) > len(raw)  # stored blocks: nothing is compressed (`-u`)
    l_text = struct.unpack_from("<I", raw, 4)[0]
    at = 8 + l_text
    assert struct.unpack_from("<I", raw, at)[0] == 3
    # the first record: n_cigar_op == 2 (the <l_seq>S<span>N placeholder), l_seq 32 769, and a CG:B:I tag of 65 537 operations
    at += 4
    for _ in range(3):
        l_name = struct.unpack_from("<I", raw, at)[0]
        at += 4 + l_name + 4
    block = struct.unpack_from("<I", raw, at)[0]
    rec = raw[at + 4 : at + 4 + block]
    l_read_name, n_cigar, l_seq = rec[8], struct.unpack_from("<H", rec, 12)[0], struct.unpack_from("<I", rec, 16)[0]
    assert (n_cigar, l_seq) == (2, 32769)
    cig = struct.unpack_from("<II", rec, 32 + l_read_name)
    assert cig == ((32769 << 4) | 4, (65537 << 4) | 3)
    aux = rec[32 + l_read_name + 8 + (l_seq + 1) // 2 + l_seq :]
    assert aux[:4] == b"CGBI" and struct.unpack_from("<I", aux, 4)[0] == 65537


@pytest.mark.parametrize("protocol", ["forward", "reverse"])
def test_split_bam_matches_the_hand_derived_table_on_the_hard_file(tmp_path, protocol):
    from ribotricer_amd.alignments import split_bam

    prefix = str(tmp_path / "h")
    cols, counts = split_bam(HARD, protocol, prefix)
    assert _nested_as_table(cols.as_nested()) == _table(EXPECT_HARD[protocol]["rows"])
    assert list(counts.items()) == [tuple(x) for x in EXPECT_HARD[protocol]["read_length_counts"]]
    st = EXPECT_HARD["stats"]
    summary = open(prefix + "_bam_summary.txt").read()
    assert summary.startswith(
        f"summary:\n\ttotal_reads: {st['total']}\n\tunique_mapped: {st['valid']}\n\tqcfail: {st['qcfail']}\n\tduplicate: {st['duplicate']}\n"
        f"\tsecondary: {st['secondary']}\n\tunmapped:{st['unmapped']}\n\tmulti:{st['multi']}\n\nlength dist:\n"
    )
    assert "\t32769: 1\n" in summary  # a read of 32 769 aligned bases is counted like any other (bam.py:99-131)
    # the read-length filter reaches the long read too
    cols2, counts2 = split_bam(HARD, protocol, str(tmp_path / "h2"), read_lengths=[32769, 8])
    assert dict(counts2) == {32769: 1, 8: 1} and int(np.asarray(cols2.count).sum()) == 2


def test_more_than_4_gib_of_uncompressed_bam(tmp_path):
    """A BAM whose uncompressed stream passes 2^32 bytes (a deep Ribo-seq run): 70 000 copies of one BGZF block of whole
    records behind the header -- 4.5 GB uncompressed, 20 MB on disk.  Every offset and counter the reader keeps must
    be 64-bit: the one key's count is the number of records written."""
    import sys

    sys.path.insert(0, GOLDEN)
    import make_spec_bam as m

    from ribotricer_amd.alignments import split_bam

    read = ("u08", 0, "ref2", 3, 255, "28M", None, [])  # forward '+' (ref2, 3), 28 nt (the g9 read of that name)
    rec = m.record(read)
    per_block = 65000 // len(rec)
    block = m.bgzf_block(rec * per_block, level=6)
    copies = (1 << 32) // (per_block * len(rec)) + 700
    path = str(tmp_path / "big.bam")
    with open(path, "wb") as fh:
        fh.write(m.bgzf_block(m.bam_header(), level=6))
        for _ in range(copies // 1000):
            fh.write(block * 1000)
        fh.write(block * (copies % 1000))
        fh.write(m.EOF_MARKER)
    n = copies * per_block
    assert n * len(rec) > (1 << 32)
    cols, counts = split_bam(path, "forward", str(tmp_path / "big"))
    assert dict(counts) == {28: n}
    assert _nested_as_table(cols.as_nested()) == {(28, "+", "ref2", 3): n}
    assert f"total_reads: {n}\n\tunique_mapped: {n}\n" in open(str(tmp_path / "big") + "_bam_summary.txt").read()
