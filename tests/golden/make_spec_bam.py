#!/usr/bin/env python3
"""Assemble tests/golden/g9_spec.bam FIELD BY FIELD from the SAM/BAM format specification
(SAMv1, sections 1.1, 4.1, 4.2), independently of the repo's own BAM test writer
(tests/tools/bamwriter.py) and of the reader under test (csrc/rp_bam.hpp).

    python tests/golden/make_spec_bam.py        # rewrites g9_spec.bam (the bytes are committed)

The file holds the six alignments of the specification's worked example (section 1.1: r001/99,
r002, r003, r004, r003/2064, r001/147 on reference `ref`, LN:45) followed by reads that walk the
decision list of ribotricer/bam.py:73-135 one rule at a time: every NH tag type (c C s S i, and A),
MAPQ 255 without NH, NH beating MAPQ, unmapped / secondary / QC-fail / duplicate flags and their
precedence, both strands, soft and hard clips, I / D / N / P / = / X operations, a second
reference.  The EXPECTED tables (g9_spec_expected.json) are not produced by this script: they were
derived by hand from bam.py:73-135 and common.py:33-70, read by read (see the comments below), and
are committed as literals.

BGZF framing (section 4.1): blocks of uneven size so that records straddle block boundaries,
three compression levels including stored (level 0) blocks, one block whose gzip extra field
carries a foreign subfield in front of `BC`, one empty data block in the middle, and the
28-byte EOF marker block quoted in the specification.
"""

from __future__ import annotations

import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))

CIGAR_OPS = "MIDNSHP=X"  # section 4.2: op codes 0..8
SEQ_CODES = "=ACMGRSVTWYHKDBN"  # section 4.2.3: 4-bit base codes

REFS = [("ref", 45), ("ref2", 40)]

# name, flag, reference, POS (1-based, 0 = unavailable), MAPQ, CIGAR, SEQ ('*' = none), tags [(tag, type, value)]
READS = [
    # --- the worked example of section 1.1, verbatim (MAPQ 30 / 17, no NH tag): under common.py:60-69 a read
    #     without NH whose MAPQ is neither 255 nor < 1 and whose flag is not in [4, 20, 256, 272, 2048] is
    #     "unable to determine" -> None -> `not None` -> counted as multi (bam.py:92-94)
    ("r001", 99, "ref", 7, 30, "8M2I4M1D3M", "TTAGATAAAGGATACTG", []),
    ("r002", 0, "ref", 9, 30, "3S6M1P1I4M", "AAAAGATAAGGATA", []),
    ("r003", 0, "ref", 9, 30, "5S6M", "GCCTAAGCTAA", [("SA", "Z", "ref,29,-,6H5M,17,0;")]),
    ("r004", 0, "ref", 16, 30, "6M14N5M", "ATAGCTTCAGC", []),
    ("r003", 2064, "ref", 29, 17, "6H5M", "TAGGC", [("SA", "Z", "ref,9,+,5S6M,30,1;")]),
    ("r001", 147, "ref", 37, 30, "9M", "CAGCGGCAT", [("NM", "i", 1)]),
    # --- unique by MAPQ 255, no NH (common.py:60-61).  8M -> 0-based 6..13, 2I -, 4M -> 14..17, 1D skips 18,
    #     3M -> 19..21: 15 reference positions; '+' mapping: forward keeps '+', 5' end = first = 6 -> (ref, 7);
    #     reverse flips to '-', 5' end = last = 21 -> (ref, 22)
    ("u01", 0, "ref", 7, 255, "8M2I4M1D3M", None, []),
    # --- NH:i:1 decides although MAPQ is 3; reverse-strand mapping.  3S -, 6M -> 8..13, 1P -, 1I -, 4M -> 14..17:
    #     length 10; forward: '-', last = 17 -> (ref, 18); reverse: '+', first = 8 -> (ref, 9)
    ("u02", 16, "ref", 9, 3, "3S6M1P1I4M", None, [("NH", "i", 1)]),
    # --- NH:C:1 with MAPQ 0.  6M -> 15..20, 14N skips 21..34, 5M -> 35..39: length 11; forward (ref, 16); reverse '-' (ref, 40)
    ("u03", 0, "ref", 16, 0, "6M14N5M", None, [("NH", "C", 1)]),
    # --- NH:c:1, hard clip, reverse strand.  6H -, 5M -> 28..32: length 5; forward '-' last -> (ref, 33); reverse '+' first -> (ref, 29)
    ("u04", 16, "ref", 29, 40, "6H5M", None, [("NH", "c", 1)]),
    # --- '=' and 'X' operations count like M (pysam get_reference_positions); NH:s:1.  36..44: length 9; forward (ref, 37); reverse '-' (ref, 45)
    ("u05", 0, "ref", 37, 11, "4=1X4=", None, [("NH", "s", 1)]),
    # --- NH:S:1; the same key as u05 -> its counter goes to 2
    ("u06", 0, "ref", 37, 12, "9M", None, [("NH", "S", 1)]),
    # --- NH = 2 and NH = 0: not unique -> multi
    ("m01", 0, "ref", 5, 255, "10M", None, [("NH", "i", 2)]),
    ("m02", 0, "ref", 5, 255, "10M", None, [("NH", "C", 0)]),
    # --- the flag chain of bam.py:77-91: qcfail before duplicate before secondary before unmapped
    ("q01", 512, "ref", 5, 255, "10M", None, [("NH", "i", 1)]),
    ("q02", 1536, "ref", 5, 255, "10M", None, [("NH", "i", 1)]),   # QC-fail + duplicate -> qcfail
    ("d01", 1024, "ref", 5, 255, "10M", None, [("NH", "i", 1)]),
    ("d02", 1280, "ref", 5, 255, "10M", None, [("NH", "i", 1)]),   # duplicate + secondary -> duplicate
    ("s01", 256, "ref", 5, 255, "10M", None, [("NH", "i", 1)]),
    ("s02", 260, "ref", 5, 255, "10M", None, [("NH", "i", 1)]),    # secondary + unmapped -> secondary
    ("x01", 4, None, 0, 0, "*", "ACGTACGTAC", []),                  # unmapped, no coordinates
    ("x02", 4, "ref", 12, 0, "*", "ACGTACGTAC", []),                # unmapped, placed with its mate
    # --- second reference, 28-nt reads on both strands.  2S28M at POS 3 -> 2..29: forward '-' last -> (ref2, 30); reverse '+' -> (ref2, 3)
    ("u07", 16, "ref2", 3, 1, "2S28M", None, [("NH", "i", 1)]),
    ("u08", 0, "ref2", 3, 255, "28M", None, []),                   # forward '+' (ref2, 3); reverse '-' (ref2, 30)
    # --- NH present but not the integer 1: the character '1' (Python: '1' == 1 is False) -> multi; NH:i:3 beats MAPQ 255 -> multi
    ("u09", 0, "ref", 5, 255, "10M", None, [("NH", "A", "1")]),
    ("u10", 0, "ref", 5, 255, "10M", None, [("XS", "i", 7), ("NH", "i", 3)]),
    # --- supplementary (0x800) is none of the four filtered flags; NH:i:1 -> counted.  5M at POS 20 -> 19..23: length 5;
    #     forward '+' (ref, 20); reverse '-' last = 23 -> (ref, 24)
    ("sup", 2048, "ref", 20, 9, "5M", None, [("AS", "i", 40), ("NH", "i", 1)]),
    # --- deletions and a skip inside: 3M -> 0..2, 2D skips 3..4, 2M -> 5..6, 1N skips 7, 3M -> 8..10: length 8;
    #     forward '+' (ref, 1); reverse '-' (ref, 11)
    ("u12", 0, "ref", 1, 60, "3M2D2M1N3M", None, [("NM", "i", 2), ("NH", "i", 1), ("MD", "Z", "3^AC5")]),
]


def parse_cigar(text):
    if text == "*":
        return []
    ops, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            ops.append((CIGAR_OPS.index(ch), int(num)))
            num = ""
    return ops


def reg2bin(beg, end):
    """Section 5.3 of the specification (the C function, transcribed)."""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def record(read, refs=None):
    refs = REFS if refs is None else refs
    name, flag, ref, pos1, mapq, cigar_text, seq, tags = read
    real_cigar = None
    if isinstance(cigar_text, tuple):  # (placeholder, real CIGAR): > 65 535 operations, section 4.2.2
        cigar_text, real_cigar = cigar_text
    cigar = parse_cigar(cigar_text)
    if real_cigar is not None:
        real = parse_cigar(real_cigar)
        assert len(real) > 65535
        tags = [("CG", "B", ("I", [(n << 4) | op for op, n in real]))] + list(tags)
    query_len = sum(n for op, n in cigar if CIGAR_OPS[op] in "MIS=X")
    if real_cigar is not None:
        query_len = sum(n for op, n in parse_cigar(real_cigar) if CIGAR_OPS[op] in "MIS=X")
    if seq is None:
        seq = ("ACGT" * (query_len // 4 + 1))[:query_len]
    if cigar and real_cigar is None:
        assert len(seq) == query_len, name
    ref_id = -1 if ref is None else [r[0] for r in refs].index(ref)
    pos0 = pos1 - 1
    ref_len = sum(n for op, n in (parse_cigar(real_cigar) if real_cigar is not None else cigar) if CIGAR_OPS[op] in "MDN=X")
    end0 = pos0 + (ref_len if ref_len else 1)
    bin_ = reg2bin(pos0, end0) if pos0 >= 0 else 4680  # reg2bin(-1, 0), the value for unplaced reads
    read_name = name.encode() + b"\0"
    body = struct.pack("<i", ref_id)                      # refID
    body += struct.pack("<i", pos0)                       # pos (0-based)
    body += struct.pack("<B", len(read_name))             # l_read_name
    body += struct.pack("<B", mapq)                       # mapq
    body += struct.pack("<H", bin_)                       # bin
    body += struct.pack("<H", len(cigar))                 # n_cigar_op
    body += struct.pack("<H", flag)                       # flag
    body += struct.pack("<I", len(seq))                   # l_seq
    body += struct.pack("<i", -1)                         # next_refID
    body += struct.pack("<i", -1)                         # next_pos
    body += struct.pack("<i", 0)                          # tlen
    body += read_name
    for op, n in cigar:
        body += struct.pack("<I", (n << 4) | op)          # cigar: op_len << 4 | op
    packed = bytearray((len(seq) + 1) // 2)
    for k, base in enumerate(seq):                        # seq: 4-bit codes, high nibble first
        packed[k // 2] |= SEQ_CODES.index(base) << (4 if k % 2 == 0 else 0)
    body += bytes(packed)
    body += b"\xff" * len(seq)                            # qual: 0xFF = not stored ('*')
    for tag, ty, val in tags:
        body += tag.encode() + ty.encode()
        fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
        if ty in "ZH":  # NUL-terminated text (H: hex digits)
            body += val.encode() + b"\0"
        elif ty == "A":
            body += val.encode()
        elif ty == "B":  # subtype, count, elements (section 4.2.4)
            sub, items = val
            body += sub.encode() + struct.pack("<I", len(items)) + b"".join(struct.pack(fmt[sub], x) for x in items)
        else:
            body += struct.pack(fmt[ty], val)
    return struct.pack("<I", len(body)) + body            # block_size


def bam_header(refs=None):
    refs = REFS if refs is None else refs
    text = "@HD\tVN:1.6\tSO:unsorted\n" + "".join(f"@SQ\tSN:{n}\tLN:{ln}\n" for n, ln in refs)
    out = b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(refs))
    for n, ln in refs:
        out += struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", ln)
    return out


def bam_stream(reads=None, refs=None):
    reads = READS if reads is None else reads
    return bam_header(refs) + b"".join(record(r, refs) for r in reads)


# ---------------------------------------------------------------------------------------------------------------
# g11_spec_hard.bam (round 4): the corners of the format a Ribo-seq BAM rarely shows and a reader must still survive,
# written as an UNCOMPRESSED BAM (`samtools view -u`: every BGZF block a stored deflate block).  Expected tables
# (g11_spec_hard_expected.json) derived by hand, read by read:
REFS_HARD = [("ref", 45), ("ref2", 40), ("long", 70000)]
LONG_REAL = "1M1D" * 32768 + "1M"  # 65 537 operations: 32 769 aligned bases at 0, 2, 4, ..., 65 536 (0-based)
HARD_READS = [
    # --- more than 65 535 CIGAR operations (section 4.2.2): the record's own CIGAR is the placeholder <l_seq>S<ref span>N,
    #     the real one sits in CG:B:I.  pysam hands the real one to get_reference_positions(): 32 769 positions,
    #     first 0, last 65 536; flag 0, MAPQ 255 + NH:i:1 -> unique.  forward '+' (long, 1); reverse '-' (long, 65537)
    ("lng", 0, "long", 1, 255, ("32769S65537N", LONG_REAL), None, [("NH", "i", 1)]),
    # --- B arrays of every element width, a Z text that spells a tag and an H string IN FRONT of NH: the scanner has to
    #     step over them by their declared sizes.  NH:i:1 with MAPQ 0 -> unique.  flag 16; 10M at POS 5 -> 4..13:
    #     forward '-' last -> (ref, 14); reverse '+' first -> (ref, 5)
    ("arr", 16, "ref", 5, 0, "10M", None, [("ZB", "B", ("c", [-1, 2, 3])), ("ZC", "B", ("S", [1, 65535])), ("ZF", "B", ("f", [1.5, 2.5])),
                                           ("XZ", "Z", "text NH:i:7"), ("XH", "H", "1AE301"), ("NH", "i", 1)]),
    # --- NH as a float: dict(read.get_tags())["NH"] == 1 holds for 1.0 (unique), not for 1.5 (multi).  12M at POS 20 ->
    #     19..30: forward '+' (ref, 20); reverse '-' (ref, 31)
    ("flt", 0, "ref", 20, 0, "12M", None, [("NH", "f", 1.0)]),
    ("fl2", 0, "ref", 20, 255, "12M", None, [("NH", "f", 1.5)]),
    # --- l_read_name at its maximum (254 characters + NUL) and at its minimum (the NUL alone); MAPQ 255, no NH -> unique.
    #     10M at POS 5 -> 4..13: forward '+' (ref, 5); reverse '-' (ref, 14).  8M at POS 30 -> 29..36: (ref, 30) / (ref, 37)
    ("n" * 254, 0, "ref", 5, 255, "10M", None, []),
    ("", 0, "ref", 30, 255, "8M", None, []),
    # --- NH twice: dict() keeps the LAST one.  28M at POS 3 on ref2 -> 2..29: forward '+' (ref2, 3); reverse '-' (ref2, 30)
    ("two", 0, "ref2", 3, 255, "28M", None, [("NH", "i", 2), ("NH", "i", 1)]),   # -> 1 -> unique
    ("tw2", 0, "ref2", 3, 255, "28M", None, [("NH", "i", 1), ("NH", "i", 2)]),   # -> 2 -> multi
    # --- array BYTES that spell a tag: 'N' 'H' 'C' 1 inside ZB:B:C is data, not NH:C:1 -> no NH, MAPQ 0 -> multi;
    #     'N' 'H' 'i' 2 0 0 0 is not NH:i:2 -> no NH, MAPQ 255 -> unique: the same keys as the 254-character read
    ("byt", 0, "ref", 5, 0, "10M", None, [("ZB", "B", ("C", [78, 72, 67, 1]))]),
    ("byu", 0, "ref", 5, 255, "10M", None, [("ZB", "B", ("C", [78, 72, 105, 2, 0, 0, 0]))]),
]


def write_hard():
    raw = bam_stream(HARD_READS, REFS_HARD)
    blocks, at = [], 0
    sizes = [65280, 4096, 60000, 1, 65280]  # whole-payload stored blocks: the 311 KB record of `lng` straddles several
    k = 0
    while at < len(raw):
        n = sizes[k % len(sizes)]
        blocks.append(bgzf_block(raw[at : at + n], level=0))
        at += n
        k += 1
    path = os.path.join(HERE, "g11_spec_hard.bam")
    with open(path, "wb") as fh:
        fh.write(b"".join(blocks) + EOF_MARKER)
    print(f"g11_spec_hard.bam: {len(HARD_READS)} reads, {len(raw)} bytes of BAM in {len(blocks)} stored BGZF blocks + EOF marker, {os.path.getsize(path)} bytes")


def bgzf_block(data: bytes, level: int, foreign_subfield: bool = False) -> bytes:
    """One BGZF block (section 4.1): a gzip member whose extra field holds the BC subfield."""
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    extra = b""
    if foreign_subfield:  # any other subfield may precede BC (RFC 1952: SI1 SI2 LEN data)
        extra += b"XY" + struct.pack("<H", 2) + b"\x2a\x2a"
    xlen = len(extra) + 6
    bsize = 12 + xlen + len(cdata) + 8 - 1  # total block size minus 1
    extra += b"BC" + struct.pack("<H", 2) + struct.pack("<H", bsize)
    header = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, xlen)  # ID1 ID2 CM FLG MTIME XFL OS XLEN
    return header + extra + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


EOF_MARKER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")  # section 4.1.2


def main():
    raw = bam_stream()
    sizes = [97, 211, 300, 64, 1, 509, 150]
    blocks, at, k = [], 0, 0
    while at < len(raw):
        n = sizes[k % len(sizes)]
        blocks.append(bgzf_block(raw[at : at + n], level=(0, 1, 9)[k % 3], foreign_subfield=(k == 2)))
        at += n
        k += 1
        if k == 4:
            blocks.append(bgzf_block(b"", level=6))  # an empty data block in the middle of the file
    path = os.path.join(HERE, "g9_spec.bam")
    with open(path, "wb") as fh:
        fh.write(b"".join(blocks) + EOF_MARKER)
    print(f"g9_spec.bam: {len(READS)} reads, {len(raw)} bytes of BAM in {len(blocks)} BGZF blocks + EOF marker, {os.path.getsize(path)} bytes")
    write_hard()


if __name__ == "__main__":
    main()
