"""Test infrastructure: a minimal BAM writer (struct + zlib -> BGZF) and a restatement of the
decision list of ribotricer/bam.py:73-135 + common.py:33-70 on plain read records, so that the
native reader (rp_bam_split_host) can be exercised without pysam.

pysam is not installed in the build container, so the reference's own split_bam cannot be run
on these files: this piece of the front end is checked against the restatement below only
(DESIGN.md says so: "BAM parsing: parity unpinned").
"""

from __future__ import annotations

import struct
import zlib
from collections import Counter, defaultdict

CIGAR_OPS = "MIDNSHP=X"
SAM_NOT_UNIQ_FLAGS = [4, 20, 256, 272, 2048]  # common.py:30


def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25  # 12 header + 6 extra + cdata + 8 trailer - 1
    return (
        struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6)
        + struct.pack("<BBHH", 66, 67, 2, bsize)
        + cdata
        + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))
    )


def _record(read, ref_ids) -> bytes:
    name = read.get("name", "r").encode() + b"\0"
    cigar = read.get("cigar", [])
    seq_len = sum(n for op, n in cigar if op in "MIS=X")
    ref_id = ref_ids.get(read.get("chrom"), -1) if read.get("chrom") is not None else -1
    body = struct.pack(
        "<iiBBHHHiiii", ref_id, read.get("pos", -1), len(name), read.get("mapq", 255), 4680, len(cigar), read.get("flag", 0),
        seq_len, -1, -1, 0,
    )
    body += name
    for op, n in cigar:
        body += struct.pack("<I", (n << 4) | CIGAR_OPS.index(op))
    body += b"\x11" * ((seq_len + 1) // 2) + b"\xff" * seq_len
    for tag, (ty, val) in read.get("tags", {}).items():
        body += tag.encode() + ty.encode()
        body += {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f", "A": "<c"}.get(ty) and struct.pack(
            {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f", "A": "<c"}[ty], val
        ) if ty != "Z" else val.encode() + b"\0"
    return struct.pack("<I", len(body)) + body


def write_bam(path: str, refs, reads, block_bytes: int = 30000) -> None:
    """refs: [(name, length)]; reads: dicts with chrom, pos (0-based), flag, mapq, cigar [(op, n)], tags {TAG: (type, value)}."""
    ref_ids = {name: k for k, (name, _) in enumerate(refs)}
    text = "@HD\tVN:1.6\tSO:unsorted\n" + "".join(f"@SQ\tSN:{n}\tLN:{ln}\n" for n, ln in refs)
    raw = b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(refs))
    for name, ln in refs:
        raw += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", ln)
    raw += b"".join(_record(r, ref_ids) for r in reads)
    with open(path, "wb") as fh:
        for k in range(0, len(raw), block_bytes):  # records straddle block boundaries on purpose
            fh.write(_bgzf_block(raw[k : k + block_bytes]))
        fh.write(_bgzf_block(b""))  # the BGZF EOF marker


def expected_split(reads, protocol: str, read_lengths=None):
    """Restatement of bam.py:73-135 on the same read records."""
    alignments = defaultdict(lambda: defaultdict(Counter))
    counts = defaultdict(int)
    stats = dict(total=0, valid=0, qcfail=0, duplicate=0, secondary=0, unmapped=0, multi=0)
    for r in reads:
        stats["total"] += 1
        flag = r.get("flag", 0)
        if flag & 0x200:
            stats["qcfail"] += 1
            continue
        if flag & 0x400:
            stats["duplicate"] += 1
            continue
        if flag & 0x100:
            stats["secondary"] += 1
            continue
        if flag & 0x4:
            stats["unmapped"] += 1
            continue
        tags = r.get("tags", {})
        if "NH" in tags:
            uniq = tags["NH"][1] == 1
        elif r.get("mapq", 255) == 255:
            uniq = True
        elif r.get("mapq", 255) < 1 or flag in SAM_NOT_UNIQ_FLAGS:
            uniq = False
        else:
            uniq = None  # "unable to determine": `not None` is True, the read is dropped as multi
        if not uniq:
            stats["multi"] += 1
            continue
        refpos, positions = r["pos"], []
        for op, n in r.get("cigar", []):
            if op in "M=X":
                positions.extend(range(refpos, refpos + n))
                refpos += n
            elif op in "DN":
                refpos += n
        length = len(positions)
        if not positions or r.get("chrom") is None:
            continue
        if read_lengths is not None and length not in read_lengths:
            continue
        map_strand = "-" if flag & 0x10 else "+"
        if protocol == "forward":
            strand, pos = (("+", positions[0]) if map_strand == "+" else ("-", positions[-1]))
        else:
            strand, pos = (("-", positions[-1]) if map_strand == "+" else ("+", positions[0]))
        alignments[length][strand][(r["chrom"], pos + 1)] += 1
        counts[length] += 1
        stats["valid"] += 1
    return alignments, counts, stats
