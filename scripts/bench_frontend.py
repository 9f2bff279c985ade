"""The front end of `detect-orfs` stage by stage on a synthetic yeast-sized
sample (run on the GPU box): BAM (written here, BGZF) -> rp_bam_split_host -> metagene profiles on
the device -> P-site offsets -> merged columns -> WIG -> export_orf_coverages.
usage: python scripts/bench_frontend.py [n_reads] [n_orfs]"""
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ribotricer_amd import alignments as al  # noqa: E402
from ribotricer_amd import detect_orfs as d  # noqa: E402
from ribotricer_amd import metagene as mg  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
n_orfs = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000
rng = np.random.default_rng(3)
chroms = [(f"chr{k}", 1_000_000) for k in range(1, 17)]
tmp = tempfile.mkdtemp()

# candidate-ORF index: annotated single-exon ORFs of 300-3000 nt, 100 nt apart, both strands
lines = ["ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n"]
cursor = {c: 2000 for c, _ in chroms}
orfs = []
for i in range(n_orfs):
    c = chroms[i % 16][0]
    ln = 3 * int(rng.integers(100, 1000))
    s = cursor[c]
    if s + ln + 500 > 1_000_000:
        continue
    cursor[c] = s + ln + 100
    strand = "+-"[(i // 16) % 2]
    orfs.append((c, s, s + ln - 1, strand))
    lines.append(f"x\tannotated\tT{i}\tprotein_coding\tG{i}\tN{i}\tprotein_coding\t{c}\t{strand}\tATG\t{s}-{s + ln - 1}\n")
index_path = os.path.join(tmp, "candidate_orfs.tsv")
open(index_path, "w").write("".join(lines))

# reads: 5' ends 12 nt upstream of codon starts (periodic), lengths 27-31, on the ORF's strand
t0 = time.perf_counter()
pick = rng.integers(0, len(orfs), n_reads)
length = rng.choice([27, 28, 29, 30, 31], n_reads, p=[0.1, 0.3, 0.3, 0.2, 0.1])
ref_ids = {c: k for k, (c, _) in enumerate(chroms)}
recs = []
name = b"r\0"
for j in range(n_reads):
    c, s, e, strand = orfs[pick[j]]
    codon = 3 * int(rng.integers(0, (e - s + 1) // 3))
    L = int(length[j])
    if strand == "+":
        pos0 = s + codon - 12 - 1  # 0-based leftmost
        flag = 0
    else:
        pos0 = e - codon + 12 - L  # the read's 5' end is its rightmost base
        flag = 16
    body = struct.pack("<iiBBHHHiiii", ref_ids[c], max(pos0, 0), 2, 255, 4680, 1, flag, L, -1, -1, 0) + name
    body += struct.pack("<I", (L << 4) | 0) + b"\x11" * ((L + 1) // 2) + b"\xff" * L + b"NHC\x01"
    recs.append(struct.pack("<I", len(body)) + body)
text = "@HD\tVN:1.6\n" + "".join(f"@SQ\tSN:{c}\tLN:{ln}\n" for c, ln in chroms)
raw = b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(chroms))
for c, ln in chroms:
    raw += struct.pack("<I", len(c) + 1) + c.encode() + b"\0" + struct.pack("<I", ln)
raw += b"".join(recs)
bam = os.path.join(tmp, "sample.bam")
with open(bam, "wb") as fh:
    for k in range(0, len(raw), 65000):
        blk = raw[k:k + 65000]
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        cd = comp.compress(blk) + comp.flush()
        fh.write(struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + struct.pack("<BBHH", 66, 67, 2, len(cd) + 25) + cd
                 + struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk)))
    fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
t_make = time.perf_counter() - t0
bam_bytes = os.path.getsize(bam)
del recs, raw

import torch  # noqa: E402

torch.zeros(1, device="cuda")
prefix = os.path.join(tmp, "out")
T = {}


def lap(name, t):
    torch.cuda.synchronize()
    T[name] = round(time.perf_counter() - t, 4)
    return time.perf_counter()


t = time.perf_counter()
annotated = mg.annotated_records(index_path); t = lap("annotated_records", t)
cols, read_length_counts = al.split_bam(bam, "forward", prefix, None); t = lap("split_bam_native", t)
metagenes = mg.metagene_coverage(annotated, cols, read_length_counts, prefix, meta_min_reads=1000); t = lap("metagene_coverage_device", t)
offsets = mg.align_metagenes(metagenes, read_length_counts, prefix, 0.428571428571, True); t = lap("align_metagenes", t)
merged = al.merge_read_lengths(cols, offsets); t = lap("merge_read_lengths", t)
d.export_wig(merged, prefix); t = lap("export_wig", t)
d.export_orf_coverages(index_path, merged, prefix); t = lap("export_orf_coverages_default", t)
rows = sum(1 for _ in open(prefix + "_translating_ORFs.tsv")) - 1
print(json.dumps({"n_reads": n_reads, "n_orfs": len(orfs), "bam_bytes": bam_bytes, "uncompressed_bam_MB_per_s": None,
                  "bam_written_in_s": round(t_make, 1), "seconds": T, "total_s": round(sum(T.values()), 3),
                  "reads_per_s_split_bam": round(n_reads / T["split_bam_native"]), "psite_offsets": {str(k): int(v) for k, v in dict(offsets).items()},
                  "translating_rows": rows}))
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
