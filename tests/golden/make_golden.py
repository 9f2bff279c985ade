#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE ITSELF.

Run only in the build container, where the reference is mounted read-only:

    python tests/golden/make_golden.py            # writes tests/golden/*.npz|json|tsv

It imports ``ribotricer.statistics.phasescore`` and (with ``pysam``/``quicksect``
stubbed in ``sys.modules`` because they are module-level imports of packages that
are not installed here) ``ribotricer.detect_orfs.export_orf_coverages`` from
/root/reference and records their outputs on seeded inputs.  Only inputs and
expected outputs are written; no reference source is copied.  The GPU box has no
/root/reference, so tests read these files and never this import path.

Fixture sets:
  g1_known_answers.json  quirk / state-machine vectors (Appendix A.5)
  g2_poisson.npz         ~4900 Poisson ORFs, lambda 0.003..5, L 3..600
  g3_adversarial.npz     6000 short vectors over {0,1,2}, L 0..24 (ties, resets)
  g4_long.npz            long ORFs up to 99 999 nt
  g5_float.json          float (metagene-like) profiles
  g6_*                   end-to-end: index + alignments + the reference's TSVs
  g7_*                   the front end in miniature: per-read-length alignments ->
                         metagene profiles, P-site offsets, merged alignments, WIG, TSV
  g8_bigties.npz         3200 sparse profiles with counts 16..1000 (some to 3e6) built to tie
  g8_float_ties.json     400 float-valued profiles built to tie
  g10_bigcounts.npz      240 profiles holding counts 2^24 .. 2^30 (beyond the fp32-exact range of the kernels)
  g10_*                  end-to-end: the g6 index + its alignments with such counts piled on 16 ORFs + the reference's TSVs
  g12_*                  end-to-end on a 1 500-line index from random_index.py (format corners, malformed-but-parsable
                         lines) + alignments + the reference's three TSVs (gzip)
"""

from __future__ import annotations

import json
import os
import platform
import sys
import tempfile
import types
from collections import Counter, defaultdict

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("RIBOTRICER_REFERENCE", "/root/reference")

sys.path.insert(0, REFERENCE)
sys.path.insert(0, REPO)

from ribotricer.statistics import phasescore  # noqa: E402  (the reference)

for _m in ("pysam", "quicksect"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["quicksect"].Interval = object
sys.modules["quicksect"].IntervalTree = dict
from ribotricer.detect_orfs import export_orf_coverages  # noqa: E402  (the reference)

from oracle.phasescore_literal import combine_frames, phasescore_frames  # noqa: E402

META = {
    "generator": "tests/golden/make_golden.py",
    "reference": "smithlabcode/ribotricer v1.5.0 (ribotricer.statistics.phasescore, "
    "ribotricer.detect_orfs.export_orf_coverages)",
    "numpy": np.__version__,
    "scipy": scipy.__version__,
    "python": platform.python_version(),
    "machine": platform.processor() or platform.machine(),
}


def run_reference(vectors):
    """Reference (phase, valid) + per-frame replay through the literal restatement.

    The replay's combined result must equal the reference's bit for bit, which
    is what licenses storing its per-frame values next to the reference's output.
    """
    n = len(vectors)
    phase = np.empty(n, np.float64)
    valid = np.empty(n, np.int32)
    fscore = np.full((n, 3), np.nan, np.float64)
    fn = np.zeros((n, 3), np.int32)
    for k, vec in enumerate(vectors):
        p, v = phasescore(vec)
        frames = phasescore_frames(vec)
        p2, v2 = combine_frames(frames)
        same = (p == p2) or (np.isnan(p) and np.isnan(p2))
        if not same or v != v2:
            raise AssertionError(f"literal restatement diverges from reference on vector {k}: {(p, v)} vs {(p2, v2)}")
        phase[k] = p
        valid[k] = v
        for f, (s, nseg) in enumerate(frames):
            fn[k, f] = nseg
            if s is not None:
                fscore[k, f] = s
    return phase, valid, fscore, fn


def pack_csr(vectors):
    lens = np.array([len(v) for v in vectors], np.int64)
    offsets = np.zeros(len(vectors) + 1, np.int64)
    np.cumsum(lens, out=offsets[1:])
    counts = np.zeros(int(offsets[-1]), np.int32)
    for v, o in zip(vectors, offsets[:-1]):
        counts[o : o + len(v)] = v
    return counts, offsets


def save_csr_set(name, vectors, extra=None):
    phase, valid, fscore, fn = run_reference(vectors)
    counts, offsets = pack_csr(vectors)
    payload = dict(
        counts=counts,
        offsets=offsets,
        phase=phase,
        valid=valid,
        frame_score=fscore,
        frame_n=fn,
        meta=np.array(json.dumps(META)),
    )
    if extra:
        payload.update(extra)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **payload)
    print(f"{name}: {len(vectors)} vectors, {counts.size} nt, {os.path.getsize(path) / 1e3:.0f} kB")


# --------------------------------------------------------------------------- G1
def g1():
    vecs = [
        [],
        [1, 2],
        [0] * 9,
        [1, 0, 0],
        [0, 0, 0, 5],
        [1, 0, 0] * 2,
        [5] + [0] * 299,
        [0, 0, 0, 5] + [0] * 296,
        [1, 1, 1] * 5,
        [1, 1, 1] * 4 + [2, 0, 0],
        [1, 0, 0, 1, 1, 1],
        [1, 0, 0, 0, 1, 0, 0, 0, 1],
        [3, 0, 0] * 30,
        [0, 3, 0] * 30,
        [0, 0, 3] * 30,
        [2, 1, 0] * 10 + [4],
        [3, 1, 0, 0, 0, 0, 2, 2, 2, 0, 5, 1, 7, 0, 0, 1, 1, 1, 0, 0, 4, 1],
        [7, 0, 0, 7, 2, 0, 1, 0, 0, 5, 0, 0, 9, 0, 0, 8, 0, 0, 7, 0, 0, 5, 0, 0, 8, 0, 0, 6, 0, 0],
        [0, 0, 7, 0, 0, 3, 0, 0, 9, 0, 0, 1, 0, 0, 2, 1],
        [4, 4, 4, 0, 0, 0, 4, 4, 4, 1, 0, 0, 0, 0, 0, 2, 2, 2],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1],
        [1000000, 0, 0, 999999, 1, 0, 1000000, 0, 1, 0, 0, 0, 5, 5, 5],
        [12, 3, 1, 9, 2, 2, 15, 1, 0, 7, 7, 7, 11, 0, 3, 8],
    ]
    phase, valid, fscore, fn = run_reference(vecs)
    rows = []
    for v, p, va, fs, n in zip(vecs, phase, valid, fscore, fn):
        rows.append(
            {
                "input": v,
                "phase": float(p),
                "phase_repr": repr(float(p)),
                "valid": int(va),
                "frame_score": [None if np.isnan(x) else float(x) for x in fs],
                "frame_n": [int(x) for x in n],
            }
        )
    with open(os.path.join(HERE, "g1_known_answers.json"), "w") as fh:
        json.dump({"meta": META, "vectors": rows}, fh, indent=1)
    print(f"g1_known_answers.json: {len(rows)} vectors")


# --------------------------------------------------------------------------- G2
def g2():
    rng = np.random.default_rng(20260213)
    vecs, lam_of = [], []
    for lam in (0.003, 0.01, 0.03, 0.1, 0.3, 1.0, 5.0):
        for k in range(700):
            if k % 4 == 0:
                length = int(rng.integers(3, 601))  # any length, incl. L%3 != 0
            else:
                length = 3 * int(rng.integers(20, 201))  # prepare-orfs lengths (multiples of 3, >= 60)
            if k % 2 == 0:
                w = np.array([2.0, 0.5, 0.5])  # framed
            else:
                w = np.array([1.0, 1.0, 1.0])
            rate = lam * w[np.arange(length) % 3]
            vecs.append(rng.poisson(rate).astype(np.int64).tolist())
            lam_of.append(lam)
    save_csr_set("g2_poisson.npz", vecs, {"lam": np.array(lam_of)})


# --------------------------------------------------------------------------- G3
def g3():
    rng = np.random.default_rng(7)
    vecs = []
    for _ in range(6000):
        length = int(rng.integers(0, 25))
        p0 = rng.choice([0.5, 0.7, 0.9])
        p = [p0, (1 - p0) * 0.6, (1 - p0) * 0.4]
        vecs.append(rng.choice([0, 1, 2], size=length, p=p).astype(np.int64).tolist())
    save_csr_set("g3_adversarial.npz", vecs)


# --------------------------------------------------------------------------- G4
def g4():
    rng = np.random.default_rng(99)
    spec = [(99999, 1.0)] + [(30000, l) for l in (0.05, 1.0, 50.0)]
    spec += [(3000, l) for l in (0.05, 0.05, 0.3, 0.3, 1.0, 1.0, 5.0, 5.0, 50.0, 50.0)]
    spec += [(12288, 0.5), (8193, 0.2), (16384, 2.0), (4097, 0.02)]
    vecs = []
    for length, lam in spec:
        w = np.array([2.0, 0.5, 0.5])
        vecs.append(rng.poisson(lam * w[np.arange(length) % 3]).astype(np.int64).tolist())
    save_csr_set("g4_long.npz", vecs)


# --------------------------------------------------------------------------- G5
def g5():
    rng = np.random.default_rng(5)
    rows = []
    vecs = [[0.5, 0.0, 0.25] * 7, [0.0] * 12, [1.5, 1.5, 1.5] * 4, [0.25, 0.0, 0.0, 0.0, 0.0, 0.125]]
    for _ in range(8):
        n = 620
        base = rng.gamma(2.0, 50.0, size=n) * np.array([3.0, 1.0, 1.0])[np.arange(n) % 3]
        vecs.append(np.round(base, 3).tolist())
    for v in vecs:
        p, va = phasescore(v)
        rows.append({"input": v, "phase": float(p), "valid": int(va)})
    with open(os.path.join(HERE, "g5_float.json"), "w") as fh:
        json.dump({"meta": META, "vectors": rows}, fh)
    print(f"g5_float.json: {len(rows)} vectors")


# --------------------------------------------------------------------------- G6
ORF_TYPES = ["annotated", "super_uORF", "super_dORF", "uORF", "dORF", "overlap_uORF", "overlap_dORF", "novel"]


def g6():
    rng = np.random.default_rng(606)
    chroms = ["chrI", "chrII", "chrM"]
    lines = []
    align = defaultdict(Counter)
    n_orfs = 220
    for k in range(n_orfs):
        # annotated rows come first in a real index (prepare_orfs.py:322-329)
        otype = "annotated" if k < 60 else ORF_TYPES[1 + (k % 7)]
        chrom = chroms[k % 3]
        strand = "+" if (k // 3) % 2 == 0 else "-"
        n_exons = int(rng.integers(1, 5))
        total = 3 * int(rng.integers(20, 120))
        if otype == "annotated" and k % 9 == 0:
            total += int(rng.integers(1, 3))  # incomplete CDS, L % 3 != 0
        if k == 7:
            total = 2  # shorter than one codon
        cuts = np.sort(rng.choice(np.arange(1, total), size=min(n_exons - 1, total - 1), replace=False)) if total > 1 else []
        sizes = np.diff(np.concatenate([[0], cuts, [total]])).astype(int)
        pos = int(rng.integers(100, 200000))
        exons = []
        for s in sizes:
            exons.append((pos, pos + s - 1))
            pos += s + int(rng.integers(20, 400))
        # reads: a third of the ORFs stay empty; others framed or flat at several depths
        mode = k % 3
        if mode != 0:
            lam = [0.02, 0.1, 0.5, 3.0][k % 4]
            w = np.array([2.2, 0.4, 0.4]) if mode == 1 else np.array([1.0, 1.0, 1.0])
            tpos = [p for a, b in exons for p in range(a, b + 1)]
            if strand == "-":
                tpos = tpos[::-1]
            vals = rng.poisson(lam * w[np.arange(len(tpos)) % 3])
            for p, v in zip(tpos, vals):
                if v > 0:
                    align[strand][(chrom, p)] += int(v)
        coord = ",".join(f"{a}-{b}" for a, b in (exons if k % 5 else exons[::-1]))  # unsorted on purpose sometimes
        start_codon = ["ATG", "CTG", "GTG", "TTG", "AT"][k % 5]
        tid = f"tx{k:04d}"
        lines.append(
            "\t".join(
                [
                    f"{tid}_{exons[0][0]}_{exons[-1][1]}_{total}",
                    otype,
                    tid,
                    "protein_coding" if k % 4 else "lncRNA",
                    f"gene{k // 2:04d}",
                    f"GN{k // 2}",
                    "protein_coding",
                    chrom,
                    strand,
                    start_codon,
                    coord,
                ]
            )
        )
    header = "\t".join(
        ["ORF_ID", "ORF_type", "transcript_id", "transcript_type", "gene_id", "gene_name", "gene_type", "chrom", "strand", "start_codon", "coordinate"]
    )
    index_path = os.path.join(HERE, "g6_index.tsv")
    with open(index_path, "w") as fh:
        fh.write(header + "\n")
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(HERE, "g6_alignments.tsv"), "w") as fh:
        fh.write("strand\tchrom\tpos\tcount\n")
        for strand in sorted(align):
            for (chrom, p), c in sorted(align[strand].items()):
                fh.write(f"{strand}\t{chrom}\t{p}\t{c}\n")
    param_sets = {
        "default": dict(),
        "report_all": dict(report_all=True),
        "strict": dict(
            report_all=True,
            phase_score_cutoff=0.3,
            min_valid_codons=8,
            min_reads_per_codon=1,
            min_valid_codons_ratio=0.75,
            min_density_over_orf=1.0,
        ),
    }
    with open(os.path.join(HERE, "g6_params.json"), "w") as fh:
        json.dump({"meta": META, "param_sets": param_sets}, fh, indent=1)
    for name, kw in param_sets.items():
        with tempfile.TemporaryDirectory() as tmp:
            prefix = os.path.join(tmp, "out")
            # reference signature: detect_orfs.py:206-216
            export_orf_coverages(index_path, align, prefix, **kw)
            with open(prefix + "_translating_ORFs.tsv") as src:
                text = src.read()
        with open(os.path.join(HERE, f"g6_expected_{name}.tsv"), "w") as dst:
            dst.write(text)
        print(f"g6_expected_{name}.tsv: {text.count(chr(10)) - 1} rows")


def g7():
    """The detect-orfs front end in miniature, on the G6 index: per-read-length 5'-end
    alignments (what split_bam returns) -> the reference's metagene_coverage, align_metagenes,
    merge_read_lengths, export_wig and export_orf_coverages (detect_orfs.py:444-520)."""
    from ribotricer.detect_orfs import export_wig, merge_read_lengths
    from ribotricer.metagene import align_metagenes, metagene_coverage
    from ribotricer.orf import ORF

    rng = np.random.default_rng(707)
    index_path = os.path.join(HERE, "g6_index.tsv")
    annotated = []
    with open(index_path) as fh:
        fh.readline()
        for line in fh:
            orf = ORF.from_string(line)
            if orf is not None and orf.category == "annotated":
                annotated.append(orf)
    true_offsets = {27: 11, 28: 12, 29: 12, 30: 13, 31: 13}
    weights = {27: 0.04, 28: 0.3, 29: 0.35, 30: 0.25, 31: 0.06}  # 27 stays under meta_min_reads
    alignments = defaultdict(lambda: defaultdict(Counter))
    read_length_counts = defaultdict(int)
    for k, orf in enumerate(annotated):
        if k % 5 == 4:
            continue  # silent gene
        lam = [0.4, 1.5, 4.0][k % 3]
        tpos = [p for iv in orf.intervals for p in range(iv.start, iv.end + 1)]
        first, last = tpos[0], tpos[-1]
        tpos = list(range(first - 30, first)) + tpos + list(range(last + 1, last + 31))  # flanks: reads spill into the UTRs
        if orf.strand == "-":
            tpos = tpos[::-1]
        w = np.array([2.5, 0.3, 0.2])
        phase = (np.arange(len(tpos)) - 30) % 3
        inside = (np.arange(len(tpos)) >= 30) & (np.arange(len(tpos)) < len(tpos) - 30)
        vals = rng.poisson(np.where(inside, lam * w[phase], 0.05))
        for p, v in zip(tpos, vals):
            for _ in range(int(v)):
                length = int(rng.choice(list(weights), p=list(weights.values())))
                off = true_offsets[length]
                five = p - off if orf.strand == "+" else p + off
                alignments[length][orf.strand][(orf.chrom, five)] += 1
                read_length_counts[length] += 1
    with open(os.path.join(HERE, "g7_alignments.tsv"), "w") as fh:
        fh.write("length\tstrand\tchrom\tpos\tcount\n")
        for length in sorted(alignments):
            for strand in sorted(alignments[length]):
                for (chrom, p), c in sorted(alignments[length][strand].items()):
                    fh.write(f"{length}\t{strand}\t{chrom}\t{p}\t{c}\n")
    meta_min_reads = 400
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, "g7")
        counts_for_meta = dict(read_length_counts)  # metagene_coverage deletes the rare lengths in place
        metagenes = metagene_coverage(annotated, alignments, counts_for_meta, prefix, meta_min_reads=meta_min_reads)
        psite_offsets = align_metagenes(metagenes, counts_for_meta, prefix, 0.428571428571, True)
        merged = merge_read_lengths(alignments, psite_offsets)
        export_wig(merged, prefix)
        export_orf_coverages(index_path, merged, prefix, report_all=True)
        for name in ("metagene_profiles_5p.tsv", "metagene_profiles_3p.tsv", "psite_offsets.txt", "pos.wig", "neg.wig", "translating_ORFs.tsv"):
            with open(f"{prefix}_{name}") as src, open(os.path.join(HERE, f"g7_expected_{name}"), "w") as dst:
                dst.write(src.read())
    with open(os.path.join(HERE, "g7_merged.tsv"), "w") as fh:
        fh.write("strand\tchrom\tpos\tcount\n")
        for strand in sorted(merged):
            for (chrom, p), c in sorted(merged[strand].items()):
                fh.write(f"{strand}\t{chrom}\t{p}\t{c}\n")
    with open(os.path.join(HERE, "g7_params.json"), "w") as fh:
        json.dump({"meta": META, "read_length_counts": {str(k): int(v) for k, v in sorted(read_length_counts.items())},
                   "meta_min_reads": meta_min_reads, "psite_offsets": {str(k): int(v) for k, v in psite_offsets.items()},
                   "metagene_phase": {str(k): [float(v[2]), int(v[3]), float(v[4]), int(v[5])] for k, v in metagenes.items()}}, fh, indent=1)
    print("g7: reads per length", dict(sorted(read_length_counts.items())), "offsets", dict(psite_offsets))


# --------------------------------------------------------------------------- G8
def g8():
    """Exact frame ties that involve counts >= 16 (where the reference's `real**2 + image**2`,
    statistics.py:83, is libm pow() and not x*x), and float-valued profiles with exact ties
    (metagene.py:243-244): what rp_tie_replay_host / rp_tie_replay_f64_host must reproduce."""
    rng = np.random.default_rng(808)
    vecs = []
    for k in range(3200):
        length = int(rng.integers(6, 601)) if k % 5 == 0 else 3 * int(rng.integers(20, 201))
        v = np.zeros(length, np.int64)
        kind = k % 8
        big = lambda size=None: rng.integers(16, 1001, size=size)  # noqa: E731
        if kind == 0:  # a few piles, all on one residue class: score == 1 in two or three frames
            r = int(rng.integers(0, 3))
            at = r + 3 * rng.integers(0, max(1, (length - r + 2) // 3), size=int(rng.integers(1, 7)))
            at = at[at < length]
            v[at] = big(at.size)
        elif kind == 1:  # perfectly periodic, one or two residues occupied
            c = big(3)
            c[int(rng.integers(0, 3))] = 0
            if k % 16 == 1:
                c[int(rng.integers(0, 3))] = 0
            v[:] = np.tile(c, length // 3 + 1)[:length]
        elif kind == 2:  # the unit vectors cancel: equal piles on residues 0, 1, 2
            c = int(big())
            trip = rng.choice(max(1, length // 3), size=3, replace=length // 3 < 3)
            for r, j in enumerate(trip):
                if 3 * int(j) + r < length:
                    v[3 * int(j) + r] = c
        elif kind == 3:  # sparse Poisson scaled up: few non-zero codons, large values
            v[:] = rng.poisson(0.01, size=length) * big(length)
        elif kind == 4:  # sparse with one residue favoured
            w = np.array([1.0, 0.05, 0.05])[(np.arange(length) + int(rng.integers(0, 3))) % 3]
            v[:] = rng.poisson(0.03 * w) * big(length)
        elif kind == 5:  # two piles (the SURVEY A.4 example shape) with big counts
            at = rng.integers(0, length, size=2)
            v[at] = big(2)
        elif kind == 6:  # very large counts on one residue
            r = int(rng.integers(0, 3))
            at = r + 3 * rng.integers(0, max(1, (length - r + 2) // 3), size=int(rng.integers(1, 5)))
            at = at[at < length]
            v[at] = rng.integers(1000, 3_000_000, size=at.size)
        else:  # mixture of small and big piles
            at = rng.integers(0, length, size=int(rng.integers(1, 6)))
            v[at] = np.where(rng.random(at.size) < 0.5, big(at.size), rng.integers(1, 4, size=at.size))
        vecs.append(v.tolist())
    save_csr_set("g8_bigties.npz", vecs)
    # float profiles
    rows, fvecs = [], []
    for k in range(400):
        length = 3 * int(rng.integers(4, 80)) + int(rng.integers(0, 3)) * (k % 4 == 0)
        v = np.zeros(length)
        kind = k % 4
        if kind == 0:
            r = int(rng.integers(0, 3))
            at = r + 3 * rng.integers(0, max(1, (length - r + 2) // 3), size=int(rng.integers(1, 7)))
            at = at[at < length]
            v[at] = np.round(rng.gamma(2.0, 3.0, size=at.size), 3)
        elif kind == 1:
            c = np.round(rng.gamma(2.0, 3.0, size=3), 3)
            c[int(rng.integers(0, 3))] = 0.0
            v[:] = np.tile(c, length // 3 + 1)[:length]
        elif kind == 2:
            v[:] = (rng.random(length) < 0.03) * np.round(rng.gamma(2.0, 0.7, size=length), 4)
        else:
            c = float(np.round(rng.gamma(2.0, 3.0), 3))
            for r, j in enumerate(rng.choice(max(1, length // 3), size=3, replace=length // 3 < 3)):
                if 3 * int(j) + r < length:
                    v[3 * int(j) + r] = c
        fvecs.append(v.tolist())
    for v in fvecs:
        p, va = phasescore(v)
        rows.append({"input": v, "phase": float(p), "valid": int(va)})
    with open(os.path.join(HERE, "g8_float_ties.json"), "w") as fh:
        json.dump({"meta": META, "vectors": rows}, fh)
    print(f"g8_float_ties.json: {len(rows)} vectors")


# --------------------------------------------------------------------------- G10
def g10():
    """Counts beyond 2^24 - 1 (RP_MAX_COUNT: what converts to fp32 exactly).  The reference has no limit
    (detect_orfs.py:176-187, 278-280 work on Python ints); round 3 aborted the export on such a count."""
    rng = np.random.default_rng(1010)
    huge = lambda size=None: (1 << rng.integers(24, 31, size=size)) + rng.integers(-3, 1000, size=size)  # noqa: E731
    vecs = []
    for k in range(240):
        length = int(rng.integers(6, 601)) if k % 5 == 0 else 3 * int(rng.integers(20, 201))
        kind = k % 6
        if kind == 0:  # ordinary Poisson profile with one saturated position
            v = rng.poisson(0.5 * np.array([2.0, 0.5, 0.5])[np.arange(length) % 3]).astype(np.int64)
            v[int(rng.integers(0, length))] = int(huge())
        elif kind == 1:  # sparse, a few huge piles on one residue class (ties between frames)
            v = np.zeros(length, np.int64)
            r = int(rng.integers(0, 3))
            at = r + 3 * rng.integers(0, max(1, (length - r + 2) // 3), size=int(rng.integers(1, 5)))
            at = at[at < length]
            v[at] = huge(at.size)
        elif kind == 2:  # every codon huge: differences of huge numbers decide the direction
            base = huge(length) if k % 12 == 2 else (1 << 30) - rng.integers(0, 5000, size=length)  # (every codon sum past 2^31)
            v = (base + rng.integers(0, 50, size=length) * np.array([40, 1, 1])[np.arange(length) % 3]).astype(np.int64)
        elif kind == 3:  # equal huge piles on residues 0, 1, 2: the unit vectors cancel
            v = np.zeros(length, np.int64)
            c = int(huge())
            for r, j in enumerate(rng.choice(max(1, length // 3), size=3, replace=length // 3 < 3)):
                if 3 * int(j) + r < length:
                    v[3 * int(j) + r] = c
        elif kind == 4:  # a codon whose sum passes 2^31 (three positions of ~2^30)
            v = rng.poisson(1.0, size=length).astype(np.int64)
            j = 3 * int(rng.integers(0, max(1, length // 3)))
            v[j : j + 3] = (1 << 30) - np.array([1, 2, 3])[: v[j : j + 3].size]
        else:  # flat huge codons (a = b = c) among ordinary ones
            v = rng.poisson(0.3, size=length).astype(np.int64)
            j = 3 * int(rng.integers(0, max(1, length // 3)))
            v[j : j + 3] = int(huge())
        vecs.append(v.tolist())
    save_csr_set("g10_bigcounts.npz", vecs)
    # end to end: the g6 index and alignments, with huge counts piled on positions of 16 ORFs
    align = defaultdict(Counter)
    with open(os.path.join(HERE, "g6_alignments.tsv")) as fh:
        fh.readline()
        for line in fh:
            strand, chrom, pos, count = line.rstrip("\n").split("\t")
            align[strand][(chrom, int(pos))] = int(count)
    index_path = os.path.join(HERE, "g6_index.tsv")
    rows = [ln.rstrip("\n").split("\t") for ln in open(index_path)][1:]
    extra = []
    for k in list(range(1, 220, 14)):
        f = rows[k]
        exons = sorted(tuple(int(x) for x in g.split("-")) for g in f[10].split(","))
        tpos = [p for a, b in exons for p in range(a, b + 1)]
        if len(tpos) < 6:
            continue
        n_piles = 1 + k % 3
        for p in rng.choice(tpos, size=n_piles, replace=False):
            c = int(huge())
            if k % 4 == 1:
                c = (1 << 30) - 1
            align[f[8]][(f[7], int(p))] += c
            extra.append((f[8], f[7], int(p), c))
    with open(os.path.join(HERE, "g10_alignments.tsv"), "w") as fh:
        fh.write("strand\tchrom\tpos\tcount\n")
        for strand in sorted(align):
            for (chrom, p), c in sorted(align[strand].items()):
                fh.write(f"{strand}\t{chrom}\t{p}\t{c}\n")
    for name, kw in (("default", dict()), ("report_all", dict(report_all=True)),
                     ("strict", dict(report_all=True, phase_score_cutoff=0.3, min_valid_codons=8, min_reads_per_codon=1,
                                     min_valid_codons_ratio=0.75, min_density_over_orf=1.0))):
        with tempfile.TemporaryDirectory() as tmp:
            prefix = os.path.join(tmp, "out")
            export_orf_coverages(index_path, align, prefix, **kw)
            text = open(prefix + "_translating_ORFs.tsv").read()
        with open(os.path.join(HERE, f"g10_expected_{name}.tsv"), "w") as dst:
            dst.write(text)
        print(f"g10_expected_{name}.tsv: {text.count(chr(10)) - 1} rows, {len(extra)} piles added")


def g12():
    """A fresh index from tests/golden/random_index.py -- 1 500 lines leaning on the format's corners (abutting /
    overlapping / duplicated exons, reversed lists, 1-nt exons, L % 3 != 0, a '.' strand, blocks with end < start,
    dressed numbers) -- with its alignments and the reference's three TSVs (gzip: ~3 MB of text otherwise)."""
    import gzip

    sys.path.insert(0, HERE)
    from random_index import PARAM_SETS, random_index, write_alignments

    text, merged = random_index(1500, seed=1212, malformed=0.08, dressed=0.05)
    index_path = os.path.join(HERE, "g12_index.tsv")
    with open(index_path, "w", newline="") as fh:
        fh.write(text)
    write_alignments(os.path.join(HERE, "g12_alignments.tsv.gz"), merged)
    for name, kw in PARAM_SETS.items():
        with tempfile.TemporaryDirectory() as tmp:
            prefix = os.path.join(tmp, "out")
            export_orf_coverages(index_path, merged, prefix, **kw)  # the reference (detect_orfs.py:206-324)
            data = open(prefix + "_translating_ORFs.tsv", "rb").read()
        with open(os.path.join(HERE, f"g12_expected_{name}.tsv.gz"), "wb") as raw:
            with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as dst:
                dst.write(data)
        print(f"g12_expected_{name}.tsv.gz: {data.count(bytes([10])) - 1} rows, {len(data)} bytes of text")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g10", "g12"]
    for name in which:
        globals()[name]()
