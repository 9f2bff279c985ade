"""Seeded generator of small ribotricer indexes + merged alignments that lean on the corners of the index format
(orf.py:122-182) and of orf_coverage (detect_orfs.py:134-203): abutting / overlapping / nested / duplicated exons, reversed and
shuffled coordinate lists, 1-2 nt exons, L % 3 != 0, ORFs shorter than a codon, start codons of 0-4 characters, strands
other than + / -, chromosomes and strands without reads, counts from 1 to ~900, and MALFORMED-BUT-PARSABLE coordinates:
blocks whose end lies below their start (the reference: ORF id with the raw sum, no positions from that block),
numbers with a sign, leading zeros or blanks around them (Python's int() takes them).

Data only: used by make_golden.py (G12), by the container-only checks check_export_vs_reference.py /
check_integration_vs_reference.py, and by tests that compare the HIP path with the host path on fresh indexes.
Nothing here imports the reference."""

from __future__ import annotations

from collections import Counter, defaultdict

import numpy as np

HEADER = "\t".join(["ORF_ID", "ORF_type", "transcript_id", "transcript_type", "gene_id", "gene_name", "gene_type", "chrom",
                    "strand", "start_codon", "coordinate"])
ORF_TYPES = ["annotated", "super_uORF", "super_dORF", "uORF", "dORF", "overlap_uORF", "overlap_dORF", "novel"]


def _number(rng, value: int, dress: bool) -> str:
    """`value` as index text; `dress`: one of the spellings int() accepts beyond the canonical one."""
    if not dress:
        return str(value)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return "+" + str(value)
    if kind == 1:
        return "00" + str(value)
    if kind == 2:
        return " " + str(value)
    return str(value) + " "


def random_index(n_orfs: int, seed: int, malformed: float = 0.04, dressed: float = 0.0, crlf: bool = False,
                 final_newline: bool = True):
    """``(index_text, merged)``: the text of a candidate_orfs.tsv (header included) and ``strand -> Counter{(chrom, pos):
    count}`` as merge_read_lengths returns it (detect_orfs.py:54-83).

    ``malformed``: fraction of lines with at least one block whose end < start.  ``dressed``: fraction of lines whose
    numbers carry a sign / leading zeros / blanks.  ``crlf``: lines end in \\r\\n (int() strips it off the last block)."""
    rng = np.random.default_rng(seed)
    chroms = ["chrI", "chrII", "chrM", "scaffold_17", "2"]
    strands = ["+", "-", "+", "-", "+", "-", "."]
    lines = []
    merged = defaultdict(Counter)
    for k in range(n_orfs):
        otype = "annotated" if k < n_orfs // 4 else ORF_TYPES[int(rng.integers(0, 8))]
        chrom = chroms[int(rng.integers(0, len(chroms)))]
        strand = strands[int(rng.integers(0, len(strands)))]
        shape = int(rng.integers(0, 12))
        if shape == 0:
            total = int(rng.integers(1, 7))  # around one codon
        elif shape == 1:
            total = 3 * int(rng.integers(2, 40)) + int(rng.integers(1, 3))  # incomplete CDS
        else:
            total = 3 * int(rng.integers(2, 140))
        n_exons = min(total, int(rng.integers(1, 7)))
        cuts = np.sort(rng.choice(np.arange(1, total), size=n_exons - 1, replace=False)) if n_exons > 1 else np.zeros(0, int)
        sizes = np.diff(np.concatenate([[0], cuts, [total]])).astype(int)
        if shape == 2 and n_exons > 2:  # force 1-2 nt exons in the middle
            sizes[1] = 1 + int(rng.integers(0, 2))
        pos = int(rng.integers(1, 60000))
        exons = []
        for s in sizes:
            exons.append([pos, pos + int(s) - 1])
            gap = int(rng.integers(0, 5))
            pos += int(s) + (0 if gap == 0 else 1 if gap == 1 else int(rng.integers(2, 300)))  # abutting, 1-nt intron, ordinary
        twist = int(rng.integers(0, 16))
        if twist == 0 and len(exons) > 1:  # overlapping exons: the shared positions appear twice in the profile
            exons[1][0] = max(exons[0][0], exons[1][0] - int(rng.integers(1, 6)))
        elif twist == 1:  # a duplicated exon
            exons.append(list(exons[int(rng.integers(0, len(exons)))]))
        elif twist == 2 and len(exons) > 1:  # two exons with one start: the sort is stable (orf.py:100)
            exons[1][0] = exons[0][0]
        elif twist == 3 and exons[-1][1] - exons[-1][0] >= 4:  # an exon nested in the last one: the ORF's "end" is not its highest position
            exons.append([exons[-1][0] + 1, exons[-1][1] - 2])
        if rng.random() < malformed:
            which = int(rng.integers(0, 3))
            a = int(rng.integers(10, 60000))
            bad = [a, a - int(rng.integers(1, 9))]  # end < start
            if which == 0:
                exons = [bad]  # nothing but: a row of length 0
            elif which == 1:
                exons.append(bad)
            else:
                exons.insert(0, bad)
        # reads: a quarter of the ORFs stay empty; the rest framed or flat at several depths
        mode = int(rng.integers(0, 4))
        if mode:
            lam = [0.02, 0.1, 0.5, 3.0, 40.0, 300.0][int(rng.integers(0, 6))]
            w = np.array([2.2, 0.4, 0.4]) if mode == 1 else np.array([1.0, 1.0, 1.0]) if mode == 2 else np.array([0.3, 0.3, 2.4])
            tpos = [p for a, b in sorted(exons, key=lambda e: e[0]) for p in range(a, b + 1)]
            if strand == "-":
                tpos = tpos[::-1]
            vals = rng.poisson(lam * w[np.arange(len(tpos)) % 3]) if tpos else []
            reads_on = strand if rng.random() < 0.97 else ("-" if strand == "+" else "+")  # now and then the other strand's
            for p, v in zip(tpos, vals):
                if v > 0:
                    merged[reads_on][(chrom, int(p))] += int(v)
        order = int(rng.integers(0, 6))
        listed = exons[::-1] if order == 0 else [exons[i] for i in rng.permutation(len(exons))] if order == 1 else exons
        dress = rng.random() < dressed
        coord = ",".join(f"{_number(rng, a, dress)}-{_number(rng, b, dress)}" for a, b in listed)
        start_codon = ["ATG", "CTG", "GTG", "TTG", "AT", "A", "", "ATGC"][int(rng.integers(0, 8))]
        tid = f"tx{k:05d}.{int(rng.integers(1, 4))}"
        lines.append("\t".join([f"{tid}_ignored", otype, tid, "protein_coding" if k % 4 else "lncRNA", f"gene{k // 2:05d}",
                                f"GN-{k // 2}", "protein_coding", chrom, strand, start_codon, coord]))
    # reads nobody looks up: off the ORFs, on a chromosome without ORFs
    for _ in range(max(4, n_orfs // 20)):
        merged["+"][("chrUn", int(rng.integers(1, 1000)))] += int(rng.integers(1, 50))
        merged["-"][(chroms[int(rng.integers(0, len(chroms)))], int(rng.integers(70000, 90000)))] += int(rng.integers(1, 50))
    eol = "\r\n" if crlf else "\n"
    text = HEADER + eol + eol.join(lines) + (eol if final_newline else "")
    return text, merged


def _open_text(path: str, mode: str):
    if path.endswith(".gz"):
        import gzip

        if "w" in mode:  # (no file name, no time stamp in the header: the same bytes on every run)
            import io

            return io.TextIOWrapper(gzip.GzipFile(filename="", fileobj=open(path, "wb"), mode="wb", mtime=0), newline="")
        return gzip.open(path, "rt", newline="")
    return open(path, mode, newline="")


def write_alignments(path: str, merged) -> None:
    """The merged alignments as the (strand, chrom, pos, count) table the fixtures keep (.gz: compressed)."""
    with _open_text(path, "w") as fh:
        fh.write("strand\tchrom\tpos\tcount\n")
        for strand in sorted(merged):
            for (chrom, p), c in sorted(merged[strand].items()):
                fh.write(f"{strand}\t{chrom}\t{p}\t{c}\n")


def read_alignments(path: str):
    merged = defaultdict(Counter)
    with _open_text(path, "r") as fh:
        fh.readline()
        for line in fh:
            strand, chrom, pos, count = line.rstrip("\n").split("\t")
            merged[strand][(chrom, int(pos))] = int(count)
    return merged


PARAM_SETS = {
    "default": dict(),
    "report_all": dict(report_all=True),
    "strict": dict(report_all=True, phase_score_cutoff=0.3, min_valid_codons=8, min_reads_per_codon=1,
                   min_valid_codons_ratio=0.75, min_density_over_orf=1.0),
}
