"""Container-only cross-check (needs /root/reference, which does not travel; not collected by pytest):
INTEGRATION.md section 2, VERBATIM -- eight names of the reference's own modules rebound to this package -- and then the
REFERENCE's orchestrator `ribotricer.detect_orfs.detect_orfs()` (detect_orfs.py:354-526) run on a BAM holding the G7
reads: its six outputs must equal tests/golden/g7_expected_* (which the unpatched reference wrote) byte for byte, and the
reference's two plot functions must accept the package's return types.  Then, function by function against the reference on random
inputs: align_metagenes, export_wig, merge_read_lengths, metagene_coverage.  GPU-less backend here.

usage: python tests/golden/check_integration_vs_reference.py       exit 0 = all identical"""
import os
import sys
import tempfile
import types
from collections import Counter, defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "tools"))
sys.path.insert(0, os.environ.get("RIBOTRICER_REFERENCE", "/root/reference"))
os.environ["RIBOTRICER_AMD_BACKEND"] = "cpu"
os.environ.setdefault("MPLBACKEND", "Agg")

# pysam / quicksect are not installed in this image; the orchestrator only touches them through split_bam (rebound
# below) and through the interval tree of annotated genes that infer_protocol would query (skipped: protocol given)
for _m in ("pysam", "quicksect"):
    sys.modules.setdefault(_m, types.ModuleType(_m))


class _Tree:
    def insert(self, *_):
        pass


sys.modules["quicksect"].Interval = lambda *a: a
sys.modules["quicksect"].IntervalTree = _Tree

# ---- INTEGRATION.md section 2, as printed there -------------------------------------------------------------------
import ribotricer.detect_orfs as ref  # noqa: E402
import ribotricer.statistics as ref_stats  # noqa: E402
import ribotricer_amd.detect_orfs as amd  # noqa: E402
import ribotricer_amd.statistics as amd_stats  # noqa: E402
import ribotricer_amd.alignments as amd_al  # noqa: E402
import ribotricer_amd.metagene as amd_mg  # noqa: E402

REFERENCE_EXPORT_WIG = ref.export_wig  # (the reference's own function objects, kept for the comparisons further down)
REFERENCE_MERGE = ref.merge_read_lengths
REFERENCE_EXPORT = ref.export_orf_coverages
ref.export_orf_coverages = amd.export_orf_coverages   # the hot loop -> one GPU launch
ref.merge_read_lengths = amd_al.merge_read_lengths     # columns instead of Counter arithmetic
ref.export_wig = amd.export_wig
ref.split_bam = amd_al.split_bam                       # native BAM reader (no pysam)
ref.metagene_coverage = amd_mg.metagene_coverage       # metagene profiles on the GPU
ref.align_metagenes = amd_mg.align_metagenes
ref.phasescore = amd_stats.phasescore                  # name imported at detect_orfs.py:37
ref_stats.phasescore = amd_stats.phasescore            # metagene.py:243-244 float profiles
# -------------------------------------------------------------------------------------------------------------------

from bamwriter import write_bam  # noqa: E402

import json  # noqa: E402

params = json.load(open(os.path.join(HERE, "g7_params.json")))
nested = defaultdict(lambda: defaultdict(Counter))
with open(os.path.join(HERE, "g7_alignments.tsv")) as fh:
    fh.readline()
    for line in fh:
        length, strand, chrom, pos, count = line.rstrip("\n").split("\t")
        nested[int(length)][strand][(chrom, int(pos))] = int(count)
rng = np.random.default_rng(3)
reads = []
for length in sorted(nested):
    for strand in nested[length]:
        for (chrom, pos), count in nested[length][strand].items():
            for _ in range(count):  # forward protocol: '+' reads start at pos, '-' reads END at pos
                reads.append(dict(name="r", chrom=chrom, pos=pos - 1 if strand == "+" else pos - length, flag=0 if strand == "+" else 16,
                                  mapq=255, cigar=[("M", length)], tags={"NH": ("C", 1)} if rng.random() < 0.5 else {}))
# lengths are met in the order the reference's generator met them (its dict order decides the line order of the offsets report)
lead = []
for length in [int(k) for k in params["psite_offsets"]]:
    k = next(i for i, r in enumerate(reads) if r["cigar"][0][1] == length)
    lead.append(reads.pop(k))
order = rng.permutation(len(reads))
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    bam = os.path.join(tmp, "g7.bam")
    write_bam(bam, [("chrI", 400000), ("chrII", 400000), ("chrM", 400000)], lead + [reads[i] for i in order])
    prefix = os.path.join(tmp, "out", "g7")
    # the reference's orchestrator, its own argument order (detect_orfs.py:354-369)
    ref.detect_orfs(bam, os.path.join(HERE, "g6_index.tsv"), prefix, "forward", None, None, 0.428571428571, 5, 0, 0, 0.0, True,
                    params["meta_min_reads"])
    for name in ("metagene_profiles_5p.tsv", "metagene_profiles_3p.tsv", "psite_offsets.txt", "pos.wig", "neg.wig", "translating_ORFs.tsv"):
        got = open(f"{prefix}_{name}", "rb").read()
        want = open(os.path.join(HERE, f"g7_expected_{name}"), "rb").read()
        same = got == want
        bad += not same
        print(f"{'ok  ' if same else 'DIFF'} {name}: {len(got)} bytes")
    for name in ("read_length_dist.pdf", "metagene_plots.pdf"):  # the reference's plots took the package's return types
        ok = os.path.getsize(f"{prefix}_{name}") > 0
        bad += not ok
        print(f"{'ok  ' if ok else 'MISSING'} {name}")
# ---- align_metagenes (metagene.py:268-328) on random inputs: offsets, report text, what is left in both dicts, and the
# exception type when the reference raises (no periodic length; no length with reads; empty profiles)
import importlib  # noqa: E402
from collections import OrderedDict  # noqa: E402

import pandas as pd  # noqa: E402

ref_align = importlib.import_module("ribotricer.metagene").align_metagenes
rng = np.random.default_rng(5)
diffs = 0
with tempfile.TemporaryDirectory() as tmp:
    for trial in range(2000):
        metagenes, reads = OrderedDict(), {}
        for length in rng.choice(np.arange(1, 45), size=int(rng.integers(1, 7)), replace=False).tolist():
            n5 = int(rng.integers(1, 130)) if trial % 3 else int(rng.integers(1, 12))
            kind = int(rng.integers(0, 4))
            v = rng.random(n5) * (np.array([3.0, 0.4, 0.3])[np.arange(n5) % 3] if kind else 1.0)
            if kind == 2:
                v = np.round(v * 2)  # ties and zeros
            if kind == 3:
                v[:] = 0.0
            score = float(rng.random()) if rng.random() > 0.05 else float("nan")
            metagenes[length] = (pd.Series(v, index=np.arange(-20, n5 - 20)), pd.Series(v[::-1]), np.float64(score), 3, np.float64(0.5), 3)
            reads[length] = int(rng.integers(0, 4 if trial % 7 == 0 else 100000))
        for remove in (False, True):
            outs = []
            for tag, fn in (("r", ref_align), ("a", amd_mg.align_metagenes)):
                m, r = OrderedDict(metagenes), dict(reads)
                try:
                    offsets = fn(m, r, os.path.join(tmp, tag), 0.428571428571, remove)
                    outs.append((list(offsets.items()), open(os.path.join(tmp, tag + "_psite_offsets.txt")).read(), list(m), r))
                except BaseException as e:  # noqa: BLE001  (sys.exit included)
                    outs.append(type(e).__name__)
            diffs += outs[0] != outs[1]
print(f"align_metagenes on 4000 random inputs: {diffs} differences")
bad += diffs
# ---- export_wig (detect_orfs.py:327-352) on random mappings: strand keys beyond '+' / '-' (every strand that is not '+' lands in
# _neg.wig, the last one staying, an empty table leaving an empty file), a chromosome named "" (no header of its own), counts past
# the packed sort key (the column-sort road)
wig_diffs = 0
with tempfile.TemporaryDirectory() as tmp:
    for trial in range(600):
        mapping = defaultdict(Counter)
        keys = [["+", "-"], ["-", "+"], ["+", ".", "-"], [".", "+"], ["-", "."], ["*", "-", "+", "."], ["."], ["+"], ["-"], []][trial % 10]
        for k in keys:
            for _ in range(int(rng.integers(0, 30))):
                mapping[k][(str(rng.choice(["chr1", "chr10", "chr2", ""])), int(rng.integers(1, 500)))] += int(rng.integers(1, 9))
            mapping[k]  # (the key exists even over an empty table)
        if trial % 5 == 0 and "+" in keys:
            mapping["+"][("chr2", 7)] += 1 << 23
        outs = []
        for tag, fn in (("r", REFERENCE_EXPORT_WIG), ("a", amd.export_wig)):
            d_out = os.path.join(tmp, f"{tag}{trial}")
            os.makedirs(d_out)
            fn(mapping, os.path.join(d_out, "w"))
            outs.append({f: open(os.path.join(d_out, f)).read() for f in sorted(os.listdir(d_out))})
        wig_diffs += outs[0] != outs[1]
print(f"export_wig on 600 random mappings: {wig_diffs} differences")
bad += wig_diffs
# ---- merge_read_lengths (detect_orfs.py:54-83) on random nested mappings: lengths without an offset, offsets without a length,
# strand keys beyond '+' / '-' (only '+' moves downstream), negative offsets; the columnar result read back as Counters
merge_diffs = 0
for trial in range(1000):
    nested = defaultdict(lambda: defaultdict(Counter))
    lengths = rng.choice(np.arange(20, 40), size=int(rng.integers(0, 6)), replace=False).tolist()
    for length in lengths:
        for strand in [["+", "-"], ["+"], ["-", "."], ["+", "-", "*"]][trial % 4]:
            for _ in range(int(rng.integers(0, 25))):
                nested[length][strand][(str(rng.choice(["chr1", "chrM", "2"])), int(rng.integers(1, 60)))] += int(rng.integers(1, 5))
    offsets_of = OrderedDict((int(length), int(rng.integers(-3, 16))) for length in (lengths + [99] if trial % 3 == 0 else lengths) if rng.random() < 0.8)
    want = {k: dict(v) for k, v in REFERENCE_MERGE(nested, offsets_of).items() if v}
    got = {k: dict(v) for k, v in amd_al.merge_read_lengths(nested, offsets_of).as_counters().items() if v}
    merge_diffs += want != got
print(f"merge_read_lengths on 1000 random inputs: {merge_diffs} differences")
bad += merge_diffs

# ---- metagene_coverage (metagene.py:160-265) on corner-case indexes (random_index.py: overlapping / nested exons, a '.' strand,
# blocks with end < start) x random per-length alignments x window sizes / offsets / read thresholds: both profile files byte for
# byte, what is left in read_lengths, the Series' index ranges, valid codons
sys.path.insert(0, HERE)
from random_index import random_index  # noqa: E402

ref_mg = importlib.import_module("ribotricer.metagene")
ref_orf = importlib.import_module("ribotricer.orf").ORF
mg_diffs = 0
with tempfile.TemporaryDirectory() as tmp:
    for trial in range(45):
        shape = [dict(malformed=0.0), dict(malformed=0.1), dict(malformed=0.0, dressed=0.2)][trial % 3]
        text, merged = random_index(int(rng.integers(8, 120)), 5000 + trial, **shape)
        path = os.path.join(tmp, f"i{trial}.tsv")
        with open(path, "w", newline="") as fh:
            fh.write(text)
        cds_ref = []
        with open(path) as fh:
            fh.readline()
            for line in fh:
                if "annotated" not in line:
                    break
                orf = ref_orf.from_string(line)
                if orf is not None and orf.category == "annotated":
                    cds_ref.append(orf)
        nested = defaultdict(lambda: defaultdict(Counter))
        reads = defaultdict(int)
        for strand, table in merged.items():
            for (chrom, pos), count in table.items():
                for _ in range(min(count, 6)):
                    length = int(rng.choice([27, 28, 29, 30]))
                    nested[length][strand][(chrom, int(pos + rng.integers(-15, 16)))] += 1
                    reads[length] += 1
        meta_min, window = int(rng.choice([1, 50, 10**9])), int(rng.choice([600, 30, 7]))
        o5, o3 = [(20, 0), (0, 0), (5, 7), (50, 50)][trial % 4]
        outs = []
        for tag, fn, cds in (("r", ref_mg.metagene_coverage, cds_ref), ("a", amd_mg.metagene_coverage, amd_mg.annotated_records(path))):
            left = dict(reads)
            try:
                m = fn(cds, nested, left, os.path.join(tmp, tag), max_positions=window, offset_5p=o5, offset_3p=o3, meta_min_reads=meta_min)
                files = tuple(open(os.path.join(tmp, f"{tag}_metagene_profiles_{side}.tsv")).read() for side in ("5p", "3p"))
                outs.append((sorted(left.items()), {k: (list(v[0].index), list(v[1].index), int(v[3]), int(v[5])) for k, v in m.items()}, files))
            except BaseException as e:  # noqa: BLE001
                outs.append(type(e).__name__)
        if outs[0] != outs[1]:
            mg_diffs += 1
            print(f"  metagene_coverage differs: trial {trial} {shape} window {window} offsets {o5}/{o3} meta_min {meta_min}: "
                  + (f"{outs[0]} vs {outs[1]}" if isinstance(outs[0], str) or isinstance(outs[1], str) else
                     "; ".join(name for name, x, y in zip(("read_lengths", "series", "files"), outs[0], outs[1]) if x != y)))
print(f"metagene_coverage on 45 corner-case indexes: {mg_diffs} differences")
bad += mg_diffs
# ---- the whole chain behind split_bam, function for function: the reference's own metagene_coverage -> align_metagenes ->
# merge_read_lengths -> export_wig -> export_orf_coverages against this package's, on corner-case indexes with periodic reads of
# several lengths: six files each, byte for byte (and the case where no read length is periodic: both must exit the same way)
ref_pure = importlib.import_module("ribotricer.detect_orfs")
chain_diffs = 0
with tempfile.TemporaryDirectory() as tmp:
    for trial in range(16):
        text, merged = random_index(int(rng.integers(30, 160)), 8000 + trial, malformed=0.05 * (trial % 3))
        path = os.path.join(tmp, f"c{trial}.tsv")
        with open(path, "w", newline="") as fh:
            fh.write(text)
        nested = defaultdict(lambda: defaultdict(Counter))
        reads = defaultdict(int)
        true_offset = {27: 11, 28: 12, 29: 12, 30: 13}
        for strand, table in merged.items():
            for (chrom, pos), count in table.items():
                for _ in range(min(count, 8)):
                    length = int(rng.choice([27, 28, 29, 30], p=[0.1, 0.4, 0.3, 0.2]))
                    five = pos - true_offset[length] if strand == "+" else pos + true_offset[length]
                    nested[length][strand][(chrom, int(five))] += 1
                    reads[length] += 1
        cds_ref = []
        with open(path) as fh:
            fh.readline()
            for line in fh:
                if "annotated" not in line:
                    break
                orf = ref_orf.from_string(line)
                if orf is not None and orf.category == "annotated":
                    cds_ref.append(orf)
        outs = []
        for tag in ("r", "a"):
            prefix = os.path.join(tmp, f"{tag}{trial}")
            left = dict(reads)
            try:
                if tag == "r":
                    m = ref_mg.metagene_coverage(cds_ref, nested, left, prefix, meta_min_reads=20)
                    offs = ref_mg.align_metagenes(m, left, prefix, 0.428571428571, trial % 4 != 3)
                    merged_now = REFERENCE_MERGE(nested, offs)
                    REFERENCE_EXPORT_WIG(merged_now, prefix)
                    REFERENCE_EXPORT(path, merged_now, prefix, report_all=bool(trial % 2))
                else:
                    m = amd_mg.metagene_coverage(amd_mg.annotated_records(path), nested, left, prefix, meta_min_reads=20)
                    offs = amd_mg.align_metagenes(m, left, prefix, 0.428571428571, trial % 4 != 3)
                    merged_now = amd_al.merge_read_lengths(nested, offs)
                    amd.export_wig(merged_now, prefix)
                    amd.export_orf_coverages(path, merged_now, prefix, report_all=bool(trial % 2))
                files = {f: open(os.path.join(tmp, f), "rb").read() for f in sorted(os.listdir(tmp)) if f.startswith(f"{tag}{trial}_")}
                outs.append((list(offs.items()), {k[len(f"{tag}{trial}"):]: v for k, v in files.items()}))
            except BaseException as e:  # noqa: BLE001
                outs.append(type(e).__name__)
        if outs[0] != outs[1]:
            chain_diffs += 1
            what = "exception" if isinstance(outs[0], str) or isinstance(outs[1], str) else [k for k in outs[0][1] if outs[0][1].get(k) != outs[1][1].get(k)] or "offsets"
            print(f"  chain differs: trial {trial}: {what}")
print(f"the chain behind split_bam on 16 corner-case samples: {chain_diffs} differences")
bad += chain_diffs
print("integration check:", "all identical" if not bad else f"{bad} differences")
sys.exit(1 if bad else 0)
