"""Container-only cross-check (needs /root/reference, which does not travel; not collected by pytest):
the product's export_orf_coverages on the GPU-less backend against the REFERENCE's own export_orf_coverages
(detect_orfs.py:206-324) on fresh random indexes from tests/golden/random_index.py -- abutting / overlapping /
duplicated exons, reversed lists, 1-nt exons, L % 3 != 0, malformed-but-parsable lines (end < start, dressed
numbers, CRLF) -- three threshold sets each, byte compare of the TSVs.

usage: python tests/golden/check_export_vs_reference.py [n_indexes] [n_orfs] [first_seed]      exit 0 = all identical"""
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.environ.get("RIBOTRICER_REFERENCE", "/root/reference"))
os.environ["RIBOTRICER_AMD_BACKEND"] = "cpu"
os.environ["RIBOTRICER_AMD_INDEX_CACHE"] = "0"
for _m in ("pysam", "quicksect"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["quicksect"].Interval = object
sys.modules["quicksect"].IntervalTree = dict
from ribotricer.detect_orfs import export_orf_coverages as ref_export  # noqa: E402  (the reference)

from random_index import PARAM_SETS, random_index  # noqa: E402
from ribotricer_amd.detect_orfs import export_orf_coverages as amd_export  # noqa: E402

n_indexes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_orfs = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 7001
SHAPES = [dict(), dict(malformed=0.15), dict(dressed=0.2), dict(crlf=True), dict(final_newline=False, malformed=0.1),
          dict(malformed=0.3, dressed=0.3)]
bad = total = 0
with tempfile.TemporaryDirectory() as tmp:
    for k in range(n_indexes):
        shape = SHAPES[k % len(SHAPES)]
        text, merged = random_index(n_orfs, seed0 + k, **shape)
        index_path = os.path.join(tmp, f"index{k}.tsv")
        with open(index_path, "w", newline="") as fh:
            fh.write(text)
        # the three fixed threshold sets, and two drawn per index -- values that sit ON what the data produce (a density of
        # exactly 1.0 or 2.5, a ratio of exactly 0.5, a cutoff equal to a printed score) included
        import numpy as np

        trng = np.random.default_rng(seed0 + k)
        drawn = {}
        for j in range(2):
            drawn[f"drawn{j}"] = dict(report_all=bool(trng.integers(0, 2)), phase_score_cutoff=float(trng.choice([0.0, 0.25, 0.428571428571, 0.5, 1.0, float(trng.random())])),
                                      min_valid_codons=int(trng.integers(0, 12)), min_reads_per_codon=float(trng.choice([0, 0, 1, 2, 0.5])),
                                      min_valid_codons_ratio=float(trng.choice([0, 0.5, 0.25, 1.0, float(trng.random())])),
                                      min_density_over_orf=float(trng.choice([0.0, 1.0, 2.5, 0.1, float(trng.random() * 3)])))
        for name, kw in {**PARAM_SETS, **drawn}.items():
            total += 1
            outs = []
            for tag, fn in (("ref", ref_export), ("amd", amd_export)):
                prefix = os.path.join(tmp, f"{tag}{k}_{name}")
                try:
                    fn(index_path, merged, prefix, **kw)
                    with open(prefix + "_translating_ORFs.tsv", "rb") as fh:
                        outs.append(fh.read())
                except BaseException as e:  # noqa: BLE001  (sys.exit included: the reference's way of refusing an index)
                    outs.append(("raised", type(e).__name__, str(e)[:200]))
            same = outs[0] == outs[1] if isinstance(outs[0], bytes) or isinstance(outs[1], bytes) else True  # both raise: fine
            if not same:
                bad += 1
                print(f"DIFF index {k} (seed {seed0 + k}, {shape}) / {name}:")
                if isinstance(outs[0], bytes) and isinstance(outs[1], bytes):
                    a, b = outs[0].split(b"\n"), outs[1].split(b"\n")
                    print(f"  {len(a)} vs {len(b)} lines")
                    for i, (x, y) in enumerate(zip(a, b)):
                        if x != y:
                            print(f"  line {i}\n   ref: {x[:300]!r}\n   amd: {y[:300]!r}")
                            break
                else:
                    print("  ref:", outs[0] if not isinstance(outs[0], bytes) else f"{len(outs[0])} bytes")
                    print("  amd:", outs[1] if not isinstance(outs[1], bytes) else f"{len(outs[1])} bytes")
            else:
                what = f"{outs[0].count(10) - 1} rows" if isinstance(outs[0], bytes) else f"both raise ({outs[0][1]} / {outs[1][1]})"
                print(f"ok   index {k} (seed {seed0 + k}, {shape}) / {name}: {what}")
print(f"{total - bad} of {total} files byte-identical")
sys.exit(1 if bad else 0)
