"""Rows/s of the native TSV renderer (rp_format_rows_host) next to the
'{}'.format loop it replaces, on a cfg2-shaped batch with report_all (every row written).
usage: python scripts/bench_format.py [n_orfs] [python_sample]"""
import os
import sys
import time
from collections import namedtuple

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ribotricer_amd import detect_orfs as d  # noqa: E402
from ribotricer_amd import synth, tsv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
n_py = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
counts, offsets = synth.synth_csr_host(n, cfg="cfg2")
rng = np.random.default_rng(1)
res = dict(phase=rng.random(n), valid=rng.integers(0, 100, n).astype(np.int32),
           read_count=rng.integers(0, 5000, n).astype(np.int64), status=(rng.random(n) < 0.36).astype(np.uint8))
Rec = namedtuple("Rec", "oid category tid ttype gid gname gtype chrom strand start_codon")
records = [Rec(f"ENST{i:011d}_{i*7}_{i*7+300}_300", "annotated", f"ENST{i:011d}", "protein_coding", f"ENSG{i:011d}",
               f"GENE{i}", "protein_coding", "chr1", "+", "ATG") for i in range(n)]

tsv.double_repr(0.5)  # loads the library
t = time.perf_counter()
tables = tsv.record_tables(records)
t_tab = time.perf_counter() - t
t = time.perf_counter()
nbytes = sum(len(c) for c in d.format_rows_native(records, counts, offsets, res, True))
t_nat = time.perf_counter() - t  # includes building the string tables again
t = time.perf_counter()
py = sum(len(r) for r in d.format_rows(records[:n_py], counts, offsets[: n_py + 1], {k: v[:n_py] for k, v in res.items()}, True))
t_py = time.perf_counter() - t
print(f"native: {n} rows, {nbytes/1e6:.1f} MB in {t_nat:.2f} s (string tables {t_tab:.2f} s) -> "
      f"{n/t_nat:,.0f} rows/s, {nbytes/1e6/t_nat:.0f} MB/s")
print(f"python: {n_py} rows, {py/1e6:.1f} MB in {t_py:.2f} s -> {n_py/t_py:,.0f} rows/s, {py/1e6/t_py:.0f} MB/s")
print(f"speed-up {n/t_nat/(n_py/t_py):.1f}x (1 thread each)")
