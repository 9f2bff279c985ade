"""Lines/s of the native index parser (rp_index_parse_host) next to the
per-line Python parser it replaces.  usage: python scripts/bench_index.py [n_lines] [python_sample]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ribotricer_amd import detect_orfs as d  # noqa: E402
from ribotricer_amd.index import NativeIndex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_py = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
rng = np.random.default_rng(3)
lines = ["ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n"]
pos = 1000
for i in range(n):
    k = int(rng.integers(1, 6))
    ivs = []
    for _ in range(k):
        ln = int(rng.integers(30, 200))
        ivs.append(f"{pos}-{pos + ln - 1}")
        pos += ln + int(rng.integers(50, 500))
    lines.append(f"ENST{i:011d}_{i}_x\tannotated\tENST{i:011d}\tprotein_coding\tENSG{i//3:011d}\tGENE{i//3}\tprotein_coding\tchr{1 + i % 22}\t{'+-'[i % 2]}\tATG\t{','.join(ivs)}\n")
text = "".join(lines).encode()
path = "/tmp/rp_bench_index.tsv"
open(path, "wb").write(text)
NativeIndex(b"")  # loads the library (HIP runtime start-up is not parser time)
times = []
for _ in range(3):
    t = time.perf_counter()
    ni = NativeIndex.from_file(path)
    times.append(time.perf_counter() - t)
print('native runs (s):', [round(x, 3) for x in times])
t_nat = min(times)
open(path, "wb").write("".join(lines[: n_py + 1]).encode())
t = time.perf_counter()
recs = d.read_index(path)
t_py = time.perf_counter() - t
os.remove(path)
assert ni.n_orfs == n and len(recs) == n_py
print(f"native: {n} lines, {len(text)/1e6:.1f} MB in {t_nat:.2f} s -> {n/t_nat:,.0f} lines/s, {len(text)/1e6/t_nat:.0f} MB/s (incl. numpy copies)")
print(f"python: {n_py} lines in {t_py:.2f} s -> {n_py/t_py:,.0f} lines/s")
print(f"speed-up {n/t_nat/(n_py/t_py):.1f}x (1 thread each)")
