#!/usr/bin/env python3
"""Index parser with and without huge pages for its output arrays (RIBOPHASE_HUGEPAGES=0/1), each in its own process,
on an 11 M-line synthetic index.  usage: bench_index_hugepages.py [n_lines]"""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import gc, os, sys, time
sys.path.insert(0, %r)
from ribotricer_amd.index import NativeIndex
path = sys.argv[1]
best = []
for rep in range(4):
    t = time.perf_counter(); ix = NativeIndex.from_file(path); dt = time.perf_counter() - t
    n = ix.n_orfs
    del ix; gc.collect(); best.append(round(dt, 3))
print(os.environ.get("RIBOPHASE_HUGEPAGES", "1"), n, best, flush=True)
""" % REPO


def main():
    n = sys.argv[1] if len(sys.argv) > 1 else "11000000"
    tmp = tempfile.mkdtemp(prefix="rphp_")
    exe = os.path.join(tmp, "gen")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(REPO, "scripts", "gen_big_index.cpp")])
    subprocess.check_call([exe, os.path.join(tmp, "big"), n], stderr=subprocess.DEVNULL)
    path = os.path.join(tmp, "big_candidate_orfs.tsv")
    open(path, "rb").read()  # page cache
    print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "usable cores", len(os.sched_getaffinity(0)))
    for hp in ("0", "1", "0", "1"):
        subprocess.check_call([sys.executable, "-c", CHILD, path], env=dict(os.environ, RIBOPHASE_HUGEPAGES=hp))


if __name__ == "__main__":
    main()
