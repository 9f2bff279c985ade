#!/usr/bin/env python3
"""A longer run of tests/test_gpu_export.py::test_corner_indexes_hip_equals_cpu_backend: seeds lo..hi of tests/golden/random_index.py
(corner-case indexes: overlapping / nested / duplicated / 1-nt exons, shuffled lists, a '.' strand, blocks with end < start, dressed
numbers, CRLF, no final newline) through the HIP export -- one GPU and three slices, default mode and report_all -- against the cpu
backend (whose bytes equal the reference's on this generator's indexes: tests/golden/check_export_vs_reference.py).
usage: soak_export_fuzz.py [lo] [hi] [n_orfs]"""
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def same_rows(a_text, b_text, tol=1e-6):
    a_rows, b_rows = a_text.splitlines(), b_text.splitlines()
    assert len(a_rows) == len(b_rows) and a_rows[0] == b_rows[0], (len(a_rows), len(b_rows))
    for a, b in zip(a_rows[1:], b_rows[1:]):
        a, b = a.split("\t"), b.split("\t")
        assert a[:3] == b[:3] and a[4:] == b[4:] and abs(float(a[3]) - float(b[3])) <= tol, (a[:9], b[:9])


def main():
    from random_index import random_index

    from ribotricer_amd import detect_orfs as d

    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    n_orfs = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    shapes = [dict(), dict(malformed=0.15), dict(dressed=0.2), dict(crlf=True), dict(final_newline=False, malformed=0.1), dict(malformed=0.3, dressed=0.3)]
    t0 = time.time()
    rows = 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(lo, hi):
            text, merged = random_index(n_orfs, seed, **shapes[seed % 6])
            index = os.path.join(tmp, f"i{seed}_candidate_orfs.tsv")
            with open(index, "w", newline="") as fh:
                fh.write(text)
            for report_all in (False, True):
                outs = {}
                for tag, backend, devices in (("cpu", "cpu", None), ("hip", "hip", None), ("three", "hip", [0, 0, 0])):
                    os.environ["RIBOTRICER_AMD_BACKEND"] = backend
                    d.forget_indexes()
                    d.export_orf_coverages(index, merged, os.path.join(tmp, tag), report_all=report_all, devices=devices)
                    outs[tag] = open(os.path.join(tmp, tag) + "_translating_ORFs.tsv", newline="").read()
                try:
                    same_rows(outs["cpu"], outs["hip"])
                    same_rows(outs["cpu"], outs["three"])
                except AssertionError as e:
                    print(f"seed {seed} report_all={report_all}: DIFFERENT {e}")
                    sys.exit(1)
                rows += outs["cpu"].count("\n") - 1
            os.remove(index)
    d.forget_indexes()
    print(f"seeds {lo}..{hi - 1} x {n_orfs} ORFs: hip == hip over three slices == cpu backend on {rows} printed rows (phase within 1e-6, every other column the same text); {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
