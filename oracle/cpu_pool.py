"""All-cores CPU baseline of the reference-equivalent path -- TEST / BENCH INFRASTRUCTURE ONLY.

``bench.py``'s ``cpu_baseline`` leg times the literal restatement of the reference's
``phasescore`` (oracle/phasescore_literal.py: the pure-Python triplet loop of
ribotricer/statistics.py:67-91 plus the identical ``scipy.signal.coherence`` call of
statistics.py:101-107) on many host cores at once, the way a user would parallelise the
reference's single-threaded loop (detect_orfs.py:274-324) with ``multiprocessing``.

The workers are plain child processes (``python -m oracle.cpu_pool``) started BEFORE the
benchmark touches the GPU and driven over pipes: load a chunk, wait for "go", score it,
report (n, t_start, t_end).  A worker that fails simply exits -- nothing respawns.
"""

from __future__ import annotations

import io
import os
import struct
import subprocess
import sys
import time

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _send_array(pipe, a: np.ndarray) -> None:
    buf = io.BytesIO()
    np.save(buf, np.ascontiguousarray(a), allow_pickle=False)
    data = buf.getvalue()
    pipe.write(struct.pack("<q", len(data)))
    pipe.write(data)


def _recv_array(pipe) -> np.ndarray:
    (n,) = struct.unpack("<q", pipe.read(8))
    return np.load(io.BytesIO(pipe.read(n)), allow_pickle=False)


def worker_main() -> None:
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"  # the reference is single-threaded; the pool supplies the cores
    sys.path.insert(0, _REPO)
    from oracle.phasescore_literal import phasescore_literal

    phasescore_literal([1, 0, 0, 2, 0, 0])  # scipy imported and warm before "ready"
    inp, out = sys.stdin.buffer, sys.stdout
    out.write("ready\n")
    out.flush()
    counts = _recv_array(inp)
    offsets = _recv_array(inp)
    profiles = [counts[offsets[i] : offsets[i + 1]].tolist() for i in range(offsets.size - 1)]
    out.write("loaded\n")
    out.flush()
    if inp.read(1) != b"g":
        return
    t0 = time.time()
    for p in profiles:
        phasescore_literal(p)
    t1 = time.time()
    out.write(f"{len(profiles)} {t0!r} {t1!r}\n")
    out.flush()


class CpuPool:
    """``CpuPool(n)`` starts n idle workers; ``run(counts, offsets, per_worker)`` scores
    ``per_worker`` ORFs on each of them concurrently and returns (ORFs, wall seconds, workers)."""

    def __init__(self, n_workers: int):
        self.n = max(1, int(n_workers))
        env = dict(os.environ, PYTHONPATH=_REPO + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        self.procs = [
            subprocess.Popen([sys.executable, "-m", "oracle.cpu_pool"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                             cwd=_REPO, env=env, text=False)
            for _ in range(self.n)
        ]
        for p in self.procs:
            line = p.stdout.readline()
            if line.strip() != b"ready":
                self.close()
                raise RuntimeError("oracle.cpu_pool worker failed to start")

    def run(self, counts: np.ndarray, offsets: np.ndarray, per_worker: int):
        n_orfs = offsets.size - 1
        per_worker = max(1, min(per_worker, n_orfs // self.n if n_orfs >= self.n else 1))
        used = []
        for w, p in enumerate(self.procs):
            lo, hi = w * per_worker, min(n_orfs, (w + 1) * per_worker)
            if hi <= lo:
                break
            a, b = int(offsets[lo]), int(offsets[hi])
            _send_array(p.stdin, counts[a:b])
            _send_array(p.stdin, offsets[lo : hi + 1] - offsets[lo])
            p.stdin.flush()
            used.append(p)
        for p in used:
            if p.stdout.readline().strip() != b"loaded":
                raise RuntimeError("oracle.cpu_pool worker died while loading")
        for p in used:  # all chunks are in place: start together
            p.stdin.write(b"g")
            p.stdin.flush()
        res = []
        for p in used:
            n, t0, t1 = p.stdout.readline().split()
            res.append((int(n), float(t0), float(t1)))
        n_done = sum(r[0] for r in res)
        wall = max(r[2] for r in res) - min(r[1] for r in res)
        return n_done, wall, len(used)

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        self.procs = []


if __name__ == "__main__":
    worker_main()
