import os, subprocess, sys, time, gc, tempfile
sys.path.insert(0, os.getcwd())
from ribotricer_amd.index import NativeIndex
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "gen"); subprocess.check_call(["g++", "-O2", "-o", exe, "scripts/gen_big_index.cpp"])
subprocess.check_call([exe, os.path.join(tmp, "big"), "11000000"], stderr=subprocess.DEVNULL)
path = os.path.join(tmp, "big_candidate_orfs.tsv")
open(path, "rb").read()
for th in (1, 4, 8, 16, 24, 32, 64, 128):
    os.environ["RIBOPHASE_INDEX_THREADS"] = str(th)
    best = 99
    for rep in range(2):
        t = time.perf_counter(); ix = NativeIndex.from_file(path); dt = time.perf_counter() - t
        del ix; gc.collect(); best = min(best, dt)
    print(th, f"{best:.3f} s", flush=True)
