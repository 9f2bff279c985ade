"""export_wig on synthetic merged columns: wall seconds.  usage: python scripts/bench_wig.py [rows]"""
import time, sys, os, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ribotricer_amd.alignments import MergedColumns
from ribotricer_amd import detect_orfs as d
n=int(sys.argv[1]) if len(sys.argv)>1 else 20_000_000
rng=np.random.default_rng(1)
cols=MergedColumns(rng.integers(0,2,n).astype(np.uint8), rng.integers(0,24,n).astype(np.int32), rng.integers(0,150_000_000,n).astype(np.int64), rng.integers(1,50,n).astype(np.int64), [f"chr{k}" for k in range(1,25)])
tmp=tempfile.mkdtemp()
t0=time.perf_counter(); d.export_wig(cols, os.path.join(tmp,"x")); print("export_wig", n, "entries:", round(time.perf_counter()-t0,2), "s", sum(os.path.getsize(os.path.join(tmp,f)) for f in os.listdir(tmp))>>20, "MiB")
