#!/usr/bin/env python3
"""Where the export's "interval table + gather plan" stage goes on the 11 M-line synthetic index (scripts/gen_big_index.cpp):
the native table pass, the uploads, the coverage map, the gather plan, the host copy of the compact table -- wall ms each.
usage: python scripts/bench_table_plan.py [n_orfs]"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ribotricer_amd.gather import CoverageMap, GatherPlan, coverage_layout, interval_table_from_index  # noqa: E402
from ribotricer_amd.index import NativeIndex  # noqa: E402

n_orfs = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
tmp = tempfile.mkdtemp(prefix="rptp_")
exe = os.path.join(tmp, "gen_big_index")
subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(REPO, "scripts", "gen_big_index.cpp")])
prefix = os.path.join(tmp, "big")
subprocess.check_call([exe, prefix, str(n_orfs)])
torch.zeros(1, device="cuda")
out = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    out[name] = round(1e3 * (time.perf_counter() - t0), 2)
    return r


index = timed("index_parse", lambda: NativeIndex.from_file(prefix + "_candidate_orfs.tsv"))
base, total = coverage_layout(index.extents() if callable(index.extents) else index.extents)
for rep in ("first", "second"):
    table = timed(f"{rep}.interval_table_host", lambda: interval_table_from_index(index, base))
    d = timed(f"{rep}.upload_iv_start_iv_len", lambda: (torch.from_numpy(table.iv_start).cuda(), torch.from_numpy(table.iv_len).cuda()))
    del d
    cmap = timed(f"{rep}.coverage_map", lambda: CoverageMap(table, total, "cuda:0"))
    plan = timed(f"{rep}.gather_plan", lambda: GatherPlan(cmap.table, cmap.compact_len, "cuda:0", device_intervals=cmap.device_intervals))
    timed(f"{rep}.compact_iv_start_to_host", lambda: cmap.table.iv_start)
    del cmap, plan, table
    torch.cuda.empty_cache()
print(json.dumps(out))
