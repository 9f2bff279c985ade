#!/usr/bin/env python3
"""Does the KIND of memory behind the record workspace change what the record stores cost the scoring kernel?
hipExtMallocWithFlags: default / fine-grained / uncached / contiguous, next to torch's allocation; the cfg3 step timed on each
(HIP events inside the library, median of 9).  usage: python scripts/ws_memtype_probe.py [n_orfs]"""
import ctypes
import json
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ribotricer_amd import _lib  # noqa: E402
from ribotricer_amd.engine import PhaseScoreEngine, _ptr, make_filter  # noqa: E402
from ribotricer_amd.synth import synth_csr_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
dev = torch.device("cuda", 0)
counts, offsets = synth_csr_device(n, cfg="cfg3", device=dev)
eng = PhaseScoreEngine(dev)
th = make_filter()
plan = eng.plan_for(offsets, counts.numel(), 0)
out = eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True, plan=plan)
torch.cuda.synchronize()
need = _lib.workspace_bytes(n, counts.numel(), _lib.RP_ALGO_TILE)
hip = ctypes.CDLL("libamdhip64.so")
lib = _lib.load()
outputs = [_ptr(out.phase), _ptr(out.valid), _ptr(out.read_count), _ptr(out.min_codon_cov), _ptr(out.flags), _ptr(out.status)]
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timed(ws_ptr):
    ms = (ctypes.c_float * 4)()
    t = []
    for k in range(12):
        _lib.check(lib.rp_phase_score_csr_dev_timed(0, _ptr(counts), _ptr(offsets), n, counts.numel(), *outputs, ctypes.byref(th),
                                                    ctypes.c_void_p(ws_ptr), need, _lib.RP_ALGO_TILE, plan.handle, stream, ctypes.byref(ms)))
        if k >= 3:
            t.append((ms[1], ms[2]))
    return round(statistics.median(x[0] for x in t), 4), round(statistics.median(x[1] for x in t), 4)


rep = {}
ws_t = torch.empty(need, dtype=torch.uint8, device=dev)
rep["torch.empty"] = timed(ws_t.data_ptr())
for name, flag in (("default", 0), ("finegrained", 1), ("uncached", 3), ("contiguous", 4)):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(need), ctypes.c_uint(flag))
    if rc != 0:
        rep[name] = f"hipExtMallocWithFlags rc={rc}"
        continue
    rep[name] = timed(p.value)
    ref = {k: getattr(out, k).clone() for k in ("phase", "valid", "read_count")}
    torch.cuda.synchronize()
    hip.hipFree(p)
rep["torch.empty again"] = timed(ws_t.data_ptr())
print(json.dumps({"kernel_ms, finish_ms by workspace memory": rep}))
