#!/usr/bin/env python3
"""Round 5: does the KIND of memory behind the record workspace change what the record stores cost?  The 11 M-ORF step with the
workspace in ordinary device memory (torch), and in memory from hipExtMallocWithFlags: fine-grained (0x1), uncached (0x3),
physically contiguous (0x4).  Kernel / step from the library's events, results compared with the ordinary run.
usage: probe_workspace_memtype.py"""
import ctypes
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class Raw:
    """A stand-in for a uint8 tensor over memory this script allocated itself."""

    def __init__(self, ptr, nbytes):
        self._ptr, self._n = ptr, nbytes

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return self._n


def main():
    import numpy as np
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    eng = PhaseScoreEngine("cuda:0")
    th = make_filter()
    counts, offsets = synth_csr_device(11_000_000, cfg="cfg3", device="cuda:0")
    plan = eng.plan_for(offsets, counts.numel(), (counts.data_ptr() // 4) % 4)
    base = eng.score(counts, offsets, thresholds=th, reuse_outputs=True, plan=plan).cpu_numpy()
    base = {k: v.copy() for k, v in base.items()}
    key = int(torch.cuda.current_stream().cuda_stream)
    ordinary = eng._workspace[key]
    need = ordinary.numel()
    hip = None
    for name in os.listdir(os.path.join(os.path.dirname(torch.__file__), "lib")):
        if name.startswith("libamdhip64"):
            hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", name))
            break
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipExtMallocWithFlags.restype = ctypes.c_int
    kinds = [("ordinary (torch)", None), ("a second ordinary one (torch)", "torch"), ("fine-grained 0x1", 1), ("uncached 0x3", 3), ("contiguous 0x4", 4)]
    spaces = {}
    for name, flag in kinds:
        if flag is None:
            spaces[name] = ordinary
        elif flag == "torch":
            spaces[name] = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        else:
            p = ctypes.c_void_p(0)
            rc = hip.hipExtMallocWithFlags(ctypes.byref(p), need, flag)
            if rc != 0 or not p.value:
                print(f"{name}: hipExtMallocWithFlags -> {rc}")
                continue
            spaces[name] = Raw(p.value, need)
    for rnd in range(2):
        for name, ws in spaces.items():
            eng._workspace[key] = ws
            for _ in range(3):
                eng.score(counts, offsets, thresholds=th, reuse_outputs=True, plan=plan)
            t = []
            for _ in range(10):
                res = eng.score(counts, offsets, thresholds=th, reuse_outputs=True, plan=plan, timings=t)
            got = res.cpu_numpy()
            same = all(np.array_equal(got[k], base[k], equal_nan=True) for k in base)
            print(f"round {rnd} {name:32s} kernel {statistics.median(x[1] for x in t):.3f} finish {statistics.median(x[2] for x in t):.3f} ms  results equal: {same}", flush=True)
    eng._workspace[key] = ordinary


if __name__ == "__main__":
    main()
