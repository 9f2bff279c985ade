#!/usr/bin/env python3
"""A read stream with a sprinkle of writes (csrc/stream_probe.hip, k_stream_rw): 32 KiB pieces DMA'd into LDS by
one loader wave, four workgroups per CU -- the scoring kernel's skeleton without its arithmetic -- and W bytes
written per workgroup in several forms.  What do k_tile_score's segment records (~1.2 KB per 31 KiB tile) cost the
read stream, and is there a cheap way to write them?   usage: probe_rw.py [GiB to read]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
MODES = {0: "no writes", 1: "one wave, dwordx4, contiguous", 2: "three waves, three planes (product layout)", 3: "as 1, nt",
         4: "as 1, sc0 sc1", 5: "as 1, dword per lane", 6: "every 8th workgroup writes 8x", 7: "scalar stores", 8: "as 1, at workgroup start",
         9: "return-less 64-bit atomic swaps", 10: "as 1, dword nt", 11: "as 1, sc1 nt", 12: "as 1, sc0 sc1 nt", 13: "as 1, sc0", 14: "as 1, sc1",
         15: "three planes, nt", 0x101: "WRITES ALONE, dwordx4", 0x103: "WRITES ALONE, nt"}


def main():
    import torch

    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 14.8
    lib = ctypes.CDLL(os.path.join(REPO, "ribotricer_amd", "csrc", "libstreamprobe.so"))
    lib.sp_stream_rw.restype = ctypes.c_int
    lib.sp_stream_rw.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    nbytes = int(gib * (1 << 30)) // 32768 * 32768
    blocks = nbytes // 32768
    src = torch.randint(0, 9, (nbytes // 4,), dtype=torch.int32, device="cuda:0")
    plane = blocks * 3072
    outs = [torch.empty(3 * plane + (64 << 20) * k, dtype=torch.uint8, device="cuda:0") for k in range(2)]
    scratch = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    stream = torch.cuda.current_stream()

    def run(mode, w, out, launches=12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(launches + 2):
            if i == 2:
                e0.record()
            rc = lib.sp_stream_rw(src.data_ptr(), nbytes, out.data_ptr(), plane, w, mode, scratch.data_ptr(), stream.cuda_stream)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / launches

    base = [run(0, 0, outs[0]) for _ in range(3)]
    print(f"read {nbytes / 1e9:.2f} GB in {blocks} pieces; no writes: {min(base):.3f} ms = {nbytes / min(base) / 1e6:.0f} GB/s")
    print(f"{'mode':<46} {'W':>5} " + " ".join(f"{'out buffer ' + str(k):>22}" for k in range(len(outs))))
    table = [(1, 48), (1, 384), (1, 1152), (1, 3072), (2, 1152), (3, 1152), (4, 1152), (5, 1152), (6, 1152), (7, 1152), (7, 48), (8, 1152), (9, 1152), (0, 0)]
    if len(sys.argv) > 2 and sys.argv[2] == "nt":
        table = [(1, 1152), (3, 48), (3, 384), (3, 768), (3, 1152), (3, 3072), (15, 1152), (10, 1152), (11, 1152), (12, 1152), (13, 1152), (14, 1152), (0x101, 3072), (0x103, 3072), (0x101, 1152), (0x103, 1152), (0, 0)]
    for mode, w in table:
        cells = []
        for out in outs:
            ms = run(mode, w, out)
            cells.append(f"{ms:.3f} ms (+{(ms / min(base) - 1) * 100:4.1f} %)")
        print(f"{MODES[mode]:<46} {w:>5} " + " ".join(f"{c:>22}" for c in cells), flush=True)


if __name__ == "__main__":
    main()
