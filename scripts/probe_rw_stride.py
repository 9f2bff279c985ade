#!/usr/bin/env python3
"""Does it matter WHERE the record bytes of one workgroup land relative to each other?  csrc/stream_probe.hip k_stream_rw mode 16: the
1 152 bytes a workgroup writes per 32 KiB read go out as 64-byte pieces S bytes apart (memory still filled densely: S / 64 neighbouring
workgroups interleave their pieces).  If a write costs the read stream one bus turn-around per channel visit, a stride that folds a
workgroup's pieces onto one channel should be cheaper than the contiguous chunk (which spreads them over ~5).  usage: probe_rw_stride.py [GiB]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 14.8
    lib = ctypes.CDLL(os.path.join(REPO, "ribotricer_amd", "csrc", "libstreamprobe.so"))
    lib.sp_stream_rw.restype = ctypes.c_int
    lib.sp_stream_rw.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    nbytes = int(gib * (1 << 30)) // 32768 * 32768
    blocks = nbytes // 32768
    src = torch.randint(0, 9, (nbytes // 4,), dtype=torch.int32, device="cuda:0")
    w = 1152
    out = torch.empty(4 << 30, dtype=torch.uint8, device="cuda:0")  # (room for every stride below: ceil(blocks / (S / 64)) regions of 18 x S bytes)
    scratch = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    stream = torch.cuda.current_stream()

    def run(mode, plane, launches=12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(launches + 2):
            if i == 2:
                e0.record()
            rc = lib.sp_stream_rw(src.data_ptr(), nbytes, out.data_ptr(), plane, w if mode else 0, mode, scratch.data_ptr(), stream.cuda_stream)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / launches

    base = min(run(0, 0) for _ in range(3))
    print(f"read {nbytes / 1e9:.2f} GB in {blocks} pieces; no writes: {base:.3f} ms = {nbytes / base / 1e6:.0f} GB/s")
    for rnd in range(2):
        ms = run(3, blocks * 3072)
        print(f"round {rnd}: contiguous 1 152-byte chunk per workgroup, nt: {ms:.3f} ms (+{(ms / base - 1) * 100:4.1f} %)")
        for shift in range(6, 27, 2):
            S = 1 << shift
            assert -(-blocks // (S // 64)) * 18 * S <= out.numel(), S
            ms = run(16, S)
            print(f"round {rnd}: 64-byte pieces {S:>9} B apart, nt:            {ms:.3f} ms (+{(ms / base - 1) * 100:4.1f} %)", flush=True)
        print(f"round {rnd}: no writes again: {run(0, 0):.3f} ms")


if __name__ == "__main__":
    main()
