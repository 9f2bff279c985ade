#!/usr/bin/env python3
"""Does the cost of the sprinkled writes depend on WHERE the written buffer lies relative to the read buffer?  One big
allocation; the read stream (15.9 GB, csrc/stream_probe.hip k_stream_rw) starts at offset A, the 1 152-byte-per-piece
write stream at offset B; a coarse (A, B) grid.  usage: probe_rw_regions.py [total GiB] [step GiB] [read GiB]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    total_gib = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    step_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    lib = ctypes.CDLL(os.path.join(REPO, "ribotricer_amd", "csrc", "libstreamprobe.so"))
    lib.sp_stream_rw.restype = ctypes.c_int
    lib.sp_stream_rw.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    read_gib = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    read_bytes = (read_gib << 30) // 32768 * 32768
    blocks = read_bytes // 32768
    plane = blocks * 384
    write_span = 3 * plane
    big = torch.zeros(total_gib << 30, dtype=torch.uint8, device="cuda:0")
    scratch = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    stream = torch.cuda.current_stream()
    base = big.data_ptr()
    print(f"allocation at {hex(base)}, {total_gib} GiB; read {read_bytes / 2**30:.1f} GiB at A, write {write_span / 2**30:.2f} GiB (three planes, nt) at B", flush=True)

    def run(a_gib, b_gib, mode, launches=8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(launches + 2):
            if i == 2:
                e0.record()
            rc = lib.sp_stream_rw(base + (a_gib << 30), read_bytes, base + (b_gib << 30), plane, 1152, mode, scratch.data_ptr(), stream.cuda_stream)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / launches

    a_list = [a for a in range(0, total_gib - read_gib + 1, step_gib)]
    b_list = [b for b in range(0, total_gib - 1, step_gib // 2 or 1)]
    print("rows: read offset A (GiB); columns: write offset B (GiB); ms per launch, '-' where the two overlap; mode 15 (three planes, nt stores)")
    print("A\\B  " + " ".join(f"{b:5d}" for b in b_list) + "   no-writes")
    for a in a_list:
        cells = []
        for b in b_list:
            if b + 1 > a and b < a + read_gib:
                cells.append("    -")
            else:
                cells.append(f"{run(a, b, 15):5.2f}")
        print(f"{a:4d} " + " ".join(cells) + f"   {run(a, 0 if a >= 1 else total_gib - 1, 0):5.2f}", flush=True)


if __name__ == "__main__":
    main()
