#!/usr/bin/env python3
"""Do the record writes cost the read stream less when their FOOTPRINT is small enough for the 256 MiB Infinity Cache?
csrc/stream_probe.hip k_stream_rw modes 18 / 19: every workgroup writes its 1 152-byte chunk per 32 KiB read as in the product, but the
chunks go round a ring of R bytes (R = everything: the product's layout).  If the die-level cache took the writes, a step cut into slices
whose records fit it -- score a slice, finish it, reuse the same record bytes for the next -- would shed the write penalty.
usage: probe_rw_ring.py [GiB]"""
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
    out = torch.empty(blocks * w + (1 << 20), dtype=torch.uint8, device="cuda:0")
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
    print(f"read {nbytes / 1e9:.2f} GB in {blocks} pieces of 32 KiB, {w} bytes written per piece ({blocks * w / 1e6:.0f} MB); no writes: {base:.3f} ms = {nbytes / base / 1e6:.0f} GB/s")
    rings = [w * 256, 1 << 20, 4 << 20, 16 << 20, 32 << 20, 64 << 20, 128 << 20, 192 << 20, 256 << 20, 384 << 20, blocks * w]
    for rnd in range(2):
        for mode, name in ((18, "nt"), (19, "ordinary")):
            for r in rings:
                ring = r // w * w
                ms = run(mode, ring)
                print(f"round {rnd}: {name:8s} stores, ring of {ring / 1048576:8.2f} MiB: {ms:.3f} ms (+{(ms / base - 1) * 100:4.1f} %)", flush=True)
        print(f"round {rnd}: no writes again: {run(0, 0):.3f} ms")


if __name__ == "__main__":
    main()
