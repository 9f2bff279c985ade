#!/usr/bin/env python3
"""What does FETCH_SIZE count for the fused kernel's access pattern?  (round-5 verdict, weak #6)

Reads a 4 GiB buffer once per launch with five kernels of csrc/stream_probe.hip -- the two 16-byte-per-lane controls the
microarch guide calibrates (x2) and the three dword LDS-DMA patterns of k_tile_score<true> -- a KNOWN byte count each.

    python scripts/fetch_calibration.py                       timings only (GB/s per pattern)
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o fc -- python3 scripts/fetch_calibration.py
    python scripts/fetch_calibration.py --parse DIR           bytes / FETCH_SIZE per kernel -> the factor

(FETCH_SIZE is reported in KiB on this rocprofv3; the factor printed is known bytes / (FETCH_SIZE x 1024).)"""
import csv
import glob
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
BYTES = 4 << 30
LAUNCHES = 6


def run():
    import torch

    from ribotricer_amd._probe import stream_read_GBps

    buf = torch.randint(0, 1 << 20, (BYTES // 4,), dtype=torch.int32, device="cuda")
    assert buf.data_ptr() % 256 == 0
    for flavour, what in (("registers", "16 B/lane global_load_dwordx4 (guide: x2)"), ("lds_dma", "16 B/lane global_load_lds_dwordx4 (guide: x2)"),
                          ("lds_dword0", "dword LDS-DMA, whole 256-byte lines"), ("lds_dword1", "dword LDS-DMA, a line in two instructions (24 + 40 lanes)"),
                          ("lds_dword2", "dword LDS-DMA, whole lines, descending lanes ('-' strand)")):
        gbps, ms = stream_read_GBps(buf, launches=LAUNCHES, warmup=2, flavour=flavour)
        print(f"{flavour:11s} {what:62s} {BYTES} bytes per launch, {ms:.3f} ms, {gbps:.0f} GB/s")
    # WRITE_SIZE: the record stream of k_tile_score (1 152 bytes per 32 KiB tile, three planes, nt stores) beside a read of
    # 2 GiB (k_stream_rw mode 15); mode 0 of the same kernel writes nothing -- the pass sees both under one kernel name,
    # half of the launches each: (mean WRITE_SIZE over all launches) x 2 = the writing launches'
    from ribotricer_amd._probe import write_penalty

    out = torch.zeros(3 * 384 * (2 << 30) // 32768 + 4096, dtype=torch.uint8, device="cuda")
    pen = write_penalty(buf, out, read_bytes=2 << 30, launches=4)
    print(f"k_stream_rw: {(2 << 30) // 32768 * 1152} bytes written per writing launch (1 152 per 32 KiB of a 2 GiB read); "
          f"read-only {pen[1]:.3f} ms, with writes {pen[2]:.3f} ms")


def parse(directory):
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            rows += [r for r in csv.DictReader(fh) if r.get("Counter_Name") == "FETCH_SIZE"]
    per = {}
    for r in rows:
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    if not rows:  # (a WRITE_SIZE pass)
        for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as fh:
                rows += [r for r in csv.DictReader(fh) if r.get("Counter_Name") == "WRITE_SIZE" and "k_stream_rw" in r.get("Kernel_Name", "")]
        vals = sorted(float(r["Counter_Value"]) for r in rows)
        writers = [v for v in vals if v > 0.25 * vals[-1]]  # (the mode-15 launches; mode 0 writes nothing)
        known = (2 << 30) // 32768 * 1152
        mid = writers[len(writers) // 2]
        print(f"k_stream_rw: {len(vals)} launches, {len(writers)} of them writing; known bytes written {known}; WRITE_SIZE median of the writing "
              f"launches {mid:.1f} KiB = {mid * 1024:.4g} B -> factor {known / (mid * 1024):.4f}; the others: max {max([v for v in vals if v not in writers] or [0]):.1f} KiB")
        return
    print(f"known bytes per launch: {BYTES}")
    for name, vals in sorted(per.items()):
        if "k_stream_read" not in name:
            continue
        vals = sorted(vals)
        mid = vals[len(vals) // 2]
        print(f"{name:60s} launches {len(vals):2d}  FETCH_SIZE median {mid:14.1f} KiB = {mid * 1024:.4g} B  ->  factor {BYTES / (mid * 1024):.4f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
