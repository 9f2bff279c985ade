#!/usr/bin/env python3
"""Why does the same kernel on the same bytes take 2.70 ms in one process and 3.08 ms in the next?

Times `k_tile_score` in blocks of steps over several seconds of sustained launches, idles, and
times again, while a sampler thread reads the GPU's clocks / power from sysfs (or rocm-smi when
sysfs is not readable).  One JSON document on stdout.

    python scripts/clock_trace.py --orfs 11000000 --sustain 3000 --idle 5,20
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _current(path):
    """pp_dpm_* lists levels; the active one carries a '*'."""
    try:
        for ln in open(path):
            if "*" in ln:
                return ln.split(":", 1)[1].replace("*", "").strip()
    except OSError:
        return None
    return None


def _num(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None


class Sampler(threading.Thread):
    def __init__(self, period, pci=None):
        super().__init__(daemon=True)
        self.period = period
        self.rows = []
        self.stop = False
        devs = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        # the card this process computes on: the one whose PCI address HIP reports
        mine = [d for d in devs if pci and pci.lower() in os.path.realpath(os.path.dirname(d)).lower()]
        self.dev = os.path.dirname((mine or devs)[0]) if devs else None
        self.matched = bool(mine)
        self.others = [os.path.dirname(d) for d in devs if os.path.dirname(d) != self.dev]
        self.hwmon = (sorted(glob.glob(self.dev + "/hwmon/hwmon*")) or [None])[0] if self.dev else None
        self.t0 = time.perf_counter()

    def sample(self):
        row = {"t": round(time.perf_counter() - self.t0, 3)}
        if self.dev:
            row["sclk"] = _current(self.dev + "/pp_dpm_sclk")
            row["mclk"] = _current(self.dev + "/pp_dpm_mclk")
            row["fclk"] = _current(self.dev + "/pp_dpm_fclk")
            row["busy"] = _num(self.dev + "/gpu_busy_percent")
            if len(self.rows) % 10 == 0:  # (64 sysfs reads: not with every sample)
                row["neighbours_busy"] = sum(1 for d in self.others if (_num(d + "/gpu_busy_percent") or 0) > 50)
        if self.hwmon:
            for key in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"):
                v = _num(self.hwmon + "/" + key)
                if v is not None:
                    row[key] = v
        if len(row) <= 2:  # nothing readable: ask rocm-smi (slow: one subprocess per sample)
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
                row["smi"] = json.loads(out.stdout) if out.stdout.strip().startswith("{") else out.stdout[-300:]
            except Exception as e:  # noqa: BLE001
                row["smi_error"] = str(e)
        return row

    def run(self):
        while not self.stop:
            self.rows.append(self.sample())
            time.sleep(self.period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orfs", type=int, default=11_000_000)
    ap.add_argument("--cfg", default="cfg3")
    ap.add_argument("--block", type=int, default=25)
    ap.add_argument("--sustain", type=int, default=3000)
    ap.add_argument("--idle", default="5,20")
    ap.add_argument("--period", type=float, default=0.1)
    ap.add_argument("--placement", action="store_true",
                    help="after the trace: the same bytes in other allocations (second copy of the counts, second engine = new plan + workspace)")
    a = ap.parse_args()

    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    props = torch.cuda.get_device_properties(0)  # (does not initialise the context)
    pci = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
    smp = Sampler(a.period, pci)
    doc = {"first_sample": smp.sample(), "pci": pci, "dev": smp.dev, "dev_matched": smp.matched, "hwmon": smp.hwmon, "cards": len(smp.others) + 1}
    smp.start()
    eng = PhaseScoreEngine("cuda:0")
    t_gen = time.perf_counter()
    counts, offsets = synth_csr_device(a.orfs, cfg=a.cfg, device="cuda:0")
    torch.cuda.synchronize()
    doc["generate_s"] = round(time.perf_counter() - t_gen, 3)
    th = make_filter()

    def blocks(n_steps, label):
        """n_steps scored back to back; the kernel time of every step by HIP events, averaged per block."""
        out = []
        done = 0
        while done < n_steps:
            t = []
            w0 = time.perf_counter() - smp.t0
            for _ in range(a.block):
                eng.score(counts, offsets, thresholds=th, algo="tile", reuse_outputs=True, timings=t)
            torch.cuda.synchronize()
            out.append({"t": round(w0, 3), "main_ms": round(sum(x[1] for x in t) / len(t), 4), "finish_ms": round(sum(x[2] for x in t) / len(t), 4)})
            done += a.block
        doc.setdefault("phases", []).append({"label": label, "blocks": out})
        ms = [b["main_ms"] for b in out]
        print(f"# {label}: first {ms[0]:.3f} min {min(ms):.3f} max {max(ms):.3f} last {ms[-1]:.3f} ms", file=sys.stderr, flush=True)

    def plain_read(label):
        """A plain streaming read of the same counts buffer (csrc/stream_probe.hip): does a kernel that
        does nothing but read see the mode too?"""
        from ribotricer_amd._probe import stream_read_GBps

        row = {"label": label}
        for flavour in ("registers", "lds_dma"):
            gbps, ms = stream_read_GBps(counts, flavour=flavour)
            row[flavour] = {"ms": round(ms, 4), "GBps": round(gbps, 1)}
        doc.setdefault("plain_read", []).append(row)
        print(f"# plain read of the counts ({label}): registers {row['registers']['GBps']:.0f} GB/s, LDS-DMA {row['lds_dma']['GBps']:.0f} GB/s", file=sys.stderr, flush=True)

    plain_read("before")
    blocks(100, "right after generating the set (bench.py's situation: 10 warmup + 100 steps)")
    blocks(a.sustain, "sustained")
    for s in [float(x) for x in a.idle.split(",") if x]:
        time.sleep(s)
        blocks(200, f"after {s:g} s idle")
    plain_read("after")
    if a.placement:
        # Is the mode a property of WHERE the buffers lie?  Same process, same bytes, other allocations.
        def timed(eng_, counts_, offsets_, label):
            t = []
            for _ in range(5):
                eng_.score(counts_, offsets_, thresholds=th, algo="tile", reuse_outputs=True)
            for _ in range(40):
                eng_.score(counts_, offsets_, thresholds=th, algo="tile", reuse_outputs=True, timings=t)
            ms = sorted(x[1] for x in t)[len(t) // 2]
            doc.setdefault("placement", []).append({"label": label, "main_ms": round(ms, 4), "counts_ptr": hex(counts_.data_ptr())})
            print(f"# placement: {label}: {ms:.3f} ms (counts at {hex(counts_.data_ptr())})", file=sys.stderr, flush=True)

        timed(eng, counts, offsets, "first counts, first engine")
        pad = torch.empty((1 << 30) + 4096 * 37, dtype=torch.uint8, device="cuda:0")  # shifts what follows
        counts2 = counts.clone()
        timed(eng, counts2, offsets, "second copy of the counts, first engine (same plan + workspace)")
        eng2 = PhaseScoreEngine("cuda:0")
        offsets2 = offsets.clone()
        timed(eng2, counts, offsets2, "first counts, second engine (new plan, workspace, outputs)")
        timed(eng2, counts2, offsets2, "second counts, second engine")
        timed(eng, counts, offsets, "first counts, first engine again")
        del pad
    smp.stop = True
    smp.join()
    doc["samples"] = smp.rows
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
