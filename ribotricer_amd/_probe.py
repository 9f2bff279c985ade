"""Measuring stick for bench.py and scripts/: a plain streaming read of a device buffer
(csrc/stream_probe.hip -> libstreamprobe.so).  Not used by any product path."""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "csrc", "libstreamprobe.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} is missing: `make -C ribotricer_amd/csrc`")
        _lib = ctypes.CDLL(_PATH)
        for name in ("sp_stream_read", "sp_stream_read_lds"):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def stream_read_GBps(buf, launches: int = 20, warmup: int = 3, flavour: str = "registers"):
    """Median-free mean rate (GB/s) and ms per launch of reading `buf` (a device tensor) once per launch."""
    import torch

    lib = _load()
    fn = lib.sp_stream_read if flavour == "registers" else lib.sp_stream_read_lds
    nbytes = (buf.numel() * buf.element_size()) // 32768 * 32768
    assert nbytes > 0 and buf.data_ptr() % 16 == 0
    scratch = torch.zeros(1, dtype=torch.int64, device=buf.device)
    stream = torch.cuda.current_stream(buf.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def go(n):
        for _ in range(n):
            rc = fn(buf.data_ptr(), nbytes, scratch.data_ptr(), stream.cuda_stream)
            if rc:
                raise RuntimeError(f"stream probe failed ({rc})")

    go(warmup)
    e0.record(stream)
    go(launches)
    e1.record(stream)
    torch.cuda.synchronize(buf.device)
    ms = e0.elapsed_time(e1) / launches
    return nbytes / ms / 1e6, ms
