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
        _lib.sp_stream_read_lds_dword.restype = ctypes.c_int
        _lib.sp_stream_read_lds_dword.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.sp_stream_rw.restype = ctypes.c_int
        _lib.sp_stream_rw.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def stream_read_GBps(buf, launches: int = 20, warmup: int = 3, flavour: str = "registers"):
    """Median-free mean rate (GB/s) and ms per launch of reading `buf` (a device tensor) once per launch."""
    import torch

    lib = _load()
    if flavour.startswith("lds_dword"):  # "lds_dword0" / "lds_dword1" / "lds_dword2": the fused kernel's pattern (stream_probe.hip)
        mode = int(flavour[-1])

        def fn(ptr, n, scratch_ptr, stream_ptr):
            return lib.sp_stream_read_lds_dword(ptr, n, scratch_ptr, stream_ptr, mode)
    else:
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



def write_penalty(read_buf, write_buf, read_bytes: int = 2 << 30, launches: int = 8, where: float = 0.5):
    """How much does a sprinkle of writes into `write_buf` (1 152 bytes per 32 KiB read, streaming stores, three
    planes: the scoring kernel's record stream) cost a read stream over a 2 GiB slice of `read_buf` (`where`)?  Returns
    (penalty, ms without writes, ms with writes); penalty = with / without - 1.  On MI355X the penalty is ~0.10 when
    the two buffers lie in different classes of the physical address space and ~0.23 when they share one
    (profiles/archive/r03_probe_rw_regions.txt) -- the spread of the scoring kernel over "placements".  Overwrites the
    head of `write_buf` (a uint8 tensor of >= 3 * 384 * read_bytes / 32768 bytes)."""
    import torch

    lib = _load()
    total = read_buf.numel() * read_buf.element_size()
    nbytes = min(read_bytes, total) // 32768 * 32768
    blocks = nbytes // 32768
    plane = blocks * 384
    need = 3 * plane
    wtotal = write_buf.numel() * write_buf.element_size()
    if nbytes == 0 or wtotal < need:
        return None
    rptr = read_buf.data_ptr() + int((total - nbytes) * min(max(where, 0.0), 1.0)) // 32768 * 32768  # `where`: 0 = head, 1 = tail of read_buf
    wptr = write_buf.data_ptr()  # the head of the buffer: where the kernels put their records
    scratch = torch.zeros(1, dtype=torch.int64, device=read_buf.device)
    stream = torch.cuda.current_stream(read_buf.device)

    def run(mode):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(launches + 2):
            if i == 2:
                e0.record(stream)
            rc = lib.sp_stream_rw(rptr, nbytes, wptr, plane, 1152, mode, scratch.data_ptr(), stream.cuda_stream)
            if rc:
                raise RuntimeError(f"stream probe failed ({rc})")
        e1.record(stream)
        torch.cuda.synchronize(read_buf.device)
        return e0.elapsed_time(e1) / launches

    plain = min(run(0), run(0))
    mixed = min(run(15), run(15))
    return mixed / plain - 1.0, plain, mixed

