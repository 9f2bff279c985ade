"""Batch phase scoring on one MI355X through the C ABI (include/ribophase.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every number
is produced by the hand-written gfx950 kernels in ``csrc/``.  Nothing here ever
falls back to the CPU -- without a GPU (or without the built library) the calls raise
(the GPU-less backend of the drop-in functions is a separate, explicitly selected path: backend.py).

The batch call replaces the per-ORF body of the reference's hot loop
(ribotricer/detect_orfs.py:274-299): ``sum(cov)``, ``phasescore(cov)``
(statistics.py:48-115), ``collapse_coverage_to_codon`` (common.py:164-180) and the
status predicate, for all ORFs of a CSR-packed batch at once.
"""

from __future__ import annotations

import ctypes
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import NamedTuple, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import FilterParams, RibophaseError, make_filter  # noqa: F401  (make_filter: re-exported, it lives beside FilterParams)


class PhaseScores(NamedTuple):
    """Per-ORF results, device tensors of length n_orfs."""

    phase: torch.Tensor  # float64  np.sqrt(coh), statistics.py:115
    valid: torch.Tensor  # int32    valid codons of the winning frame
    read_count: torch.Tensor  # int64    sum(cov), detect_orfs.py:278
    min_codon_cov: torch.Tensor  # int32    min codon sum (MIN_CODON_COV_EMPTY if L == 0)
    flags: torch.Tensor  # uint8    FLAG_TIE | FLAG_RECHECK64 | FLAG_SPLIT
    status: Optional[torch.Tensor]  # uint8    1 = translating (None when no thresholds given)

    def cpu_numpy(self, pinned: Optional[dict] = None) -> dict:
        """Host numpy arrays.  ``pinned``: a dict the caller keeps (one per index): the arrays are then views of pinned
        staging tensors kept in it -- the copies run at PCIe speed instead of through pageable memory (290 MB for 11 M
        ORFs: 6 ms instead of 29) -- and the NEXT call with the same dict overwrites them: for a caller that is done with
        one sample's results before it scores the next (export_orf_coverages)."""
        if pinned is None:
            return {k: (None if v is None else v.cpu().numpy()) for k, v in self._asdict().items()}
        out = {}
        dev = None
        for k, v in self._asdict().items():
            if v is None:
                out[k] = None
                continue
            dev = v.device
            host = pinned.get(k)
            if host is None or host.numel() != v.numel() or host.dtype != v.dtype:
                host = pinned[k] = torch.empty(v.numel(), dtype=v.dtype, pin_memory=True)
            host.copy_(v.reshape(-1), non_blocking=True)
            out[k] = host.numpy().reshape(tuple(v.shape))
        if dev is not None and dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        return out


class FrameDiagnostics(NamedTuple):
    score: torch.Tensor  # float64 [n,3]  (NaN when M == 0 < N, 0 when N == 0)
    n: torch.Tensor  # int32 [n,3]
    m: torch.Tensor  # int32 [n,3]


def _require_gpu() -> None:
    if not torch.cuda.is_available():
        # let the library produce its own diagnostic
        _lib.device_count()
        raise RibophaseError(-6, "no HIP device visible to PyTorch; the hip backend never falls back to the CPU "
                                 "(RIBOTRICER_AMD_BACKEND=cpu selects the host backend of export_orf_coverages / phasescore)")


def _as_device(x, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = np.ascontiguousarray(x)
        if x.flags.writeable:
            x = torch.from_numpy(x)
        else:  # (a read-only view of a parsed index: only ever copied FROM -- torch's "not writable" warning is not for this)
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                x = torch.from_numpy(x)
    elif not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if x.dtype != dtype:
        x = x.to(dtype)
    if x.device != device:
        # Asynchronous only from PINNED host memory.  A pageable source is often a temporary of the caller (a sub-table, a
        # slice): an asynchronous copy that outlived it would read freed memory.  CUDA stages pageable sources before it
        # returns; whether every ROCm path does is not something this package should depend on (round 6: one unexplained
        # after-the-fact verify failure whose device results were right and whose re-gathered comparison profiles were not).
        x = x.to(device, non_blocking=bool(x.device.type == "cpu" and x.is_pinned()))
    return x.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None or t.numel() == 0 else t.data_ptr())


def _alloc_outputs(dev: torch.device, n: int, with_status: bool) -> PhaseScores:
    return PhaseScores(
        torch.empty(n, dtype=torch.float64, device=dev),
        torch.empty(n, dtype=torch.int32, device=dev),
        torch.empty(n, dtype=torch.int64, device=dev),
        torch.empty(n, dtype=torch.int32, device=dev),
        torch.empty(n, dtype=torch.uint8, device=dev),
        torch.empty(n, dtype=torch.uint8, device=dev) if with_status else None,
    )


class TilePlan:
    """A tile plan (``rp_plan``): what the tile path derives from the offsets alone, built --
    and the offsets validated -- once per candidate-ORF index.  Owns its device memory."""

    def __init__(self, device: torch.device, offsets: torch.Tensor, total_nt: int, counts_phase: int, stream):
        n = offsets.numel() - 1
        self.device = device
        self.n_orfs = n
        self.total_nt = total_nt
        self.counts_phase = counts_phase
        self.offsets = offsets  # keeps the storage alive: its address cannot be recycled under the cache
        self.offsets_version = offsets._version
        self._mem = torch.empty(_lib.plan_bytes(n, total_nt), dtype=torch.uint8, device=device)
        handle = ctypes.c_void_p(0)
        _lib.check(
            _lib.load().rp_plan_create_dev(
                device.index, _ptr(offsets), n, total_nt, counts_phase, _ptr(self._mem), self._mem.numel(),
                stream, ctypes.byref(handle),
            )
        )
        self.handle = handle

    def matches(self, offsets: torch.Tensor, total_nt: int, counts_phase: int) -> bool:
        return (
            offsets.data_ptr() == self.offsets.data_ptr()
            and offsets.numel() == self.offsets.numel()
            and offsets._version == self.offsets_version == self.offsets._version
            and total_nt == self.total_nt
            and counts_phase == self.counts_phase
        )

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _lib.load().rp_plan_free(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass


AUTO_WAVE_NT = 2 << 20  # RP_ALGO_AUTO: wave kernel below this many nucleotides (ribophase.hip)
MAX_CACHED_PLANS = 4


class PhaseScoreEngine:
    """Owns the reusable device buffers (outputs, workspaces, tile plans) for one GPU.

    Thread-safe for concurrent ``score`` calls on DIFFERENT streams: the workspace and the
    reusable outputs are kept per stream, the plan cache is locked."""

    def __init__(self, device=None):
        _lib.load()
        _require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("PhaseScoreEngine needs a cuda (HIP) device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._workspace: dict = {}  # stream handle -> uint8 tensor
        # > 1: a stream's workspace is allocated as a block of that many workspaces side by side (set BEFORE the first call);
        # the others serve further streams of the same batch (share_placed_workspace) -- and the block as first allocated
        # is then a candidate of tune_workspace like any other, so the search can only improve on it
        self.workspace_copies = 1
        self._placed_blocks: dict = {}  # stream handle -> the workspaces (views) of the block that holds the stream's workspace
        self._out: dict = {}  # stream handle -> PhaseScores
        self._plans: list = []  # most recently used last
        self._lock = threading.Lock()

    # -- buffers -----------------------------------------------------------------
    def _get_workspace(self, nbytes: int, stream_key: int) -> Optional[torch.Tensor]:
        if nbytes == 0:
            return None
        ws = self._workspace.get(stream_key)
        if ws is None or ws.numel() < nbytes:
            copies = max(1, int(self.workspace_copies))
            if copies == 1:
                ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                self._placed_blocks.pop(stream_key, None)
            else:
                stride = (nbytes + 4095) & ~4095
                block = torch.empty(stride * copies, dtype=torch.uint8, device=self.device)
                self._placed_blocks[stream_key] = [block[k * stride : k * stride + nbytes] for k in range(copies)]
                ws = self._placed_blocks[stream_key][0]
            self._workspace[stream_key] = ws
        return ws

    def tune_workspace(self, counts, offsets=None, thresholds: Optional[FilterParams] = None, tries: int = 6,
                       spacer_gib: float = 8.0, launches: int = 4, spread: float = 0.05, gather_plan=None,
                       release: bool = True, copies: int = 1) -> dict:
        """Once per engine and index: place the current stream's record workspace where its writes do not share a
        class of physical memory with the counts they ride beside -- and hold nothing else afterwards.

        On MI355X a write stream costs a read stream ~10 % when the two buffers lie in different classes of the
        physical address space (runs of 16-32 GiB) and ~23 % when they share one; the tile kernel writes its
        segment records while it streams the counts and takes 2.6 or 3.0 ms per 4 G nt accordingly (DESIGN.md
        section 4, profiles/archive/r03_probe_rw_regions.txt).  HIP has no placement hint and a fresh allocation usually
        lands next to the previous one, so candidates are made one after the other -- each exactly the size the
        batch needs, each behind a SPACER of ``spacer_gib`` that walks the next one through physical memory -- the
        scoring step of THIS batch is timed on each, and the search stops once a new candidate beats the slowest
        seen by ``spread`` (or after ``tries``).  The fastest stays; every spacer and every other candidate is
        freed and, with ``release`` (default), handed back to the driver (``torch.cuda.empty_cache()``): what
        remains reserved is the workload plus ONE workspace.  (The driver wipes freed memory in the background,
        which costs the kernels of the next second 1-4 %.)  Never slower than before (the workspace as first
        allocated is a candidate); ~0.1 s.  The drop-in export runs it once per cached index
        (``detect_orfs.score_index``), ``bench.py`` once before its timed steps.
        With ``gather_plan`` the fused path is tuned instead: ``counts`` is then the dense coverage
        (:meth:`score_coverage`).  ``copies`` > 1: every candidate is one block of that many workspaces side by side
        (the step is timed on the first); the others of the chosen block are kept for further streams of the same
        batch (:meth:`share_placed_workspace`: two samples in flight on two streams write their records into the
        SAME class of physical memory the search found, instead of each stream searching on its own and one of them
        ending up where the first allocation put it).  Returns what it measured."""
        dev = self.device
        if gather_plan is None:
            counts = _as_device(counts, torch.int32, dev)
            offsets = _as_device(offsets, torch.int64, dev)
            n, total_nt = offsets.numel() - 1, counts.numel()

            def run(t=None):
                self.score(counts, offsets, thresholds=thresholds, algo="tile", reuse_outputs=True, timings=t)
        else:  # the fused path: `counts` is the dense coverage, the profiles are read through the gather plan
            n, total_nt = gather_plan.n_orfs, gather_plan.total_nt

            def run(t=None):
                self.score_coverage(counts, gather_plan, thresholds=thresholds, reuse_outputs=True, timings=t)
        stream_key = int(torch.cuda.current_stream(dev).cuda_stream)
        need = _lib.workspace_bytes(n, total_nt, _lib.RP_ALGO_TILE)
        copies = max(1, int(copies))
        copy_stride = (need + 4095) & ~4095  # (each copy starts on a 4 KiB boundary of the block)
        block_bytes = need if copies == 1 else copy_stride * copies
        what = ("engine.tune_workspace: candidate record workspaces of the batch's own size, each behind a spacer allocation, "
                "the scoring step timed on each; spacers and rejected candidates freed")
        if n <= 0 or need == 0:
            return {"step_ms": [], "chosen": None, "spacer_gib": spacer_gib, "what": what, "skipped": "empty batch"}

        def step_ms():
            t: list = []
            run()
            for _ in range(launches):
                run(t)
            return sorted(x[1] + x[2] for x in t)[len(t) // 2]

        lib = _lib.load()
        was = lib.rp_measurement_tag(1)  # the search's launches run under a second kernel name (profilers)
        first = first_views = None
        candidates: list = []
        spacers: list = []
        times: list = []
        oom = False
        try:
            for _ in range(2):  # plan, outputs, first workspace; clocks up
                run()
            first = self._workspace.get(stream_key)
            if first is None:
                first = self._get_workspace(need, stream_key)
            # candidate 0 is the workspace AS FIRST ALLOCATED, whatever its shape (a block of `copies` when the engine
            # was told so before its first call -- workspace_copies -- else the single workspace: if that one wins, further
            # streams allocate their own): the search may only improve on what the process had
            first_views = self._placed_blocks.get(stream_key)
            candidates.append(first)
            times.append(step_ms())
            if release:  # no cached block may serve a candidate: each must be a fresh allocation behind its spacer
                torch.cuda.synchronize(dev)
                torch.cuda.empty_cache()
            while len(candidates) <= tries and not (len(times) > 1 and times[-1] <= (1.0 - spread) * max(times)):
                try:
                    # (three candidates without a gain: longer strides -- a 16 GB counts array can span two classes,
                    # and the third one may start 40 GiB further on)
                    stride = spacer_gib * (2.0 if len(candidates) > 3 else 1.0)
                    spacers.append(torch.empty(int(stride * (1 << 30)), dtype=torch.uint8, device=dev))
                    cand = torch.empty(block_bytes, dtype=torch.uint8, device=dev)
                except torch.cuda.OutOfMemoryError:
                    oom = True
                    break
                self._workspace[stream_key] = cand[:need] if copies > 1 else cand
                candidates.append(cand)
                times.append(step_ms())
        finally:
            torch.cuda.synchronize(dev)
            lib.rp_measurement_tag(was)
            candidates = candidates[: len(times)] if times else candidates[:1]
            best = min(range(len(times)), key=times.__getitem__) if times else 0
            if times and times[best] > 0.99 * times[0]:
                best = 0  # nothing to gain on this box: stay where the first allocation put it (no move for noise)
            if candidates:  # (an exception before the first timing leaves the workspace as it was)
                chosen = candidates[best]
                if best == 0:  # the first allocation stays, with whatever block it belongs to
                    self._workspace[stream_key] = chosen
                    if first_views:
                        self._placed_blocks[stream_key] = first_views
                    else:
                        self._placed_blocks.pop(stream_key, None)
                elif copies > 1 and chosen.numel() >= block_bytes:
                    self._workspace[stream_key] = chosen[:need]
                    self._placed_blocks[stream_key] = [chosen[k * copy_stride : k * copy_stride + need] for k in range(copies)]
                else:
                    self._workspace[stream_key] = chosen
                    self._placed_blocks.pop(stream_key, None)
                del chosen
            n_spacers = len(spacers)
            del spacers, candidates, first, first_views
            if release:
                torch.cuda.empty_cache()
        return {"step_ms": [round(t, 4) for t in times], "chosen": best, "spacer_gib": spacer_gib, "spacers": n_spacers,
                "workspace_bytes": need, "copies": copies, "out_of_memory": oom, "released_to_driver": bool(release),
                "reserved_bytes_after": int(torch.cuda.memory_reserved(dev)), "what": what}

    def tune_source(self, counts, offsets=None, thresholds: Optional[FilterParams] = None, tries: int = 3,
                    spacer_gib: float = 8.0, launches: int = 4, good_gbps: Optional[float] = 5950.0, gather_plan=None,
                    max_bytes: int = 40 << 30, release: bool = True):
        """The other half of the placement: where the COUNTS lie.  ``tune_workspace`` walks the record workspace through
        physical memory; what it cannot change is the class of memory the read stream itself comes from, and a step is
        slow whenever the two share one (one process, six copies of the same 16 GB of counts against two workspaces:
        2.63-3.04 ms per launch, some copies slow with every workspace -- profiles/archive/r05_source_placement.txt).  For an
        input the CALLER OF THIS METHOD owns and may swap for a copy (no product path calls this; bench.py times it AFTER its headline as
        value_source_placed) the
        same search is run on the source: unless the scoring kernel already moves its bytes at ``good_gbps`` (the fast
        class on MI355X: 6.0-6.1 TB/s of algorithmic bytes for the CSR kernel; ``None``: no absolute mark -- stop at a
        copy 5 % faster than the slowest seen) copies of ``counts`` are made one after the other, each behind a spacer
        allocation, the step is timed on each with the stream's current workspace, and the search stops at the first
        copy that reaches the mark or after ``tries``.  Returns ``(kept, info)``: ``kept`` is
        ``counts`` itself or the fastest copy (same bytes; the caller drops its other reference), everything else is
        freed.  Inputs above ``max_bytes`` are left where they are.  With ``gather_plan``: ``counts`` is the coverage
        of the fused path."""
        dev = self.device
        if gather_plan is None:
            counts = _as_device(counts, torch.int32, dev)
            offsets = _as_device(offsets, torch.int64, dev)
            n, total_nt = offsets.numel() - 1, counts.numel()

            def run(src, t=None):
                self.score(src, offsets, thresholds=thresholds, algo="tile", reuse_outputs=True, timings=t)
        else:
            counts = _as_device(counts, torch.int32, dev)
            n, total_nt = gather_plan.n_orfs, gather_plan.total_nt

            def run(src, t=None):
                self.score_coverage(src, gather_plan, thresholds=thresholds, reuse_outputs=True, timings=t)
        what = ("engine.tune_source: copies of the input array, each behind a spacer allocation, the scoring step timed on each "
                "with the stream's workspace; the fastest kept (same bytes), the rest freed")
        nbytes = counts.numel() * 4
        algorithmic = 4.0 * total_nt + 32.0 * n  # (counts in; offsets in and outputs out per ORF -- bench.py's figure)
        if n <= 0 or nbytes == 0 or nbytes > max_bytes:
            return counts, {"step_ms": [], "kernel_gbps": [], "chosen": 0, "what": what, "skipped": "empty or larger than max_bytes"}

        def measure(src):
            t: list = []
            run(src)
            for _ in range(launches):
                run(src, t)
            t.sort(key=lambda x: x[1] + x[2])
            mid = t[len(t) // 2]
            return mid[1] + mid[2], algorithmic / (mid[1] * 1e6)

        lib = _lib.load()
        was = lib.rp_measurement_tag(1)
        candidates = [counts]
        spacers: list = []
        steps: list = []
        rates: list = []
        oom = False
        try:
            run(counts)
            # (a search that has just handed memory back -- tune_workspace -- leaves the driver wiping it in the background,
            # 1-4 % off every kernel for up to a second: measure the copy in hand once batches of ten stop getting faster)
            t_calm, prev = time.perf_counter(), None
            while time.perf_counter() - t_calm < 1.5:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(counts)
                e1.record()
                torch.cuda.synchronize(dev)
                now = e0.elapsed_time(e1)
                if prev is not None and now >= 0.997 * prev:
                    break
                prev = now
            ms, gbps = measure(counts)
            steps.append(ms)
            rates.append(gbps)
            def done():  # an absolute mark when the caller names one, else the workspace search's rule: a copy 5 % faster than the slowest seen
                if good_gbps is not None:
                    return rates[-1] >= good_gbps
                return len(steps) > 1 and steps[-1] <= 0.95 * max(steps)

            while not done() and len(candidates) <= tries:
                try:
                    spacers.append(torch.empty(int(spacer_gib * (1 << 30)), dtype=torch.uint8, device=dev))
                    cand = counts.clone()
                except torch.cuda.OutOfMemoryError:
                    oom = True
                    break
                candidates.append(cand)
                ms, gbps = measure(cand)
                steps.append(ms)
                rates.append(gbps)
        finally:
            torch.cuda.synchronize(dev)
            lib.rp_measurement_tag(was)
        best = min(range(len(steps)), key=steps.__getitem__)
        if steps[best] > 0.99 * steps[0]:
            best = 0  # (no move for noise)
        kept = candidates[best]
        n_spacers = len(spacers)
        del spacers, candidates
        if release and n_spacers:
            torch.cuda.empty_cache()
        return kept, {"step_ms": [round(x, 4) for x in steps], "kernel_gbps": [round(x, 1) for x in rates], "chosen": best,
                      "good_gbps": good_gbps, "spacer_gib": spacer_gib, "spacers": n_spacers, "out_of_memory": oom,
                      "source_bytes": nbytes, "what": what}

    def share_placed_workspace(self, streams, source_stream=None) -> int:
        """Give ``streams[k]`` the k-th workspace of the block that ``tune_workspace(copies=...)`` placed for
        ``source_stream`` (default: the current stream; its own workspace is copy 0).  Samples in flight on several
        streams then all write their records where the search found the write stream cheapest.  Returns how many
        streams were served (0: no such block -- the streams allocate their own on first use)."""
        src = int((source_stream if source_stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)
        block = self._placed_blocks.get(src)
        if not block:
            return 0
        served = 0
        for k, st in enumerate(streams):
            if k < len(block):
                self._workspace[int(st.cuda_stream)] = block[k]
                served += 1
        return served

    def _get_outputs(self, n: int, with_status: bool, stream_key: int) -> PhaseScores:
        o = self._out.get(stream_key)
        if o is None or o.phase.numel() != n or (with_status and o.status is None):
            o = _alloc_outputs(self.device, n, with_status)
            self._out[stream_key] = o
        if not with_status and o.status is not None:
            o = o._replace(status=None)
        return o

    def plan_for(self, offsets: torch.Tensor, total_nt: int, counts_phase: int = 0) -> TilePlan:
        """The cached plan for this offsets tensor (same storage, unmodified), or a new one.

        Building a plan validates the offsets (RibophaseError -3 on a bad index) and
        synchronises the current stream once."""
        with self._lock:
            for k, p in enumerate(self._plans):
                if p.matches(offsets, total_nt, counts_phase):
                    self._plans.append(self._plans.pop(k))
                    return p
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        p = TilePlan(self.device, offsets, total_nt, counts_phase, stream)
        with self._lock:
            self._plans.append(p)
            del self._plans[:-MAX_CACHED_PLANS]
        return p

    # -- the hot path --------------------------------------------------------------
    def score(
        self,
        counts,
        offsets,
        thresholds: Optional[FilterParams] = None,
        algo: str = "auto",
        reuse_outputs: bool = False,
        timings: Optional[list] = None,
        plan="auto",
    ) -> PhaseScores:
        """Score every ORF of a CSR batch (counts int32, offsets int64, offsets[-1] == len(counts)).

        Asynchronous on the current torch stream unless ``timings`` is a list, in
        which case the call blocks and appends [index_ms, main_ms, finish_ms, total_ms].

        ``plan``: "auto" (default) keeps a tile plan per offsets TENSOR -- the first call on
        a device-resident index builds and validates it (one stream sync), later calls with
        the same, unmodified tensor skip the index pass, which is how ``detect-orfs`` uses
        one index for many samples; ``None`` rebuilds the tile index inside every call; a
        :class:`TilePlan` uses that plan.  Host (numpy) inputs are checked on the host and
        never planned.
        """
        dev = self.device
        host_offsets = offsets if isinstance(offsets, np.ndarray) else None
        counts = _as_device(counts, torch.int32, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        if offsets.dim() != 1 or offsets.numel() < 1 or counts.dim() != 1:
            raise ValueError("counts and offsets must be 1-D; offsets needs n_orfs+1 entries")
        n = offsets.numel() - 1
        total_nt = counts.numel()
        algo_id = _lib.ALGOS[algo]
        tile_path = algo == "tile" or (algo == "auto" and total_nt >= AUTO_WAVE_NT) or isinstance(plan, TilePlan)
        if host_offsets is not None:
            # the caller's index is on the host: check the one thing the kernels rely on for
            # in-bounds reads without touching the device
            if int(host_offsets[-1]) != total_nt or int(host_offsets[0]) != 0:
                raise RibophaseError(-3, f"offsets must run from 0 to len(counts)={total_nt}, got {int(host_offsets[0])}..{int(host_offsets[-1])}")
            if plan == "auto":
                plan = None
        elif plan == "auto":
            if tile_path and n > 0:
                plan = self.plan_for(offsets, total_nt, (counts.data_ptr() // 4) % 4 if total_nt else 0)
            else:  # small device-resident batch on the wave path: one scalar read-back
                if int(offsets[-1]) != total_nt:
                    raise RibophaseError(-3, f"offsets[-1]={int(offsets[-1])} but len(counts)={total_nt}")
                plan = None
        stream_obj = torch.cuda.current_stream(dev)
        stream_key = int(stream_obj.cuda_stream)
        stream = ctypes.c_void_p(stream_key)
        ws = self._get_workspace(_lib.workspace_bytes(n, total_nt, _lib.RP_ALGO_TILE if tile_path else algo_id), stream_key)
        with_status = thresholds is not None
        out = self._get_outputs(n, with_status, stream_key) if reuse_outputs else _alloc_outputs(dev, n, with_status)
        lib = _lib.load()
        outputs = [_ptr(out.phase), _ptr(out.valid), _ptr(out.read_count), _ptr(out.min_codon_cov), _ptr(out.flags), _ptr(out.status)]
        filt = ctypes.byref(thresholds) if thresholds is not None else None
        ws_args = [_ptr(ws), 0 if ws is None else ws.numel()]
        if timings is not None:
            ms = (ctypes.c_float * 4)()
            _lib.check(
                lib.rp_phase_score_csr_dev_timed(
                    dev.index, _ptr(counts), _ptr(offsets), n, total_nt, *outputs, filt, *ws_args,
                    _lib.RP_ALGO_TILE if plan is not None else algo_id, plan.handle if plan is not None else None,
                    stream, ctypes.byref(ms),
                )
            )
            timings.append([float(x) for x in ms])
        elif plan is not None:
            if plan.device != dev or plan.n_orfs != n:
                raise ValueError("plan belongs to another device / index")
            _lib.check(lib.rp_phase_score_csr_plan_dev(plan.handle, _ptr(counts), _ptr(offsets), *outputs, filt, *ws_args, stream))
        else:
            _lib.check(lib.rp_phase_score_csr_dev(dev.index, _ptr(counts), _ptr(offsets), n, total_nt, *outputs, filt, *ws_args, algo_id, stream))
        return out

    def score_coverage(
        self,
        coverage,
        gather_plan,
        thresholds: Optional[FilterParams] = None,
        reuse_outputs: bool = False,
        timings: Optional[list] = None,
        tile_plan: Optional[TilePlan] = None,
    ) -> PhaseScores:
        """Fused gather + score (``rp_phase_score_coverage_dev``): every ORF of the index behind
        ``gather_plan`` (:class:`ribotricer_amd.gather.GatherPlan`) scored straight from the dense
        coverage -- the CSR counts array of detect_orfs.py:134-203 is never materialised.  Same
        results, bit for bit, as ``gather_profiles_device`` + :meth:`score`."""
        dev = self.device
        if gather_plan.device != dev:
            raise ValueError("gather plan belongs to another device")
        coverage = _as_device(coverage, torch.int32, dev)
        offsets = gather_plan.offsets
        n, total_nt = gather_plan.n_orfs, gather_plan.total_nt
        plan = (tile_plan if tile_plan is not None else self.plan_for(offsets, total_nt, 0)) if n > 0 else None
        if plan is not None and not plan.matches(offsets, total_nt, 0):
            raise ValueError("tile plan belongs to another index")
        stream_obj = torch.cuda.current_stream(dev)
        stream_key = int(stream_obj.cuda_stream)
        ws = self._get_workspace(_lib.workspace_bytes(n, total_nt, _lib.RP_ALGO_TILE), stream_key)
        with_status = thresholds is not None
        out = self._get_outputs(n, with_status, stream_key) if reuse_outputs else _alloc_outputs(dev, n, with_status)
        outputs = [_ptr(out.phase), _ptr(out.valid), _ptr(out.read_count), _ptr(out.min_codon_cov), _ptr(out.flags), _ptr(out.status)]
        filt = ctypes.byref(thresholds) if thresholds is not None else None
        ms = (ctypes.c_float * 4)() if timings is not None else None
        _lib.check(
            _lib.load().rp_phase_score_coverage_dev(
                dev.index, _ptr(coverage), coverage.numel(), _ptr(offsets), n, total_nt, *outputs, filt,
                _ptr(ws), 0 if ws is None else ws.numel(), plan.handle if plan is not None else None,
                gather_plan.handle, ctypes.c_void_p(stream_key), ctypes.byref(ms) if ms is not None else None,
            )
        )
        if timings is not None:
            timings.append([float(x) for x in ms])
        return out

    def score_sharded(self, counts, offsets, devices, thresholds: Optional[FilterParams] = None, algo: str = "auto") -> dict:
        """See :func:`score_sharded` (this engine's device is not special)."""
        return score_sharded(counts, offsets, devices, thresholds=thresholds, algo=algo)

    def validate(self, counts, offsets) -> None:
        """Synchronous input check; raises RibophaseError on bad offsets / counts."""
        dev = self.device
        counts = _as_device(counts, torch.int32, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(
            _lib.load().rp_validate_csr_dev(
                dev.index, _ptr(counts), _ptr(offsets), offsets.numel() - 1, counts.numel(), stream
            )
        )

    def frames(self, counts, offsets) -> FrameDiagnostics:
        """Per-frame float64 (score_f, N_f, M_f) -- what statistics.py:67-108 computes per frame."""
        dev = self.device
        counts = _as_device(counts, torch.int32, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        n = offsets.numel() - 1
        d = FrameDiagnostics(
            torch.empty((n, 3), dtype=torch.float64, device=dev),
            torch.empty((n, 3), dtype=torch.int32, device=dev),
            torch.empty((n, 3), dtype=torch.int32, device=dev),
        )
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(
            _lib.load().rp_phase_score_frames_dev(
                dev.index, _ptr(counts), _ptr(offsets), n, _ptr(d.score), _ptr(d.n), _ptr(d.m), stream
            )
        )
        return d

    def score_float_profiles(self, values, offsets):
        """float64 profiles (metagene.py:243-244 -> statistics.py:48): (phase, valid, flags)."""
        dev = self.device
        values = _as_device(values, torch.float64, dev)
        offsets = _as_device(offsets, torch.int64, dev)
        n = offsets.numel() - 1
        phase = torch.empty(n, dtype=torch.float64, device=dev)
        valid = torch.empty(n, dtype=torch.int32, device=dev)
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(
            _lib.load().rp_phase_score_f64_csr_dev(
                dev.index, _ptr(values), _ptr(offsets), n, _ptr(phase), _ptr(valid), _ptr(flags), stream
            )
        )
        return phase, valid, flags


    def score_host(self, counts, offsets, thresholds: Optional[FilterParams] = None, algo: str = "auto", plan="auto",
                   pinned: Optional[dict] = None) -> dict:
        """:meth:`score`, waited for and brought to the host as a dict of numpy arrays, with the
        exact frame ties the device could not finish resolved (:func:`resolve_big_ties`): every
        ``phase`` / ``valid`` then carries the reference's bits wherever a tie is flagged.  ORFs that hold a
        count beyond ``RP_MAX_COUNT`` are finished in float64 / int64 (:func:`fix_big_counts_csr`;
        ``RP_FLAG_BIGCOUNT``): no count the int32 array can hold is refused."""
        d_counts = _as_device(counts, torch.int32, self.device)
        res = self.score(d_counts, offsets, thresholds=thresholds, algo=algo, plan=plan)
        host = res.cpu_numpy(pinned)  # (``pinned``: PhaseScores.cpu_numpy -- staging tensors the caller keeps and lets the next call overwrite)
        torch.cuda.synchronize(self.device)
        resolve_big_ties(host, csr_profiles_of(counts, offsets), thresholds)
        fix_big_counts_csr(host, d_counts, offsets, thresholds)  # counts beyond 2^24 - 1: those ORFs again, in float64 / int64
        return host

    def release_stream(self, stream) -> None:
        """Drop the workspace / reusable outputs kept for ``stream`` (a torch stream that is about to
        go away: the sharded helpers create one per slice)."""
        key = int(stream.cuda_stream)
        self._workspace.pop(key, None)
        self._out.pop(key, None)
        self._placed_blocks.pop(key, None)


def status_host(thresholds: FilterParams, phase, valid, read_count, min_codon_cov, lengths) -> np.ndarray:
    """The status predicate of detect_orfs.py:281,285-299 on host arrays (the same IEEE divisions
    and comparisons the device makes) -- for the few ORFs whose phase / valid_codons are patched on
    the host (:func:`resolve_big_ties`)."""
    n_codons = np.maximum(1, np.asarray(lengths, np.int64) // 3)
    ok = (
        (np.asarray(phase) >= thresholds.phase_score_cutoff)
        & (np.asarray(valid) >= thresholds.min_valid_codons)
        & (np.asarray(min_codon_cov).astype(np.float64) >= thresholds.min_reads_per_codon)
        & (np.asarray(valid) / n_codons >= thresholds.min_valid_codons_ratio)
        & (np.asarray(read_count) / n_codons >= thresholds.min_density_over_orf)
    )
    return ok.astype(np.uint8)


def resolve_big_ties(res: dict, profiles_of, thresholds: Optional[FilterParams] = None) -> int:
    """Finish the exact frame ties the device could not (``RP_FLAG_BIGTIE``: the tie involves a
    count >= 16, where the reference's ``real**2 + image**2`` -- statistics.py:83 -- goes through
    the host C library's ``pow``): their profiles are fetched with ``profiles_of(orf_ids) ->
    (counts, offsets)`` (host CSR), replayed by ``rp_tie_replay_host`` with this host's libm, and
    ``phase`` / ``valid`` (and ``status``, given ``thresholds``) of ``res`` -- host arrays -- are
    patched in place.  Returns how many ORFs that was (none on typical data: such an ORF is
    sparse AND piles >= 16 reads on one codon)."""
    idx = np.flatnonzero(res["flags"] & _lib.FLAG_BIGTIE)
    if idx.size == 0:
        return 0
    counts, offsets = profiles_of(idx)
    phase, valid = _lib.tie_replay_host(counts, offsets)
    res["phase"][idx] = phase
    res["valid"][idx] = valid
    if thresholds is not None and res.get("status") is not None:
        res["status"][idx] = status_host(thresholds, phase, valid, res["read_count"][idx], res["min_codon_cov"][idx], np.diff(offsets))
    return int(idx.size)


def rescore_big_count_orfs(res: dict, orf_ids, profiles_of, thresholds: Optional[FilterParams] = None, device=None) -> int:
    """Finish the ORFs that hold a count beyond ``RP_MAX_COUNT`` = 2^24 - 1 (a saturated position: rRNA / tRNA
    pile-ups).  The reference has no limit -- ``sum(cov)`` and ``phasescore(cov)`` work on Python ints
    (detect_orfs.py:278-280) -- but the scoring kernels' fp32 codon arithmetic and 32-bit codon sums are exact
    only up to there, so what they wrote for such an ORF is discarded: its profile (``profiles_of(orf_ids) ->
    (counts int32, offsets)``, host CSR) is scored again by the float64 kernel (``rp_phase_score_f64_csr_dev``:
    exact below 2^53; exact frame ties replayed with the reference's own arithmetic as everywhere), ``read_count``
    and the codon minimum are summed in int64 on the host, the status predicate is taken on those exact values,
    and ``res`` (host arrays) is patched in place; ``flags`` gets ``RP_FLAG_BIGCOUNT``.  ``min_codon_cov``
    (int32) saturates at 2^31 - 2 when a codon sum passes int32.  Returns how many ORFs that was."""
    orf_ids = np.asarray(orf_ids, np.int64)
    if orf_ids.size == 0:
        return 0
    counts, offsets = profiles_of(orf_ids)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    eng = get_engine(device)
    d_phase, d_valid, d_flags = eng.score_float_profiles(counts.astype(np.float64), offsets)
    torch.cuda.synchronize(eng.device)
    phase, valid, flags = d_phase.cpu().numpy(), d_valid.cpu().numpy(), d_flags.cpu().numpy()
    lengths = np.diff(offsets)
    tied = np.flatnonzero(flags & _lib.FLAG_TIE)
    if tied.size:  # the reference's strict `>` between equal frames: its own float64 arithmetic, replayed
        sub_off = np.zeros(tied.size + 1, np.int64)
        np.cumsum(lengths[tied], out=sub_off[1:])
        sub = np.concatenate([counts[offsets[i] : offsets[i + 1]] for i in tied]) if sub_off[-1] else np.zeros(0, np.int32)
        phase[tied], valid[tied] = _lib.tie_replay_host(sub, sub_off)
        flags[tied] |= _lib.FLAG_REPLAY
    read_count = np.zeros(orf_ids.size, np.int64)
    min_codon = np.full(orf_ids.size, _lib.MIN_CODON_COV_EMPTY, np.int64)
    for k in range(orf_ids.size):  # (a handful of ORFs; common.py:164-180: codon sums, the last one may be partial)
        prof = counts[offsets[k] : offsets[k + 1]].astype(np.int64)
        if prof.size:
            read_count[k] = prof.sum()
            min_codon[k] = np.pad(prof, (0, -prof.size % 3)).reshape(-1, 3).sum(axis=1).min()
    res["phase"][orf_ids] = phase
    res["valid"][orf_ids] = valid
    res["read_count"][orf_ids] = read_count
    res["min_codon_cov"][orf_ids] = np.where(lengths > 0, np.minimum(min_codon, _lib.MIN_CODON_COV_EMPTY - 1), _lib.MIN_CODON_COV_EMPTY).astype(np.int32)
    res["flags"][orf_ids] = (flags & (_lib.FLAG_TIE | _lib.FLAG_REPLAY)) | _lib.FLAG_BIGCOUNT
    if thresholds is not None and res.get("status") is not None:
        res["status"][orf_ids] = status_host(thresholds, phase, valid, read_count, min_codon, lengths)
    return int(orf_ids.size)


def fix_big_counts_csr(res: dict, counts, offsets, thresholds: Optional[FilterParams] = None) -> int:
    """:func:`rescore_big_count_orfs` for a CSR batch whose ``counts`` live on a device: one scan for entries
    beyond ``RP_MAX_COUNT`` (none on ordinary data), then the ORFs that hold one."""
    from .alignments import big_positions_device

    if not isinstance(counts, torch.Tensor) or not counts.is_cuda or counts.numel() == 0:
        return 0
    positions = big_positions_device(counts)
    if positions.size == 0:
        return 0
    off = offsets.cpu().numpy() if isinstance(offsets, torch.Tensor) else np.asarray(offsets, np.int64)
    ids = np.unique(np.searchsorted(off, positions, side="right") - 1)
    return rescore_big_count_orfs(res, ids, csr_profiles_of(counts, off), thresholds, counts.device)


def csr_profiles_of(counts, offsets):
    """``profiles_of`` for :func:`resolve_big_ties` over a CSR batch (host arrays or device tensors)."""

    def fetch(idx):
        off = offsets.cpu().numpy() if isinstance(offsets, torch.Tensor) else np.asarray(offsets)
        lens = off[idx + 1] - off[idx]
        out = np.zeros(idx.size + 1, np.int64)
        np.cumsum(lens, out=out[1:])
        parts = [counts[int(off[i]) : int(off[i + 1])] for i in idx]
        parts = [p.cpu().numpy() if isinstance(p, torch.Tensor) else np.asarray(p) for p in parts]
        return (np.concatenate(parts).astype(np.int32) if parts and out[-1] else np.zeros(0, np.int32)), out

    return fetch


def _wait_for_producers(stream, *inputs) -> None:
    """Make ``stream`` wait for the work already queued on the current stream of every device that
    holds one of ``inputs`` (device tensors produced by the caller moments ago): a fresh stream has
    no ordering with them otherwise."""
    import torch as _t

    seen = set()
    for x in inputs:
        if isinstance(x, _t.Tensor) and x.is_cuda and x.device not in seen:
            seen.add(x.device)
            stream.wait_stream(_t.cuda.current_stream(x.device))


_engines: dict = {}
_engines_lock = threading.Lock()


def get_engine(device=None) -> PhaseScoreEngine:
    """Process-wide engine per device."""
    _require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    with _engines_lock:
        if dev not in _engines:
            _engines[dev] = PhaseScoreEngine(dev)
        return _engines[dev]


def _devices(devices) -> list:
    devs = [torch.device(d) if not isinstance(d, int) else torch.device("cuda", d) for d in devices]
    if not devs:
        raise ValueError("sharded scoring needs at least one device")
    return devs


def _run_slices(work, n: int) -> list:
    if n == 1:
        return [work(0)]
    with ThreadPoolExecutor(max_workers=n) as pool:
        return list(pool.map(work, range(n)))


class CsrShards:
    """One candidate-ORF index (CSR ``offsets``) cut over several GPUs of this node, kept for every sample scored
    against it: the nt-balanced slice bounds (``sharding.slice_bounds``), and per slice a stream and the re-based
    offsets on the slice's device -- the tile plan hangs on that tensor in the device engine's plan cache
    (:meth:`PhaseScoreEngine.plan_for`), the record workspace and the reusable outputs on the stream.  The second
    sample therefore skips the offsets upload, the index pass and every allocation, as the one-GPU path does
    (detect_orfs._table_and_plan).  A device may appear more than once (two slices on one GPU, on two streams)."""

    def __init__(self, offsets, devices: Sequence):
        from .sharding import slice_bounds

        self.devices = _devices(devices)
        off = offsets.cpu().numpy() if isinstance(offsets, torch.Tensor) else np.ascontiguousarray(offsets, dtype=np.int64)
        if off.size < 1 or int(off[0]) != 0:
            raise RibophaseError(-3, "offsets must start at 0")
        self.offsets = off
        self.total_nt = int(off[-1])
        self.bounds = slice_bounds(off, len(self.devices))
        self._slices: list = [None] * len(self.devices)  # (stream, device offsets, tile plans), made by the slice's own thread
        self.plans_built = 0  # (diagnostics / tests: tile plans built so far, all slices)

    def matches(self, offsets, devices) -> bool:
        off = offsets.cpu().numpy() if isinstance(offsets, torch.Tensor) else np.asarray(offsets)
        return _devices(devices) == self.devices and off.shape == self.offsets.shape and np.array_equal(off, self.offsets)

    def _slice(self, k: int):
        if self._slices[k] is None:
            dev = self.devices[k]
            lo, hi = int(self.bounds[k]), int(self.bounds[k + 1])
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                o = torch.from_numpy(self.offsets[lo : hi + 1] - self.offsets[lo]).to(dev)  # (a temporary: a blocking copy)
            self._slices[k] = (stream, o, {})  # {counts phase: TilePlan}
        return self._slices[k]

    def score(self, counts, thresholds: Optional[FilterParams] = None, algo: str = "auto") -> dict:
        """Host numpy arrays, in ORF order, for one sample's ``counts`` (host array or a tensor on any device)."""
        from .sharding import concat_results

        n_counts = counts.numel() if isinstance(counts, torch.Tensor) else int(np.asarray(counts).size)
        if n_counts != self.total_nt:
            raise RibophaseError(-3, "offsets must run from 0 to len(counts)")
        off_host = self.offsets

        def work(k: int) -> dict:
            dev = self.devices[k]
            lo, hi = int(self.bounds[k]), int(self.bounds[k + 1])
            a, b = int(off_host[lo]), int(off_host[hi])
            eng = get_engine(dev)
            with torch.cuda.device(dev):
                stream, o, plans = self._slice(k)
                _wait_for_producers(stream, counts)
                with torch.cuda.stream(stream):
                    c = counts[a:b]
                    c = torch.from_numpy(np.ascontiguousarray(c, dtype=np.int32)) if not isinstance(c, torch.Tensor) else c
                    c = c.to(dev, non_blocking=bool(c.device.type != "cpu" or c.is_pinned()))
                    plan = "auto"
                    if algo == "tile" or (algo == "auto" and c.numel() >= AUTO_WAVE_NT):  # the tile path: this slice's plan, kept HERE
                        phase = (c.data_ptr() // 4) % 4 if c.numel() else 0  # (an engine's own cache holds 4 plans)
                        if phase not in plans and hi > lo:
                            plans[phase] = TilePlan(dev, o, c.numel(), phase, ctypes.c_void_p(stream.cuda_stream))
                            self.plans_built += 1
                        plan = plans.get(phase)
                    res = eng.score(c, o, thresholds=thresholds, algo=algo, reuse_outputs=True, plan=plan)
                    host = {k_: (None if v is None else v.cpu()) for k_, v in res._asdict().items()}
                stream.synchronize()
            out = {k_: (None if v is None else v.numpy()) for k_, v in host.items()}
            resolve_big_ties(out, csr_profiles_of(c, o), thresholds)
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                fix_big_counts_csr(out, c, o, thresholds)
            return out

        return concat_results(_run_slices(work, len(self.devices)))

    def release(self) -> None:
        for k, sl in enumerate(self._slices):
            if sl is not None:
                get_engine(self.devices[k]).release_stream(sl[0])
        self._slices = [None] * len(self.devices)


def score_sharded(counts, offsets, devices: Sequence, thresholds: Optional[FilterParams] = None, algo: str = "auto",
                  shards: Optional[CsrShards] = None) -> dict:
    """Score one CSR batch on several GPUs of this node: host numpy arrays back, in ORF order.

    The batch is cut into ``len(devices)`` contiguous ORF-index slices balanced on
    nucleotides (``sharding.slice_bounds``; ORFs are independent: the loop body of
    ribotricer/detect_orfs.py:274-324 carries no state between iterations), slice k is
    copied to ``devices[k]`` and scored there on a stream of its own by a thread of its own
    (ctypes releases the GIL for the duration of the library call), and the per-ORF results
    are concatenated on the host.  No collective, no peer traffic besides the slice copies.
    A device may appear more than once (two slices on one GPU, on two streams).

    ``shards``: a :class:`CsrShards` of the same offsets and devices, kept by a caller that scores many samples
    against one index (per-device offsets, tile plans, workspaces and outputs are then reused); without it the
    per-device state lives for this call only.
    """
    if shards is not None:
        return shards.score(counts, thresholds=thresholds, algo=algo)
    once = CsrShards(offsets, devices)
    try:
        return once.score(counts, thresholds=thresholds, algo=algo)
    finally:
        once.release()


class CoverageShards:
    """One interval table (``gather.IntervalTable``: candidate-ORF index + coverage layout) cut over several GPUs
    for the fused gather + score, kept for every sample: per slice the nt-balanced ORF range, the WINDOWS of the
    dense coverage its exons touch (``sharding.coverage_windows``: on a human-sized index an eighth of the 25 GB
    array per GPU of eight, not a full copy each), the slice's table re-based onto them and -- built by the first
    sample, on the slice's own thread and stream -- its gather plan and tile plan on the device.  A later sample
    uploads its windows and launches; nothing is planned again (the one-GPU path: detect_orfs._table_and_plan)."""

    def __init__(self, table, devices: Sequence, coverage_len: int):
        from .gather import IntervalTable, slice_orfs
        from .sharding import coverage_windows_native, slice_bounds

        self.devices = _devices(devices)
        self.coverage_len = int(coverage_len)
        self.n_orfs = int(len(table.offsets) - 1)
        self.bounds = slice_bounds(np.asarray(table.offsets, np.int64), len(self.devices))
        self.parts: list = []
        for k in range(len(self.devices)):
            lo, hi = int(self.bounds[k]), int(self.bounds[k + 1])
            sub = slice_orfs(table, lo, hi)  # (views: the slices of an 11 M-ORF index are not copied)
            windows = None
            if len(self.devices) > 1:  # only what this slice reads crosses to the device
                w_start, w_len, w_base, w_total, rebased = coverage_windows_native(sub.iv_start, sub.iv_len)
                windows = (w_start, w_len, w_base, max(w_total, 16))
                sub = IntervalTable(rebased, sub.iv_len, sub.orf_iv, sub.reverse, sub.offsets)
            self.parts.append({"table": sub, "windows": windows, "plan": None, "tile_plan": None, "stream": None})
        self.plans_built = 0

    def score(self, coverage, thresholds: Optional[FilterParams] = None) -> dict:
        from .gather import GatherPlan, coverage_profiles_of
        from .sharding import compact_coverage, concat_results

        n_cov = coverage.numel() if isinstance(coverage, torch.Tensor) else int(np.asarray(coverage).size)
        if n_cov != self.coverage_len:
            raise ValueError(f"coverage has {n_cov} positions, the shards were laid out for {self.coverage_len}")

        def work(k: int) -> dict:
            dev = self.devices[k]
            part = self.parts[k]
            eng = get_engine(dev)
            with torch.cuda.device(dev):
                if part["stream"] is None:
                    part["stream"] = torch.cuda.Stream(device=dev)
                stream = part["stream"]
                _wait_for_producers(stream, coverage)
                with torch.cuda.stream(stream):
                    if part["windows"] is not None:
                        cov = compact_coverage(coverage, *part["windows"], dev)
                    else:
                        cov = _as_device(coverage, torch.int32, dev)
                    if part["plan"] is None:
                        part["plan"] = gp = GatherPlan(part["table"], cov.numel(), dev)
                        part["tile_plan"] = TilePlan(dev, gp.offsets, gp.total_nt, 0, ctypes.c_void_p(stream.cuda_stream)) if gp.n_orfs else None
                        self.plans_built += 1
                    res = eng.score_coverage(cov, part["plan"], thresholds=thresholds, reuse_outputs=True, tile_plan=part["tile_plan"])
                    host = {k_: (None if v is None else v.cpu()) for k_, v in res._asdict().items()}
                stream.synchronize()
                out = {k_: (None if v is None else v.numpy()) for k_, v in host.items()}
                with torch.cuda.stream(stream):
                    resolve_big_ties(out, coverage_profiles_of(cov, part["table"], dev), thresholds)
                stream.synchronize()
            return out

        return concat_results(_run_slices(work, len(self.devices)))

    def release(self) -> None:
        for k, part in enumerate(self.parts):
            if part["stream"] is not None:
                get_engine(self.devices[k]).release_stream(part["stream"])
            part["plan"] = part["tile_plan"] = part["stream"] = None


def score_coverage_sharded(coverage, table, devices: Sequence, thresholds: Optional[FilterParams] = None,
                           shards: Optional[CoverageShards] = None) -> dict:
    """The fused gather + score (:meth:`PhaseScoreEngine.score_coverage`) on several GPUs of this
    node: ``table`` (``gather.IntervalTable``) is cut into nt-balanced contiguous ORF-index slices,
    every device gets the WINDOWS of the dense coverage that its slice's exons touch
    (``sharding.coverage_windows``) and the gather plan of its slice re-based onto them; the uploads of
    the devices run side by side (one thread and one stream per device), results are concatenated on
    the host.  No collective; the profiles exist on no device.  Raises ``RibophaseError`` (status
    ``ERR_INTERVALS``) for a table that cannot be planned.  ``shards``: a :class:`CoverageShards` kept by the
    caller across samples (windows, gather plans and tile plans are then built once per index)."""
    if shards is not None:
        return shards.score(coverage, thresholds=thresholds)
    n_cov = coverage.numel() if isinstance(coverage, torch.Tensor) else int(np.asarray(coverage).size)
    once = CoverageShards(table, devices, n_cov)
    try:
        return once.score(coverage, thresholds=thresholds)
    finally:
        once.release()


def phase_score_csr(counts, offsets, thresholds: Optional[FilterParams] = None, algo: str = "auto", device=None) -> PhaseScores:
    """Functional form of :meth:`PhaseScoreEngine.score`."""
    return get_engine(device).score(counts, offsets, thresholds=thresholds, algo=algo)
