"""One candidate-ORF index cut over the GPUs of a node for the drop-in export (BASELINE configs[3] as the product runs
it): ``export_orf_coverages(devices=[...])`` / ``RIBOTRICER_AMD_DEVICES``.

The reference's loop (ribotricer/detect_orfs.py:274-324) carries no state between ORFs, so the index is cut into
contiguous nt-balanced slices (``sharding.slice_bounds``) and every device does, for ITS slice and on its own thread and
stream, exactly what the one-GPU export does for the whole index -- nothing is built on one device and shipped to the
others (rounds 3-4 built the whole coverage on ``devices[0]`` and copied windows of it out, per sample, into fresh
buffers; the translating ORFs' profiles came back through a numpy sub-table on device 0):

  once per index   the slice's interval table (views of the index's), its compact-coverage block map (``gather.CoverageMap``
                   over the slice's exons only), its gather plan and tile plan, a coverage buffer of the slice's compact
                   length, pinned result staging -- built by the first sample, kept with the cached index
  per sample       the sample's alignment columns go up to every distinct device once (21 bytes per row, uploads side by
                   side); each slice accumulates its own compact coverage from them (``rp_coverage_build_rows_dev`` with
                   the slice's map: rows under none of ITS exons are dropped on the device), runs the fused gather + score,
                   copies its results into its range of the shared host arrays, and gathers the profiles the TSV prints
                   (all of them with ``report_all``, else the translating ORFs' through the plan's pieces) on its device

No collective, no peer traffic; the host concatenates nothing (every slice writes its own range).  The TSV writer then
streams the profile parts back device by device (``detect_orfs._profile_slices``).
"""

from __future__ import annotations

import ctypes
import threading
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import FilterParams

RESULT_DTYPES = {"phase": torch.float64, "valid": torch.int32, "read_count": torch.int64, "min_codon_cov": torch.int32,
                 "flags": torch.uint8, "status": torch.uint8}


class ColumnsOnDevices(dict):
    """``{torch.device: alignments.UploadedColumns}``: one sample's merged P-site columns on every device of a sharded export."""


def columns_on_devices(merged_alignments, devices) -> ColumnsOnDevices:
    """The sample's columns uploaded to every distinct device of ``devices``, the uploads side by side (threads; the
    copies release the GIL).  The reference's ``strand -> Counter`` is converted to columns once, not once per device."""
    from .alignments import MergedColumns, UploadedColumns, upload_columns
    from .engine import _devices, _run_slices

    distinct = list(dict.fromkeys(_devices(devices)))
    if isinstance(merged_alignments, UploadedColumns):
        have = {merged_alignments.device: merged_alignments}
        missing = [d for d in distinct if d not in have]
        if missing:
            raise ValueError("columns that live on one device cannot reach the others: pass MergedColumns or the reference's Counters")
        return ColumnsOnDevices(have)
    cols = merged_alignments if isinstance(merged_alignments, MergedColumns) else MergedColumns.from_counters(merged_alignments)

    def upload(i: int):
        with torch.cuda.device(distinct[i]):
            return upload_columns(cols, distinct[i])

    return ColumnsOnDevices(zip(distinct, _run_slices(upload, len(distinct))))


class CoverageLayout:
    """What ``alignments.build_coverage_device`` needs of an index -- the (strand, chrom) groups and their extents -- and
    nothing else: the shards hang on the cached index, so they must not hold the index (a reference cycle would keep
    gigabytes of per-slice device buffers alive until a cyclic GC pass)."""

    __slots__ = ("extents", "group_keys")

    def __init__(self, index):
        self.extents = dict(index.extents)
        self.group_keys = list(index.group_keys)


class IndexShards:
    """See the module docstring.  Lifetime: built by the first sharded sample of a cached index, kept in that index's
    ``_shard_cache`` (``detect_orfs.score_index``), released -- streams' workspaces and outputs, coverage buffers, maps,
    plans, pinned staging -- by :meth:`release` when another device list replaces it, when a build fails, and when the index
    leaves the index cache (``detect_orfs._forget_index``)."""

    def __init__(self, index, dense_table, dense_len: int, devices: Sequence):
        from .engine import _devices
        from .gather import slice_orfs
        from .sharding import slice_bounds

        self.layout = CoverageLayout(index)  # (NOT the index: see CoverageLayout)
        self.released = False
        self.devices = _devices(devices)
        self.dense_len = int(dense_len)
        self.n_orfs = int(len(dense_table.offsets) - 1)
        self.bounds = slice_bounds(np.asarray(dense_table.offsets, np.int64), len(self.devices))
        self.parts: list = []
        for k, dev in enumerate(self.devices):
            lo, hi = int(self.bounds[k]), int(self.bounds[k + 1])
            self.parts.append({"lo": lo, "hi": hi, "device": dev, "table": slice_orfs(dense_table, lo, hi) if hi > lo else None, "built": None})
        self.plans_built = 0  # (diagnostics / tests: slices whose map + plans were built so far)
        self._host: dict = {}  # result name -> pinned host tensor of n_orfs entries, every slice fills its own range
        self._lock = threading.Lock()

    # -- once per index and slice (on the slice's thread, inside its device / stream context) -----------------------
    def _build(self, part: dict, stream) -> dict:
        from .engine import TilePlan
        from .gather import CoverageMap, GatherPlan

        dev = part["device"]
        cmap = CoverageMap(part["table"], self.dense_len, dev)  # this slice's exons only; its table in compact coordinates
        plan = GatherPlan(cmap.table, cmap.compact_len, dev, device_intervals=cmap.device_intervals)
        cmap.release_device_intervals()
        built = {
            "cmap": cmap, "plan": plan,
            "tile_plan": TilePlan(dev, plan.offsets, plan.total_nt, 0, ctypes.c_void_p(stream.cuda_stream)) if plan.n_orfs else None,
            "coverage": torch.empty(max(16, cmap.compact_len), dtype=torch.int32, device=dev),  # reused by every sample (zeroed by the build)
            "lengths": np.diff(np.asarray(part["table"].offsets, np.int64)),
            "stream": stream,
        }
        with self._lock:
            self.plans_built += 1
        return built

    def _host_arrays(self, with_status: bool) -> dict:
        with self._lock:
            for name, dt in RESULT_DTYPES.items():
                if name not in self._host:
                    self._host[name] = torch.empty(self.n_orfs, dtype=dt, pin_memory=True)
        return self._host

    def score(self, merged_alignments, thresholds: Optional[FilterParams], report_all: bool, timings: Optional[dict] = None,
              reuse_result_buffers: bool = False):
        """One sample: ``(results, profile_parts)`` -- ``results`` the per-ORF host arrays of the whole index (views of
        pinned staging tensors kept here when ``reuse_result_buffers``, else copies), ``profile_parts`` a list of
        ``(device counts tensor, host offsets of the slice's ORFs, first ORF)`` in index order: the profiles the TSV
        prints, on the device that gathered them."""
        import time

        from .alignments import build_coverage_device
        from .engine import _run_slices, _wait_for_producers, get_engine, rescore_big_count_orfs, resolve_big_ties
        from .gather import coverage_profiles_of, orfs_touching

        if self.released:
            raise RuntimeError("these IndexShards were released: build new ones for the index")
        t0 = time.perf_counter()
        # the sample's columns on every distinct device (export_orf_coverages sends them up beside the index parse)
        ups = merged_alignments if isinstance(merged_alignments, ColumnsOnDevices) else columns_on_devices(merged_alignments, self.devices)
        if timings is not None:
            for dev in ups:
                torch.cuda.synchronize(dev)
            timings["columns_to_devices"] = timings.get("columns_to_devices", 0.0) + time.perf_counter() - t0
        host = self._host_arrays(thresholds is not None)
        stages = [dict() for _ in self.parts]

        def work(k: int):
            part = self.parts[k]
            lo, hi, dev = part["lo"], part["hi"], part["device"]
            if hi <= lo:
                return None
            eng = get_engine(dev)
            lap = time.perf_counter()
            with torch.cuda.device(dev):
                if part["built"] is None:
                    stream = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(stream):
                        part["built"] = self._build(part, stream)
                    stream.synchronize()
                    stages[k]["map_and_plans"] = time.perf_counter() - lap
                    lap = time.perf_counter()
                b = part["built"]
                stream = b["stream"]
                _wait_for_producers(stream, ups[dev].pos)
                with torch.cuda.stream(stream):
                    big: dict = {}
                    cov, _ = build_coverage_device(ups[dev], self.layout, dev, big=big, cmap=b["cmap"], out=b["coverage"])
                    res = eng.score_coverage(cov, b["plan"], thresholds=thresholds, reuse_outputs=True, tile_plan=b["tile_plan"])
                    for name, t in res._asdict().items():
                        if t is not None:
                            host[name][lo:hi].copy_(t, non_blocking=True)
                    stream.synchronize()
                    stages[k]["coverage_build_score_results"] = time.perf_counter() - lap
                    lap = time.perf_counter()
                    mine = {name: (host[name][lo:hi].numpy() if getattr(res, name) is not None else None) for name in RESULT_DTYPES}
                    profiles_of = coverage_profiles_of(cov, b["cmap"].table, dev)
                    resolve_big_ties(mine, profiles_of, thresholds)
                    if big["positions"].size:
                        rescore_big_count_orfs(mine, orfs_touching(b["cmap"].table, big["positions"]), profiles_of, thresholds, dev)
                    # the profiles the TSV prints (detect_orfs.py:301-324), gathered where the coverage lies
                    if report_all:
                        d_counts, offsets_k = b["plan"].gather(cov), np.asarray(part["table"].offsets, np.int64)
                    else:
                        keep = mine["status"] != 0 if mine["status"] is not None else np.ones(hi - lo, bool)  # (no thresholds: nothing is filtered)
                        d_counts, offsets_k = b["plan"].gather_selected(cov, keep, lengths=b["lengths"], reuse_arrays=reuse_result_buffers)
                    stream.synchronize()
                    stages[k]["ties_and_profiles"] = time.perf_counter() - lap
            return (d_counts, offsets_k, lo)

        parts = [p for p in _run_slices(work, len(self.parts)) if p is not None]
        if timings is not None:
            timings["shard_stages"] = stages
        with_status = thresholds is not None
        results = {name: (None if (name == "status" and not with_status) else
                          (host[name].numpy() if reuse_result_buffers else host[name].numpy().copy())) for name in RESULT_DTYPES}
        return results, parts

    def release(self) -> None:
        """Hand everything back: every slice's stream (the engine's workspace and outputs registered under it), coverage
        buffer, map and plans, and the pinned result staging.  Idempotent; the object cannot score afterwards."""
        from .engine import get_engine

        for part in self.parts:
            b = part["built"]
            if b is not None:
                b["stream"].synchronize()
                get_engine(part["device"]).release_stream(b["stream"])
                b.clear()
            part["built"] = None
        self._host = {}
        self.released = True
