"""Sharding of a candidate-ORF batch over the GPUs of one node.

ORFs are independent (the loop body of ribotricer/detect_orfs.py:274-324 carries no
state between iterations besides the output order), so the batch is cut into
contiguous ORF-index slices balanced on the prefix sum of profile lengths (nucleotides,
not ORF counts), each rank scores its slice on its own GPU, and the per-ORF result
arrays are concatenated on the host in slice order.  No collective is on the data
path; ``gather_results`` moves 26 bytes per ORF at the very end.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np


def slice_bounds(offsets: np.ndarray, world_size: int) -> np.ndarray:
    """ORF-index cut points ``b[0]=0 <= b[1] <= ... <= b[world]=n_orfs``, nt-balanced.

    Cut r is the first ORF whose start offset is >= r/world of the total length; long
    ORFs are never split across ranks.
    """
    offsets = np.asarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    total = int(offsets[-1])
    targets = (np.arange(1, world_size, dtype=np.float64) * total / world_size).astype(np.int64)
    cuts = np.searchsorted(offsets[:-1], targets, side="left") if n > 0 else np.zeros(world_size - 1, np.int64)
    if total == 0:  # only empty profiles: balance on ORF count instead
        cuts = (np.arange(1, world_size) * n) // world_size
    bounds = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def shard_csr(counts, offsets, world_size: int, rank: int):
    """Rank ``rank``'s slice as its own CSR batch: ``(counts_r, offsets_r, first_orf, last_orf)``.

    Works on numpy arrays and on torch tensors (views, no copy of ``counts``).
    """
    off_np = offsets.cpu().numpy() if hasattr(offsets, "cpu") else np.asarray(offsets)
    bounds = slice_bounds(off_np, world_size)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    a, b = int(off_np[lo]), int(off_np[hi])
    return counts[a:b], offsets[lo : hi + 1] - offsets[lo], lo, hi


def concat_results(parts: Sequence[dict]) -> dict:
    """Host-side concat of per-rank result dicts (numpy arrays) in slice order."""
    keys = [k for k, v in parts[0].items() if v is not None]
    return {k: np.concatenate([p[k] for p in parts]) for k in keys}


def gather_results(local: dict, group=None) -> dict:
    """All ranks receive the concatenated results (torch.distributed, any backend)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, {k: v for k, v in local.items() if v is not None}, group=group)
    return concat_results(parts)
