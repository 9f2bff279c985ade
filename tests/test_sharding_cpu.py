"""N>1 path on CPU: nt-balanced ORF-index slices + host concat reproduce the unsharded
result (world_size-2 gloo run; per-rank scoring is done by the C oracle here because
there is no GPU -- the GPU twin of this test is test_gpu_parity.test_full_size_properties)."""

import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle
from ribotricer_amd.sharding import concat_results, gather_results, shard_csr, slice_bounds
from ribotricer_amd.synth import synth_csr_host


def _oracle_dict(counts, offsets):
    r = c_oracle.phase_score_csr(counts, offsets)
    return dict(phase=r.phase, valid=r.valid, read_count=r.read_count, min_codon_cov=r.min_codon_cov, flags=r.flags)


def test_slice_bounds_are_nt_balanced_and_monotone():
    counts, offsets = synth_csr_host(5000, seed=3, cfg="cfg5")
    for world in (1, 2, 4, 8):
        b = slice_bounds(offsets, world)
        assert b[0] == 0 and b[-1] == offsets.size - 1 and np.all(np.diff(b) >= 0)
        nt = np.diff(offsets[b])
        assert nt.sum() == offsets[-1]
        # no slice exceeds its fair share by more than the longest profile
        assert nt.max() <= offsets[-1] / world + np.diff(offsets).max()


def test_slice_bounds_edge_cases():
    assert list(slice_bounds(np.array([0]), 4)) == [0, 0, 0, 0, 0]
    assert list(slice_bounds(np.zeros(9, np.int64), 2)) == [0, 4, 8]
    b = slice_bounds(np.array([0, 100000, 100003, 100006]), 2)  # one giant ORF first
    assert list(b) == [0, 1, 3]


def test_shard_concat_equals_whole_numpy():
    counts, offsets = synth_csr_host(4000, seed=9, cfg="cfg3")
    whole = _oracle_dict(counts, offsets)
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            c, o, lo, hi = shard_csr(counts, offsets, world, r)
            assert o[0] == 0 and o[-1] == c.size and o.size == hi - lo + 1
            parts.append(_oracle_dict(c, o))
        cat = concat_results(parts)
        for k in whole:
            assert np.array_equal(cat[k], whole[k], equal_nan=True), k


def _worker(rank, world, port, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts, offsets = synth_csr_host(3000, seed=21, cfg="cfg2")  # same seeded batch on every rank
    c, o, lo, hi = shard_csr(torch.from_numpy(counts), torch.from_numpy(offsets), world, rank)
    local = _oracle_dict(c.numpy(), o.numpy())
    full = gather_results(local)
    whole = _oracle_dict(counts, offsets)
    good = all(np.array_equal(full[k], whole[k], equal_nan=True) for k in whole)
    t = torch.tensor([1 if good else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ok.value = int(t.item())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    world = 2
    ok = mp.get_context("spawn").Value("i", 0)
    mp.spawn(_worker, args=(world, 29531 + os.getpid() % 200, ok), nprocs=world, join=True)
    assert ok.value == 1
