"""Seeded fuzz of the tile paths against the C oracle: batch shapes built around the tile
size (ORFs that end exactly on, one before and one after a tile boundary, runs of empty
ORFs on a boundary, ORFs spanning several tiles, many tiny ORFs per tile), random pointer
misalignment, counts up to RP_MAX_COUNT, lengths that are not multiples of 3."""

import numpy as np
import pytest
import torch

from helpers import assert_matches_oracle

pytestmark = pytest.mark.gpu

TILES = {"tile": 7936}


@pytest.fixture(scope="module")
def eng():
    from ribotricer_amd.engine import PhaseScoreEngine

    return PhaseScoreEngine("cuda:0")


def lengths_for(seed: int, tile: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    kind = seed % 6
    if kind == 0:  # boundary hugging: cumulative ends at tile*k + {-2..2}
        ends = np.sort(np.unique(np.concatenate([tile * np.arange(1, 9) + d for d in (-2, -1, 0, 1, 2)])))
        lens = np.diff(np.concatenate([[0], ends]))
    elif kind == 1:  # empty ORFs piled on boundaries and at both ends
        body = rng.integers(1, 900, 60)
        lens = np.concatenate([[0, 0], body[:20], [tile - int(body[:20].sum()) % tile], [0] * 5, body[20:], [0, 0, 0]])
    elif kind == 2:  # several multi-tile ORFs in a row, then short ones
        lens = np.concatenate([rng.integers(tile, 4 * tile, 4), rng.integers(1, 50, 300), [3 * tile + 1]])
    elif kind == 3:  # > 64 and > 128 segments per tile
        lens = rng.integers(1, 40, 3000)
    elif kind == 4:  # one ORF exactly one tile, one exactly two, neighbours of 1 and 2 nt
        lens = np.array([1, tile, 2, 2 * tile, 1, 1, tile - 1, tile + 1, 3, 5])
    else:  # plain ragged
        lens = np.clip(np.rint(rng.lognormal(np.log(240), 1.0, 400)), 0, 40000).astype(np.int64)
    return np.asarray(lens, np.int64)


@pytest.mark.parametrize("algo", ["tile"])
@pytest.mark.parametrize("seed", range(18))
def test_fuzz_against_oracle(eng, algo, seed):
    rng = np.random.default_rng(1000 + seed)
    lens = lengths_for(seed, TILES[algo])
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offsets[-1])
    lam = rng.choice([0.02, 0.3, 3.0], size=lens.size)
    counts = rng.poisson(np.repeat(lam, lens)).astype(np.int32)
    if seed % 4 == 1 and total:  # a few huge counts (exact in fp32 up to 2^24 - 1)
        counts[rng.integers(0, total, 20)] = 2**24 - 1
        counts[rng.integers(0, total, 20)] = 2**24 - 2
    mis = seed % 4  # start the device array 0..3 ints off a 16-byte boundary
    dev = torch.zeros(total + 8, dtype=torch.int32, device="cuda:0")
    view = dev[mis : mis + total]
    view.copy_(torch.from_numpy(counts))
    r = eng.score(view, torch.from_numpy(offsets).cuda(), algo=algo)
    torch.cuda.synchronize()
    res = r.cpu_numpy()
    assert_matches_oracle(res, counts, offsets)
    split = (res["flags"] & 4) != 0
    from ribotricer_amd import _lib

    tile = _lib.tile_positions(lens.size, total)  # 7 936, or 6 144 for an index of short ORFs
    starts, ends = offsets[:-1] + mis, offsets[1:] + mis
    expect_split = (lens > 0) & ((ends - 1) // tile > starts // tile)
    assert np.array_equal(split, expect_split)


@pytest.mark.parametrize("seed", range(6))
def test_small_tile_boundaries(eng, seed):
    """Indexes of short ORFs run on 6 144-position tiles: ORF ends hugging those boundaries, empty
    ORFs on them, a few multi-tile ORFs among thousands of short ones, every 16-byte phase."""
    from ribotricer_amd import _lib

    rng = np.random.default_rng(500 + seed)
    small = 6144
    lens = []
    total = 0
    for k in range(1, 9):  # short ORFs up to each boundary, the last one ending at boundary + d
        d = int(rng.integers(-2, 3))
        target = small * k + d - (seed % 4)  # (the device array starts `mis` ints off the 16-byte grid)
        while target - total > 130:
            n = int(rng.integers(1, 100))
            lens.append(n)
            total += n
        lens.append(target - total)
        total = target
        if k % 3 == 0:
            lens += [0, 0]  # empty ORFs sitting on a boundary
    lens += [int(x) for x in rng.integers(1, 60, 3000)]
    lens[len(lens) // 2] = 3 * small + 7          # a multi-tile ORF among the short ones
    lens.append(2 * small)
    lens = np.asarray(lens, np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offsets[-1])
    assert _lib.tile_positions(lens.size, total) == small
    counts = rng.poisson(np.repeat(rng.choice([0.02, 0.4, 3.0], size=lens.size), lens)).astype(np.int32)
    mis = seed % 4
    dev = torch.zeros(total + 8, dtype=torch.int32, device="cuda:0")
    view = dev[mis : mis + total]
    view.copy_(torch.from_numpy(counts))
    res = eng.score(view, torch.from_numpy(offsets).cuda(), algo="tile").cpu_numpy()
    assert_matches_oracle(res, counts, offsets)
    starts, ends = offsets[:-1] + mis, offsets[1:] + mis
    assert np.array_equal((res["flags"] & 4) != 0, (lens > 0) & ((ends - 1) // small > starts // small))


@pytest.mark.parametrize("seed", range(4))
def test_lane_per_segment_rounds(eng, seed):
    """Tiles of nothing but very short segments (more than 64 of them, none over 22 triplets) take the lane-per-segment
    rounds of round 4 (one lane walks a whole segment straight to its record); one segment of 23 triplets (67-69 nt) in
    a tile sends that tile back to the 64-slot rounds.  Lengths around both limits, empty ORFs, L % 3 != 0 (partial last
    codons), ORFs cut by tile boundaries, > 256 segments per tile (1-5-nt ORFs: several lane rounds), sparse and dense
    counts (exact frame ties, flat codons); the CSR scorer and the fused scorer against the oracle."""
    from ribotricer_amd.engine import make_filter
    from ribotricer_amd.gather import GatherPlan, IntervalTable

    rng = np.random.default_rng(900 + seed)
    parts = [rng.choice([60, 60, 60, 63, 66, 61, 62, 64, 65], size=4000),          # lane rounds, partial codons
             rng.choice([60, 66, 67, 68, 69, 70], size=3000, p=[0.5, 0.3, 0.05, 0.05, 0.05, 0.05]),  # tiles that fall back
             rng.integers(0, 6, size=6000),                                          # > 256 segments per tile, empty ORFs
             rng.choice([60, 63], size=3000)]
    lens = np.concatenate(parts).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    lam = np.repeat(rng.choice([0.01, 0.05, 0.5, 4.0], size=lens.size), lens)
    counts = (rng.poisson(lam) * (rng.random(lam.size) < 0.9)).astype(np.int32)
    counts[rng.random(counts.size) < 0.02] = 7  # (runs of flat codons somewhere)
    res = eng.score_host(counts, offsets, thresholds=make_filter(), algo="tile")
    assert_matches_oracle(res, counts, offsets)
    n = lens.size
    keep = lens > 0  # (an interval may not be empty: the fused path takes the ORFs that have a profile)
    sub_off = np.concatenate([[0], np.cumsum(lens[keep])]).astype(np.int64)
    table = IntervalTable(offsets[:-1][keep].copy(), lens[keep].astype(np.int32), np.arange(int(keep.sum()) + 1, dtype=np.int64),
                          (np.arange(int(keep.sum())) % 2).astype(np.uint8) * 0, sub_off)
    fused = eng.score_coverage(counts, GatherPlan(table, counts.size), thresholds=make_filter()).cpu_numpy()
    for k in ("phase", "valid", "read_count", "min_codon_cov", "status"):
        assert np.array_equal(fused[k], res[k][keep]) or k == "phase", k
    assert_matches_oracle(fused, counts, sub_off)
