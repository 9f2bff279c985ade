"""The gather plan (rp_pieces.hpp): the tile gather against the per-ORF gather and a numpy
restatement of orf_coverage (detect_orfs.py:134-203), and the fused gather + score against
gather-then-score -- which must agree bit for bit, ties and float64 re-walks included."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_table(rng, n, cov_len, max_exons=6, exon_len=(1, 400), special=()):
    """IntervalTable with every interval inside [0, cov_len)."""
    from ribotricer_amd.gather import IntervalTable

    n_iv = rng.integers(1, max_exons + 1, size=n)
    for i, k in special:
        n_iv[i] = k
    orf_iv = np.concatenate([[0], np.cumsum(n_iv)]).astype(np.int64)
    iv_len = rng.integers(exon_len[0], exon_len[1] + 1, size=int(orf_iv[-1])).astype(np.int32)
    iv_start = rng.integers(0, cov_len - exon_len[1] - 1, size=int(orf_iv[-1])).astype(np.int64)
    reverse = rng.integers(0, 2, size=n).astype(np.uint8)
    lengths = np.add.reduceat(iv_len.astype(np.int64), orf_iv[:-1])
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    return IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets)


def numpy_gather(cov, t):
    out = np.empty(int(t.offsets[-1]), np.int32)
    for i in range(len(t.offsets) - 1):
        parts = [cov[t.iv_start[k] : t.iv_start[k] + t.iv_len[k]] for k in range(t.orf_iv[i], t.orf_iv[i + 1])]
        prof = np.concatenate(parts) if parts else np.zeros(0, np.int32)
        out[t.offsets[i] : t.offsets[i + 1]] = prof[::-1] if t.reverse[i] else prof
    return out


CASES = {
    "exons": dict(n=4000, max_exons=6, exon_len=(1, 400), special=((5, 150), (6, 64), (7, 65), (3999, 70))),
    "tiny_exons": dict(n=3000, max_exons=40, exon_len=(1, 3)),       # far more chunks than a row holds: the slow path
    "small_exons": dict(n=6000, max_exons=12, exon_len=(8, 40)),      # rows around the 256-slot limit of the fast path: both staging paths
    "single_long": dict(n=300, max_exons=1, exon_len=(5000, 30000)),  # pieces spanning several tiles
    "short_orfs": dict(n=20000, max_exons=1, exon_len=(60, 150)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_tile_gather(case):
    import torch

    from ribotricer_amd.gather import GatherPlan, gather_profiles_device

    rng = np.random.default_rng(11)
    cov = rng.poisson(0.7, size=400000).astype(np.int32)
    t = random_table(rng, cov_len=cov.size, **CASES[case])
    plan = GatherPlan(t, cov.size)
    assert plan.stats()["wide_tiles"] == 0  # (a 1.6 MB coverage: every tile takes the 32-bit issue loop)
    got, off = gather_profiles_device(cov, t, plan=plan)
    legacy, _ = gather_profiles_device(cov, t)
    torch.cuda.synchronize()
    want = numpy_gather(cov, t)
    assert np.array_equal(off.cpu().numpy(), t.offsets)
    assert np.array_equal(legacy.cpu().numpy(), want)
    assert np.array_equal(got.cpu().numpy(), want)


def test_gather_selected_equals_the_sub_table_gather():
    """GatherPlan.gather_selected (one wave per chosen ORF through the plan's pieces; ids and prefix sums on the device)
    against the per-ORF gather over select_orfs' sub-table and against numpy: none / some / all ORFs chosen, multi-exon
    and '-' strand ORFs, every other ORF with an empty range in the returned offsets."""
    import torch

    from ribotricer_amd.gather import GatherPlan, gather_profiles_device, select_orfs

    rng = np.random.default_rng(77)
    cov = rng.poisson(0.7, size=400000).astype(np.int32)
    t = random_table(rng, cov_len=cov.size, **CASES["exons"])
    plan = GatherPlan(t, cov.size)
    whole = numpy_gather(cov, t)
    n = len(t.offsets) - 1
    for frac in (0.0, 0.2, 1.0):
        keep = rng.random(n) < frac
        counts, offsets = plan.gather_selected(cov, keep)
        torch.cuda.synchronize()
        lengths = np.diff(t.offsets)
        assert np.array_equal(np.diff(offsets), np.where(keep, lengths, 0))
        want = np.concatenate([whole[t.offsets[i] : t.offsets[i + 1]] for i in np.flatnonzero(keep)]) if keep.any() else np.zeros(0, np.int32)
        assert np.array_equal(counts.cpu().numpy(), want)
        sub, _ = gather_profiles_device(cov, select_orfs(t, np.flatnonzero(keep)))
        assert torch.equal(sub, counts)


def test_reused_host_arrays_of_the_export_path():
    """What export_orf_coverages reuses from sample to sample: PhaseScores.cpu_numpy(pinned) returns views of pinned
    staging tensors kept in the caller's dict (equal to the plain copies; the next call overwrites them), and
    GatherPlan.gather_selected(reuse_arrays=True) returns the plan's scratch offsets (a copy by default)."""
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan

    rng = np.random.default_rng(21)
    cov = rng.poisson(0.8, size=600_000).astype(np.int32)
    t = random_table(rng, 3000, cov.size, max_exons=4, exon_len=(30, 300))
    plan = GatherPlan(t, cov.size)
    eng = get_engine("cuda:0")
    res = eng.score_coverage(cov, plan, thresholds=make_filter())
    plain = res.cpu_numpy()
    keep_dict: dict = {}
    pinned = res.cpu_numpy(keep_dict)
    assert all(np.array_equal(plain[k], pinned[k]) for k in plain)
    assert all(keep_dict[k].is_pinned() for k in keep_dict)
    first_phase = pinned["phase"]
    res2 = eng.score_coverage(np.zeros_like(cov), plan, thresholds=make_filter())
    again = res2.cpu_numpy(keep_dict)
    assert again["phase"] is not first_phase and np.shares_memory(again["phase"], first_phase)  # (same staging memory)
    assert not again["phase"].any() and not first_phase.any()  # ... which the second call overwrote
    keep = plain["status"] != 0
    lengths = np.diff(t.offsets)
    c1, o1 = plan.gather_selected(cov, keep, lengths=lengths)
    c2, o2 = plan.gather_selected(cov, keep, lengths=lengths, reuse_arrays=True)
    assert torch.equal(c1, c2) and np.array_equal(o1, o2)
    _, o3 = plan.gather_selected(cov, np.zeros_like(keep), lengths=lengths, reuse_arrays=True)
    assert np.shares_memory(o2, o3) and not o3.any() and o1.any()  # (the default's copy is untouched)


@pytest.mark.parametrize("block", [1, 8, 64])
def test_compact_coverage_map(block):
    """gather.CoverageMap: only the blocks of `block` positions under an exon interval keep a slot.  The table in compact
    coordinates reads out of the compact coverage exactly what the dense table reads out of the dense one (per-ORF gather,
    tile gather, fused scores bit for bit), the compact length is the number of kept blocks x block, and positions under
    no exon have no slot.  With one-position blocks the plan stages merged runs: fewer chunks than pieces need alone."""
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import CoverageMap, GatherPlan, gather_profiles_device

    rng = np.random.default_rng(123)
    dense_len = 3_000_000
    cov = rng.poisson(0.5, size=dense_len).astype(np.int32)
    t = random_table(rng, 5000, dense_len, max_exons=5, exon_len=(1, 500))
    cmap = CoverageMap(t, dense_len, block_positions=block)
    assert cmap.block_positions == block
    blocks = np.zeros((dense_len + block - 1) // block, bool)
    for s0, n in zip(t.iv_start, t.iv_len):
        blocks[s0 // block : (s0 + n - 1) // block + 1] = True
    assert cmap.compact_len == block * int(blocks.sum()) < dense_len
    slot_of_block = np.cumsum(blocks) - 1
    want_start = slot_of_block[t.iv_start // block] * block + t.iv_start % block
    assert np.array_equal(cmap.table.iv_start, want_start)
    # a second map whose device copy of the compact starts is dropped before anybody asks for the host copy (what the
    # export does once its gather plan is built): the host copy is re-made through the map (rp_coverage_map_remap_dev)
    late = CoverageMap(t, dense_len, block_positions=block)
    late.release_device_intervals()
    assert late.table._d_iv_start is None and late.device_intervals is None
    assert np.array_equal(late.table.iv_start, want_start) and np.array_equal(late.remap(t.iv_start[:7]), want_start[:7])
    del late
    padded = np.zeros(blocks.size * block, np.int32)
    padded[:dense_len] = cov
    compact = padded.reshape(-1, block)[blocks].reshape(-1)  # the compact coverage as the build kernel would fill it
    assert compact.size == cmap.compact_len
    dense_counts, _ = gather_profiles_device(cov, t)
    compact_counts, _ = gather_profiles_device(compact, cmap.table)
    assert torch.equal(dense_counts, compact_counts)
    eng = get_engine("cuda:0")
    dense_plan, compact_plan = GatherPlan(t, dense_len), GatherPlan(cmap.table, cmap.compact_len)
    a = eng.score_coverage(cov, dense_plan, thresholds=make_filter()).cpu_numpy()
    b = eng.score_coverage(compact, compact_plan, thresholds=make_filter()).cpu_numpy()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert compact_plan.stats()["slow_tiles"] == 0


def test_contiguous_pieces_are_staged_as_runs():
    """Exons that face each other across an intron are neighbours in a compact coverage with one-position blocks: the plan
    stages them as one run (chunks of 64 positions that span the exon boundaries), so a layout without reuse needs about
    tile / 64 chunks per tile whatever its exon structure -- and reads the same counts."""
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import CoverageMap, GatherPlan, IntervalTable, gather_profiles_device
    from ribotricer_amd.synth import orf_lengths, synth_exon_layout

    lengths = orf_lengths(40_000, 5, "cfg3")
    iv_start, iv_len, orf_iv, reverse, offsets, dense_len = synth_exon_layout(lengths, 5)
    t = IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets)
    rng = np.random.default_rng(9)
    cov = rng.poisson(0.7, size=dense_len).astype(np.int32)
    stats = {}
    results = {}
    for block in (64, 1):
        cmap = CoverageMap(t, dense_len, block_positions=block)
        padded = np.zeros(-(-dense_len // block) * block, np.int32)
        padded[:dense_len] = cov
        blocks = np.zeros(padded.size // block, bool)
        for s0, n in zip(t.iv_start, t.iv_len):
            blocks[s0 // block : (s0 + n - 1) // block + 1] = True
        compact = padded.reshape(-1, block)[blocks].reshape(-1)
        plan = GatherPlan(cmap.table, cmap.compact_len)
        stats[block] = plan.stats()
        results[block] = get_engine("cuda:0").score_coverage(compact, plan, thresholds=make_filter()).cpu_numpy()
        counts, _ = gather_profiles_device(compact, cmap.table)
        want, _ = gather_profiles_device(cov, t)
        assert torch.equal(counts, want)
    assert all(np.array_equal(results[1][k], results[64][k]) for k in results[1])
    assert stats[1]["slow_tiles"] == stats[64]["slow_tiles"] == 0
    assert stats[1]["chunks_per_tile_mean"] < 0.95 * stats[64]["chunks_per_tile_mean"]


def abutting_table(rng, n, cov_len):
    """ORFs laid along the coverage with exons that often ABUT (gap 0: pieces that continue their predecessor, staged
    as one run), lengths around the chunk size and starts around the 256-byte lines of the source -- the places where
    the run merge and the source-aligned chunk cuts of k_chunk_rows decide something."""
    from ribotricer_amd.gather import IntervalTable

    lens_pool = np.array([1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 200, 640, 1000, 9000], np.int64)
    n_iv = rng.integers(1, 7, size=n)
    orf_iv = np.concatenate([[0], np.cumsum(n_iv)]).astype(np.int64)
    m = int(orf_iv[-1])
    iv_len = rng.choice(lens_pool, size=m)
    gap = np.where(rng.random(m) < 0.55, 0, rng.integers(1, 200, size=m))
    snap = rng.random(m) < 0.3  # some exons start exactly on / next to a 64-element boundary
    iv_start = np.zeros(m, np.int64)
    at = 70
    for k in range(m):
        s0 = at + int(gap[k])
        if snap[k]:
            s0 = (s0 + 63) // 64 * 64 + int(rng.integers(-1, 2))
            s0 = max(s0, at)
        iv_start[k] = s0
        at = s0 + int(iv_len[k])
    assert at < cov_len
    reverse = (rng.random(n) < 0.5).astype(np.uint8)
    reverse[: n // 4] = 0  # a stretch of forward ORFs in a row: runs that span ORFs
    reverse[n // 4 : n // 2] = 1
    lengths = np.add.reduceat(iv_len, orf_iv[:-1])
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    return IntervalTable(iv_start, iv_len.astype(np.int32), orf_iv, reverse, offsets)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_abutting_exons_and_line_boundaries(seed):
    """Run merge + source-aligned cuts: the tile gather equals numpy's orf_coverage, the fused scores equal
    gather-then-score bit for bit, and no tile falls back to the slow path (that merging happens is
    test_contiguous_pieces_are_staged_as_runs' business)."""
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan, gather_profiles_device

    rng = np.random.default_rng(seed)
    n = 3000
    cov_len = 12_000_000
    cov = rng.poisson(0.6, size=cov_len).astype(np.int32)
    t = abutting_table(rng, n, cov_len)
    plan = GatherPlan(t, cov_len)
    st = plan.stats()
    assert st["slow_tiles"] == 0
    got, _ = gather_profiles_device(cov, t, plan=plan)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), numpy_gather(cov, t))
    eng = get_engine("cuda:0")
    fused = eng.score_coverage(cov, plan, thresholds=make_filter()).cpu_numpy()
    two_step = eng.score(got, torch.from_numpy(t.offsets).cuda(), thresholds=make_filter(), algo="tile").cpu_numpy()
    assert all(np.array_equal(fused[k], two_step[k]) for k in fused)


def test_unplannable_tables():
    from ribotricer_amd import _lib
    from ribotricer_amd.gather import GatherPlan, IntervalTable, make_gather_plan

    rng = np.random.default_rng(3)
    t = random_table(rng, 50, 10000)
    off_array = t._replace(iv_start=np.where(np.arange(t.iv_start.size) == 7, 10000 - 2, t.iv_start))
    assert make_gather_plan(off_array, 10000) is None  # the per-ORF kernel reads those positions as 0
    with pytest.raises(_lib.RibophaseError) as e:
        GatherPlan(off_array, 10000)
    assert e.value.status == _lib.ERR_INTERVALS
    short = t._replace(offsets=t.offsets + np.arange(t.offsets.size))  # lengths no longer add up
    with pytest.raises(_lib.RibophaseError) as e:
        GatherPlan(IntervalTable(*short), 10000)
    assert e.value.status == -3
    empty = GatherPlan(IntervalTable(np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.uint8), np.zeros(1, np.int64)), 100)
    assert empty.gather(np.zeros(100, np.int32)).numel() == 0


@pytest.mark.parametrize("lam", [0.01, 0.3, 3.0])
@pytest.mark.parametrize("case", list(CASES))
def test_fused_score_equals_gather_then_score(case, lam):
    import torch

    from helpers import assert_matches_oracle
    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan

    rng = np.random.default_rng(int(lam * 100) + 17)
    cov = rng.poisson(lam, size=400000).astype(np.int32)
    t = random_table(rng, cov_len=cov.size, **CASES[case])
    eng = get_engine("cuda:0")
    plan = GatherPlan(t, cov.size)
    th = make_filter()
    d_cov = torch.from_numpy(cov).cuda()
    fused = eng.score_coverage(d_cov, plan, thresholds=th).cpu_numpy()
    counts = plan.gather(d_cov)
    plain = eng.score(counts, plan.offsets, thresholds=th, algo="tile").cpu_numpy()
    for k in ("phase", "valid", "read_count", "min_codon_cov", "flags", "status"):
        assert np.array_equal(fused[k], plain[k]), k  # phase: the same float64 bits
    assert_matches_oracle(fused, counts.cpu().numpy(), t.offsets)
    if lam < 0.1 and case != "single_long":  # (long ORFs hold too many reads to tie)
        assert (fused["flags"] & 0x08).any()  # ties were replayed through the piece view


def test_pieces_of_one_tile_gigabytes_apart():
    """Consecutive ORFs of an index on different chromosomes: the pieces of ONE tile lie gigabytes apart in the
    dense coverage (12.9 GB here, three islands).  Until round 4 a chunk's byte offset had 32 bits and such tiles
    took the scalar slow path (93 % of the tiles of the 11 M-ORF nested index: profiles/archive/r04_fused_nested_before.json);
    now the offset carries 7 more bits.  Tile gather == per-ORF gather (a code path of its own), fused == gather +
    score bit for bit, and the oracle agrees."""
    import torch

    from helpers import assert_matches_oracle
    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan, IntervalTable, gather_profiles_device

    rng = np.random.default_rng(41)
    cov_len = 3 * (1 << 30) + 12345
    island = 300000
    bases = [0, (1 << 30) + 77, (1 << 31) + 5, cov_len - island]  # byte offsets 0, 4 GiB + .., 8 GiB + .., 12 GiB + ..
    cov = torch.zeros(cov_len, dtype=torch.int32, device="cuda")
    for b in bases:
        cov[b : b + island] = torch.from_numpy(rng.poisson(0.6, size=island).astype(np.int32)).cuda()
    t = random_table(rng, 6000, island, max_exons=4, exon_len=(20, 300))
    # every interval moves to the island of its ORF; ORF i lives on island i % 4, so every tile mixes all four
    orf_of_iv = np.repeat(np.arange(6000), np.diff(t.orf_iv))
    shift = np.asarray(bases, np.int64)[orf_of_iv % 4]
    t = IntervalTable(t.iv_start + shift, t.iv_len, t.orf_iv, t.reverse, t.offsets)
    plan = GatherPlan(t, cov_len)
    st = plan.stats()
    # the point: the tiles do not fall back to the scalar loop any more -- none of them, although the islands sit right
    # behind multiples of 4 GiB and some chunks' low offset halves end within 256 bytes of 2^32 (the chunk's offset is
    # added to the tile's base in 64-bit scalar arithmetic, the lanes' own 252 bytes in the address unit: no carry is
    # lost; while the lanes added them in 32-bit vector arithmetic such tiles were planned slow)
    assert st["slow_tiles"] == 0 and st["tiles"] >= 100, st
    assert st["wide_tiles"] > 0.8 * st["tiles"], st  # (offsets past 32 bits: the longer issue loop)
    eng = get_engine("cuda:0")
    th = make_filter()
    counts = plan.gather(cov)
    legacy, _ = gather_profiles_device(cov, t)
    assert torch.equal(counts, legacy)
    fused = eng.score_coverage(cov, plan, thresholds=th).cpu_numpy()
    plain = eng.score(counts, plan.offsets, thresholds=th, algo="tile").cpu_numpy()
    for k in ("phase", "valid", "read_count", "min_codon_cov", "flags", "status"):
        assert np.array_equal(fused[k], plain[k]), k
    assert_matches_oracle(fused, counts.cpu().numpy(), t.offsets)
    del cov
    torch.cuda.empty_cache()


def test_fused_long_rewalk_reads_through_the_plan():
    """A cutoff placed on a long ORF's own phase score sends it to k_rewalk_long (float64,
    a workgroup per ORF), which in fused mode reads the coverage through the piece view."""
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan

    rng = np.random.default_rng(23)
    cov = torch.from_numpy(rng.poisson(0.3, size=400000).astype(np.int32)).cuda()
    t = random_table(rng, 200, 400000, max_exons=3, exon_len=(3000, 9000))
    eng = get_engine("cuda:0")
    plan = GatherPlan(t, 400000)
    first = eng.score_coverage(cov, plan, thresholds=make_filter()).cpu_numpy()
    pick = [7, 100, 199]
    for i in pick:
        th = make_filter(phase_score_cutoff=float(first["phase"][i]))
        fused = eng.score_coverage(cov, plan, thresholds=th).cpu_numpy()
        plain = eng.score(plan.gather(cov), plan.offsets, thresholds=th, algo="tile").cpu_numpy()
        assert all(np.array_equal(fused[k], plain[k]) for k in fused)
        assert fused["flags"][i] & 0x02, "the ORF on the cutoff was not re-walked in float64"


def test_fused_score_repeats_and_reuses_the_plan():
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan

    rng = np.random.default_rng(5)
    t = random_table(rng, 5000, 300000)
    plan = GatherPlan(t, 300000)
    eng = get_engine("cuda:0")
    for seed in (1, 2):  # two "samples" against one index
        cov = torch.from_numpy(np.random.default_rng(seed).poisson(0.5, 300000).astype(np.int32)).cuda()
        a = eng.score_coverage(cov, plan, thresholds=make_filter()).cpu_numpy()
        b = eng.score(plan.gather(cov), plan.offsets, thresholds=make_filter(), algo="tile").cpu_numpy()
        assert all(np.array_equal(a[k], b[k]) for k in a)


def test_fused_sharded_equals_single():
    """engine.score_coverage_sharded: slices of the interval table, a gather plan per slice."""
    from ribotricer_amd.engine import get_engine, make_filter, score_coverage_sharded
    from ribotricer_amd.gather import GatherPlan

    rng = np.random.default_rng(9)
    cov = rng.poisson(0.4, size=300000).astype(np.int32)
    t = random_table(rng, 4000, cov.size)
    whole = get_engine("cuda:0").score_coverage(cov, GatherPlan(t, cov.size), thresholds=make_filter()).cpu_numpy()
    for devices in (["cuda:0"], [0, 0], [0, 0, 0, 0, 0]):
        parts = score_coverage_sharded(cov, t, devices, thresholds=make_filter())
        for k in ("valid", "read_count", "min_codon_cov", "status"):
            assert np.array_equal(parts[k], whole[k]), (k, devices)
        assert np.abs(parts["phase"] - whole["phase"]).max() <= 1e-6  # another tiling moves fp32 sums by <= 2e-7


def test_coverage_shards_keep_their_plans_across_samples():
    """engine.CoverageShards: windows, gather plans and tile plans of the slices are built by the FIRST sample only;
    later samples (other coverage, same layout) upload their windows and launch -- and equal the one-GPU result."""
    from ribotricer_amd.engine import CoverageShards, get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan

    rng = np.random.default_rng(19)
    n_cov = 300000
    t = random_table(rng, 4000, n_cov)
    whole_plan = GatherPlan(t, n_cov)
    shards = CoverageShards(t, [0, 0, 0], n_cov)
    for sample in range(3):
        cov = rng.poisson(0.2 + 0.3 * sample, size=n_cov).astype(np.int32)
        whole = get_engine("cuda:0").score_coverage(cov, whole_plan, thresholds=make_filter()).cpu_numpy()
        parts = shards.score(cov, thresholds=make_filter())
        for k in ("valid", "read_count", "min_codon_cov", "status"):
            assert np.array_equal(parts[k], whole[k]), (k, sample)
        assert np.abs(parts["phase"] - whole["phase"]).max() <= 1e-6
        assert shards.plans_built == 3, (sample, shards.plans_built)
    with pytest.raises(ValueError):
        shards.score(np.zeros(n_cov + 16, np.int32))
    shards.release()


def test_gather_plan_abi_errors():
    """Argument checks of the gather-plan entry points: statuses, not crashes."""
    import ctypes

    import torch

    from ribotricer_amd import _lib
    from ribotricer_amd.engine import _ptr, get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan

    rng = np.random.default_rng(2)
    t = random_table(rng, 300, 50000)
    plan = GatherPlan(t, 50000)
    lib = _lib.load()
    cov = torch.zeros(50000, dtype=torch.int32, device="cuda")
    out = torch.empty(plan.total_nt + 4, dtype=torch.int32, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rp_gather_profiles_plan_dev(None, _ptr(cov), 50000, _ptr(out), stream) == -1
    assert lib.rp_gather_profiles_plan_dev(plan.handle, _ptr(cov), 49999, _ptr(out), stream) == -8  # another coverage layout
    assert lib.rp_gather_profiles_plan_dev(plan.handle, None, 50000, _ptr(out), stream) == -1
    assert lib.rp_gather_profiles_plan_dev(plan.handle, _ptr(cov), 50000, ctypes.c_void_p(out.data_ptr() + 4), stream) == -8  # 16-byte alignment
    assert b"aligned" in lib.rp_last_error()
    need = ctypes.c_size_t(0)
    assert lib.rp_gather_plan_bytes(-1, 0, 0, ctypes.byref(need)) == -2
    assert lib.rp_gather_plan_bytes(300, len(t.iv_start), plan.total_nt, ctypes.byref(need)) == 0 and need.value > 0
    handle = ctypes.c_void_p(0)
    small = torch.empty(64, dtype=torch.uint8, device="cuda")
    args = [_ptr(torch.from_numpy(a).cuda()) for a in (t.iv_start, t.iv_len, t.orf_iv, t.reverse, t.offsets)]
    rc = lib.rp_gather_plan_create_dev(0, *args, 300, len(t.iv_start), plan.total_nt, 50000, _ptr(small), 64, stream, ctypes.byref(handle))
    assert rc == -5 and not handle.value  # plan memory too small
    # a plan of another index handed to the fused scorer
    other = GatherPlan(random_table(rng, 200, 50000), 50000)
    eng = get_engine("cuda:0")
    with pytest.raises(_lib.RibophaseError) as e:
        bad = GatherPlan(t, 50000)
        bad.offsets = other.offsets  # mismatched offsets / plan
        bad.n_orfs, bad.total_nt = other.n_orfs, other.total_nt
        eng.score_coverage(cov, bad, thresholds=make_filter())
    assert e.value.status == -8
    with pytest.raises(_lib.RibophaseError):
        eng.score_coverage(torch.zeros(10, dtype=torch.int32, device="cuda"), plan)  # coverage of another length


def test_fused_with_empty_orfs():
    """ORFs without any interval (empty profiles) among ordinary ones, and an index of nothing else."""
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan, IntervalTable

    rng = np.random.default_rng(31)
    cov = rng.poisson(0.5, size=100000).astype(np.int32)
    t = random_table(rng, 2000, cov.size)
    # every third ORF loses its intervals
    keep = np.arange(2000) % 3 != 0
    nk = np.diff(t.orf_iv) * keep
    orf_iv = np.concatenate([[0], np.cumsum(nk)]).astype(np.int64)
    pick = np.concatenate([np.arange(t.orf_iv[i], t.orf_iv[i + 1]) for i in range(2000) if keep[i]])
    lengths = np.diff(t.offsets) * keep
    mixed = IntervalTable(t.iv_start[pick], t.iv_len[pick], orf_iv, t.reverse, np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64))
    none = IntervalTable(np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(51, np.int64), np.zeros(50, np.uint8), np.zeros(51, np.int64))
    eng = get_engine("cuda:0")
    d_cov = torch.from_numpy(cov).cuda()
    for table in (mixed, none):
        plan = GatherPlan(table, cov.size)
        fused = eng.score_coverage(d_cov, plan, thresholds=make_filter()).cpu_numpy()
        plain = eng.score(plan.gather(d_cov), plan.offsets, thresholds=make_filter(), algo="tile").cpu_numpy()
        assert all(np.array_equal(fused[k], plain[k]) for k in fused)
        empty = np.diff(table.offsets) == 0
        assert (fused["phase"][empty] == 0).all() and (fused["valid"][empty] == 0).all() and (fused["status"][empty] == 0).all()
