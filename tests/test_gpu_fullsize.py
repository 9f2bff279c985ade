"""Parity at the headline sizes (BASELINE configs[2] and configs[4]): the reference's loop has no size
limit (detect_orfs.py:274-324), so neither may the kernels -- in particular no 32-bit index may
survive where offsets pass 2^31 and 2^32 nucleotides.

One synthetic candidate index per config (ribotricer_amd/synth.py: exon layout over ONE dense
coverage array, '-' strand ORFs read backwards), scored at full size by
  * the fused gather + score (rp_phase_score_coverage_dev, 32-bit chunk offsets inside),
  * the tile gather (rp_gather_profiles_plan_dev) + the CSR scorer with a plan
    (rp_phase_score_csr_plan_dev),
  * the wave-per-ORF kernel (configs[2] only),
then head / middle / TAIL slices of >= 20 000 ORFs each -- the tail entirely past 2^31 resp. 2^32
nucleotides -- and the ORFs straddling those boundaries are pulled to the host, their profiles
rebuilt from the coverage by plain numpy (no kernel involved), and compared with the C oracle.
"""

import numpy as np
import pytest

from helpers import assert_matches_oracle

pytestmark = pytest.mark.gpu

SLICE = 20_000
# (config, ORFs, boundary that the tail slice must lie behind)
SETS = {
    "cfg3": ("cfg3", 11_000_000, 1 << 31),   # BASELINE configs[2]: 3.97 G nt
    "cfg5": ("cfg5", 13_500_000, 1 << 32),   # BASELINE configs[4] law, cut where it passes 2^32 nt (4.6 G nt)
}


def _build(cfg, n_orfs, seed):
    import torch

    from ribotricer_amd.gather import GatherPlan, IntervalTable
    from ribotricer_amd.synth import orf_lengths, synth_coverage_device, synth_exon_layout

    lengths = orf_lengths(n_orfs, seed, cfg)
    iv_start, iv_len, orf_iv, reverse, offsets, coverage_len = synth_exon_layout(lengths, seed)
    table = IntervalTable(iv_start, iv_len, orf_iv, reverse, offsets)
    cov = synth_coverage_device(coverage_len, seed, device="cuda:0")
    plan = GatherPlan(table, coverage_len, "cuda:0")
    torch.cuda.synchronize()
    return table, cov, plan


def _slices(offsets, behind):
    """ORF ranges to check: head, middle, tail, and the ORFs around every 2^31-multiple boundary."""
    n = offsets.size - 1
    out = {"head": (0, SLICE), "middle": (n // 2 - SLICE // 2, n // 2 + SLICE // 2), "tail": (n - SLICE, n)}
    for name, b in (("at_2^31", 1 << 31), ("at_2^32", 1 << 32)):
        if int(offsets[-1]) > b:
            i = int(np.searchsorted(offsets, b))
            out[name] = (max(0, i - SLICE // 4), min(n, i + SLICE // 4))
    assert int(offsets[out["tail"][0]]) > behind, "the tail slice must lie entirely past the boundary"
    return out


def _host_profiles(table, cov, lo, hi):
    """Profiles of ORFs lo..hi rebuilt on the host from the coverage bytes (numpy only)."""
    from ribotricer_amd.synth import profiles_from_coverage

    k0, k1 = int(table.orf_iv[lo]), int(table.orf_iv[hi])
    w_lo = int(table.iv_start[k0:k1].min())
    w_hi = int((table.iv_start[k0:k1] + table.iv_len[k0:k1]).max())
    window = cov[w_lo:w_hi].cpu().numpy()
    return profiles_from_coverage(window, w_lo, table.iv_start, table.iv_len, table.orf_iv, table.reverse, lo, hi)


def _cut(res, lo, hi):
    return {k: v[lo:hi].cpu().numpy() for k, v in res._asdict().items() if v is not None}


@pytest.mark.parametrize("name", ["cfg3", "cfg5"])
def test_full_size_against_the_oracle(name):
    import torch

    from ribotricer_amd.engine import get_engine, make_filter

    cfg, n_orfs, behind = SETS[name]
    eng = get_engine("cuda:0")
    table, cov, plan = _build(cfg, n_orfs, seed=20260213)
    offsets = table.offsets
    total_nt = int(offsets[-1])
    assert total_nt > behind + (1 << 24)
    th = make_filter()

    fused = eng.score_coverage(cov, plan, thresholds=th)
    counts = plan.gather(cov)  # the CSR profiles of the whole index (tile gather)
    csr = eng.score(counts, plan.offsets, thresholds=th, algo="tile")  # plan="auto": rp_phase_score_csr_plan_dev
    torch.cuda.synchronize()
    assert counts.numel() == total_nt

    # the fused path and gather-then-score agree bit for bit on EVERY ORF
    for k in ("phase", "valid", "read_count", "min_codon_cov", "flags", "status"):
        assert torch.equal(getattr(fused, k), getattr(csr, k)), k
    # size-independent properties over the whole set
    assert int(csr.read_count.sum()) == int(counts.sum(dtype=torch.int64))
    lengths = plan.offsets[1:] - plan.offsets[:-1]
    assert bool((csr.valid <= lengths // 3).all()) and bool((csr.valid >= 0).all())
    assert bool(((csr.phase >= 0) & (csr.phase <= 1.0 + 1e-6)).all())
    n_translating = int(csr.status.sum())
    assert 0 < n_translating < n_orfs

    if name == "cfg3":  # the wave-per-ORF kernel on the same bytes: integers equal, phase within tolerance
        wave = eng.score(counts, plan.offsets, thresholds=th, algo="wave", plan=None)
        torch.cuda.synchronize()
        for k in ("valid", "read_count", "min_codon_cov"):  # (exact ties are replayed by both kernels: equal too)
            assert torch.equal(getattr(wave, k), getattr(csr, k)), k
        assert float((wave.phase - csr.phase).abs().max()) <= 1e-6
        assert int(wave.status.sum()) == n_translating
        del wave

    checked, worst = 0, 0.0
    for label, (lo, hi) in _slices(offsets, behind).items():
        c_host, o_host = _host_profiles(table, cov, lo, hi)
        a, b = int(offsets[lo]), int(offsets[hi])
        # the tile gather wrote exactly the profiles the host rebuilds from the coverage
        assert np.array_equal(counts[a:b].cpu().numpy(), c_host), f"{label}: gathered profiles differ from the host gather"
        assert np.array_equal(o_host, offsets[lo : hi + 1] - offsets[lo])
        for path, res in (("fused", fused), ("csr", csr)):
            part = _cut(res, lo, hi)
            o = assert_matches_oracle(part, c_host, o_host)
            worst = max(worst, float(np.abs(part["phase"] - o.phase).max(initial=0.0)))
            expect = ((part["phase"] >= 0.428571428571) & (part["valid"] >= 5)).astype(np.uint8)
            assert np.array_equal(part["status"], expect), f"{label}/{path}: status"
        checked += hi - lo
    assert checked >= 3 * SLICE
    print(f"{name}: {n_orfs} ORFs, {total_nt} nt, {checked} ORFs checked against the oracle on both paths, max |dphase| {worst:.2e}")


def test_configs4_at_its_stated_size():
    """BASELINE configs[4] AS STATED: 50 M ORFs, Pareto tail clipped at 100 k nt, 17.2 G nt -- 69 GB of counts, offsets
    past 2^34 -- in ONE launch each of the planned CSR scorer and of the fused kernel (every ORF one forward piece of the
    counts array itself: the same bytes read through the gather plan's chunk rows, byte offsets past 2^36).  Whole-set
    properties, fused == CSR on every ORF, and oracle slices of 20 000 ORFs at the head, the middle, the tail (entirely
    past 2^34 nt) and around 2^32, 2^33 and 2^34 nt.  The reference's loop has no size limit (detect_orfs.py:274-324)."""
    import torch

    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.gather import GatherPlan, IntervalTable
    from ribotricer_amd.synth import synth_csr_device

    n_orfs = 50_000_000
    counts, d_offsets = synth_csr_device(n_orfs, seed=20260213, cfg="cfg5", device="cuda:0")
    offsets = d_offsets.cpu().numpy()
    total_nt = int(offsets[-1])
    assert total_nt == counts.numel() > (1 << 34) + (1 << 22)
    lengths_np = np.diff(offsets)
    assert int(lengths_np.max()) == 100_000 and int((lengths_np > 30_000).sum()) >= 1_000  # the long tail is there
    eng = get_engine("cuda:0")
    th = make_filter()
    csr = eng.score(counts, d_offsets, thresholds=th, algo="tile")  # plan="auto": rp_plan_create_dev + rp_phase_score_csr_plan_dev
    torch.cuda.synchronize()
    table = IntervalTable(offsets[:-1].copy(), lengths_np.astype(np.int32), np.arange(n_orfs + 1, dtype=np.int64),
                          np.zeros(n_orfs, np.uint8), offsets)
    gplan = GatherPlan(table, counts.numel(), "cuda:0")
    assert gplan.stats()["slow_tiles"] == 0
    fused = eng.score_coverage(counts, gplan, thresholds=th)
    torch.cuda.synchronize()
    for k in ("phase", "valid", "read_count", "min_codon_cov", "flags", "status"):
        assert torch.equal(getattr(fused, k), getattr(csr, k)), k
    del fused, gplan
    # size-independent properties over the whole set
    assert int(csr.read_count.sum()) == int(counts.sum(dtype=torch.int64))
    lengths = d_offsets[1:] - d_offsets[:-1]
    assert bool((csr.valid <= lengths // 3).all()) and bool((csr.valid >= 0).all())
    assert bool(((csr.phase >= 0) & (csr.phase <= 1.0 + 1e-6)).all())
    assert bool((csr.min_codon_cov >= 0).all())
    expect_status = ((csr.phase >= 0.428571428571) & (csr.valid >= 5)).to(torch.uint8)
    assert torch.equal(csr.status, expect_status)
    assert 0 < int(csr.status.sum()) < n_orfs
    split = (csr.flags & 4) != 0  # RP_FLAG_SPLIT: the ORF spans tiles -- every ORF longer than a tile must carry it
    assert bool(split[lengths > 7936].all())
    ranges = {"head": (0, SLICE), "middle": (n_orfs // 2 - SLICE // 2, n_orfs // 2 + SLICE // 2), "tail": (n_orfs - SLICE, n_orfs)}
    for name, b in (("at_2^32", 1 << 32), ("at_2^33", 1 << 33), ("at_2^34", 1 << 34)):
        i = int(np.searchsorted(offsets, b))
        ranges[name] = (i - SLICE // 2, i + SLICE // 2)
    assert int(offsets[ranges["tail"][0]]) > (1 << 34)
    checked, worst, longest = 0, 0.0, 0
    for label, (lo, hi) in ranges.items():
        a, b = int(offsets[lo]), int(offsets[hi])
        c_host, o_host = counts[a:b].cpu().numpy(), offsets[lo : hi + 1] - offsets[lo]
        part = _cut(csr, lo, hi)
        o = assert_matches_oracle(part, c_host, o_host)
        worst = max(worst, float(np.abs(part["phase"] - o.phase).max(initial=0.0)))
        longest = max(longest, int(np.diff(o_host).max()))
        checked += hi - lo
    # ... and the 200 longest ORFs of the set (the k_rewalk_long / many-tile end of the law), wherever they lie
    top = np.sort(np.argsort(lengths_np)[-200:])
    sub_off = np.zeros(top.size + 1, np.int64)
    np.cumsum(lengths_np[top], out=sub_off[1:])
    sub = np.concatenate([counts[int(offsets[i]) : int(offsets[i + 1])].cpu().numpy() for i in top])
    idx = torch.from_numpy(top).to("cuda:0")
    part = {k: v[idx].cpu().numpy() for k, v in csr._asdict().items() if v is not None}
    assert_matches_oracle(part, sub, sub_off)
    print(f"configs[4]: {n_orfs} ORFs, {total_nt} nt, {checked + top.size} ORFs against the oracle, max |dphase| {worst:.2e}, longest ORF in a slice {longest}")
