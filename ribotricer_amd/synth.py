"""Seeded synthetic CSR P-site batches for the BASELINE.json configs.

cfg2  1 M ORFs, L = 3*k, k = max(20, round(lognormal(ln 80, 0.7)))  (mean ~300 nt,
      min 60 nt = the reference CLI's --min_orf_length default, cli.py:64-69);
      counts ~ Poisson(lambda_i * w_i[p mod 3]); lambda_i in {0 (20 % empty ORFs),
      0.05, 0.3, 2.0}; frame weights (2, .5, .5) for half the ORFs, flat otherwise.
cfg3  11 M ORFs, sigma = 0.9, 1 % of the ORFs with L % 3 != 0 (incomplete CDS).
cfg5  long-tail stress: lognormal body (mean ~100 codons) + Pareto(1.5) tail clipped
      at 33 333 codons (100 k nt).
gencode  a candidate-index-like law (prepare_orfs.py:217 emits every uORF / dORF / overlapping
      ORF >= --min_orf_length, cli.py:64-69): 60 % of the ORFs short, uniform 20..50 codons
      (60-150 nt), the rest cfg3's lognormal with median 120 codons.
gencode_short  the same with 70 % short ORFs (a uORF / dORF-dominated candidate set): mean ~200 nt.
short150  every ORF 60-150 nt (uniform).
orf60  every ORF 60 nt, the CLI minimum: the worst case for per-segment overheads.

Lengths always come from numpy (identical on every machine for a given seed).  Counts
come from numpy on the host (tests: the oracle and the GPU must see the same bytes) or
from torch on the device (bench: no 13 GB H2D copy).
"""

from __future__ import annotations

import numpy as np

CONFIGS = {
    "cfg2": dict(sigma=0.7, median_codons=80, frac_non_mult3=0.0, pareto_frac=0.0),
    "cfg3": dict(sigma=0.9, median_codons=80, frac_non_mult3=0.01, pareto_frac=0.0),
    "cfg5": dict(sigma=0.7, median_codons=78, frac_non_mult3=0.01, pareto_frac=0.02),
    "gencode": dict(sigma=0.9, median_codons=120, frac_non_mult3=0.01, pareto_frac=0.0, short_frac=0.6),
    "gencode_short": dict(sigma=0.9, median_codons=100, frac_non_mult3=0.01, pareto_frac=0.0, short_frac=0.7),
    "short150": dict(sigma=0.9, median_codons=80, frac_non_mult3=0.0, pareto_frac=0.0, short_frac=1.0),
    "orf60": dict(sigma=0.9, median_codons=80, frac_non_mult3=0.0, pareto_frac=0.0, short_frac=1.0, short_codons=(20, 21)),
}
LAMBDAS = np.array([0.0, 0.05, 0.3, 2.0])
LAMBDA_P = np.array([0.2, 0.3, 0.3, 0.2])
MAX_CODONS = 33333


def orf_lengths(n_orfs: int, seed: int, cfg: str = "cfg2") -> np.ndarray:
    c = CONFIGS[cfg]
    rng = np.random.default_rng(seed)
    k = np.rint(rng.lognormal(np.log(c["median_codons"]), c["sigma"], size=n_orfs))
    if c["pareto_frac"] > 0:
        tail = rng.random(n_orfs) < c["pareto_frac"]
        k[tail] = 300.0 * (1.0 + rng.pareto(1.5, size=int(tail.sum())))
    if c.get("short_frac", 0.0) > 0:
        short = rng.random(n_orfs) < c["short_frac"]
        lo, hi = c.get("short_codons", (20, 51))
        k[short] = rng.integers(lo, hi, size=int(short.sum()))
    k = np.clip(k, 20, MAX_CODONS).astype(np.int64)
    lengths = 3 * k
    if c["frac_non_mult3"] > 0:
        odd = rng.random(n_orfs) < c["frac_non_mult3"]
        lengths[odd] += rng.integers(1, 3, size=int(odd.sum()))
    return lengths


def orf_rates(n_orfs: int, seed: int):
    """Per-ORF Poisson rate and whether the ORF is framed (weights 2/.5/.5)."""
    rng = np.random.default_rng(seed + 1)
    lam = rng.choice(LAMBDAS, size=n_orfs, p=LAMBDA_P)
    framed = rng.random(n_orfs) < 0.5
    return lam, framed


def offsets_from_lengths(lengths: np.ndarray) -> np.ndarray:
    offsets = np.zeros(lengths.size + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    return offsets


def synth_csr_host(n_orfs: int, seed: int = 20260213, cfg: str = "cfg2"):
    """(counts int32, offsets int64) as numpy arrays."""
    lengths = orf_lengths(n_orfs, seed, cfg)
    offsets = offsets_from_lengths(lengths)
    lam, framed = orf_rates(n_orfs, seed)
    total = int(offsets[-1])
    orf_id = np.repeat(np.arange(n_orfs), lengths)
    frame = (np.arange(total, dtype=np.int64) - offsets[:-1][orf_id]) % 3
    w = np.where(framed[orf_id], np.where(frame == 0, 2.0, 0.5), 1.0)
    rate = lam[orf_id] * w
    rng = np.random.default_rng(seed + 2)
    counts = rng.poisson(rate).astype(np.int32)
    return counts, offsets


DEVICE_CHUNK_ORFS = 1_000_000  # fixed chunk grid of the device generator (part of the data definition)


def synth_csr_device(n_orfs: int, seed: int = 20260213, cfg: str = "cfg2", device="cuda", orf_range=None):
    """(counts int32, offsets int64) as device tensors; counts drawn by torch.poisson on the GPU.

    The ORF set is defined on a fixed grid of ``DEVICE_CHUNK_ORFS``-ORF chunks, chunk c drawn
    from its own generator seeded ``seed + 2 + 7919 * c``, so any rank can materialise any
    ORF-index slice ``orf_range=(lo, hi)`` of the SAME set without generating the rest
    (BASELINE configs[3]: one 11 M-ORF set sharded over the GPUs).  With ``orf_range`` the
    returned offsets are rebased to 0 and cover ORFs lo..hi only.
    """
    import torch

    dev = torch.device(device)
    lengths = orf_lengths(n_orfs, seed, cfg)
    offsets_np = offsets_from_lengths(lengths)
    lam_np, framed_np = orf_rates(n_orfs, seed)
    lo_all, hi_all = (0, n_orfs) if orf_range is None else (int(orf_range[0]), int(orf_range[1]))
    base = int(offsets_np[lo_all])
    total = int(offsets_np[hi_all]) - base
    counts = torch.empty(total, dtype=torch.int32, device=dev)
    for c in range(lo_all // DEVICE_CHUNK_ORFS, (max(hi_all, lo_all + 1) - 1) // DEVICE_CHUNK_ORFS + 1):
        lo, hi = c * DEVICE_CHUNK_ORFS, min(n_orfs, (c + 1) * DEVICE_CHUNK_ORFS)
        if hi <= lo:
            continue
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed + 2 + 7919 * c)
        ln = torch.from_numpy(lengths[lo:hi]).to(dev)
        off = torch.from_numpy(offsets_np[lo:hi] - offsets_np[lo]).to(dev)
        lam = torch.from_numpy(lam_np[lo:hi].astype(np.float32)).to(dev)
        framed = torch.from_numpy(framed_np[lo:hi]).to(dev)
        orf_id = torch.repeat_interleave(torch.arange(hi - lo, device=dev), ln)
        pos = torch.arange(orf_id.numel(), device=dev) - off[orf_id]
        w = torch.where(framed[orf_id], torch.where(pos % 3 == 0, 2.0, 0.5), 1.0)
        rate = lam[orf_id] * w
        chunk = torch.poisson(rate, generator=gen).to(torch.int32)
        # the part of this chunk that falls inside [lo_all, hi_all)
        a, b = max(lo, lo_all), min(hi, hi_all)
        if b > a:
            src0 = int(offsets_np[a] - offsets_np[lo])
            n_nt = int(offsets_np[b] - offsets_np[a])
            dst0 = int(offsets_np[a]) - base
            counts[dst0 : dst0 + n_nt] = chunk[src0 : src0 + n_nt]
        del ln, off, lam, framed, orf_id, pos, w, rate, chunk
    offsets = torch.from_numpy(offsets_np[lo_all : hi_all + 1] - base).to(dev)
    return counts, offsets


# ------------------------------------------------------------------------------------------------
# A synthetic candidate-ORF INDEX over a dense coverage array (what detect_orfs.py:134-203 walks:
# exon intervals per ORF, '-' strand ORFs read backwards), for the fused gather + score path at
# BASELINE sizes.  Profiles are windows onto ONE coverage array -- the expected profile of an ORF is
# a host-side gather of its intervals (``profiles_from_coverage``), independent of every kernel.
# ------------------------------------------------------------------------------------------------
EXON_P = np.array([0.35, 0.30, 0.20, 0.15])  # 1..4 exons per ORF
COVERAGE_BLOCK = 2048  # positions that share one Poisson rate / frame weighting
COVERAGE_CHUNK = 1 << 26  # positions drawn per device generator (part of the data definition)


def synth_exon_layout(lengths: np.ndarray, seed: int, max_intron: int = 300, max_spacer: int = 64, reverse_frac: float = 0.5):
    """ORFs laid one after the other along a coverage array: every ORF is cut into 1-4 exons at
    random positions, ``[0, max_intron)`` unused positions between its exons, ``[0, max_spacer)``
    between ORFs, ``reverse_frac`` of them on the '-' strand.  Returns ``(iv_start int64[], iv_len
    int32[], orf_iv int64[n+1], reverse uint8[n], offsets int64[n+1], coverage_len)`` -- the fields
    of ``gather.IntervalTable`` plus the coverage length.  Vectorised numpy: ~1 s per million ORFs."""
    lengths = np.asarray(lengths, np.int64)
    n = lengths.size
    rng = np.random.default_rng(seed + 11)
    k = rng.choice(np.arange(1, 5), size=n, p=EXON_P)
    # cut points in [1, L-1]: unused columns (and ORFs too short to cut) get the sentinel L
    cuts = 1 + np.floor(rng.random((n, 3)) * np.maximum(lengths - 1, 1)[:, None]).astype(np.int64)
    cuts[np.arange(3)[None, :] >= (k - 1)[:, None]] = -1
    cuts = np.where((cuts < 0) | (lengths[:, None] < 2), lengths[:, None], cuts)
    cuts.sort(axis=1)
    edges = np.concatenate([np.zeros((n, 1), np.int64), cuts, lengths[:, None]], axis=1)
    exon = np.diff(edges, axis=1)  # (n, 4), zeros where cut points coincide / are unused
    live = exon > 0
    n_ex = live.sum(axis=1)
    orf_iv = np.zeros(n + 1, np.int64)
    np.cumsum(n_ex, out=orf_iv[1:])
    iv_len = exon[live]  # row-major: exons of ORF 0, then ORF 1, ... in ascending coverage order
    n_iv = iv_len.size
    gap = rng.integers(0, max(1, max_intron), size=n_iv)
    first = np.zeros(n_iv, bool)
    first[orf_iv[:-1][n_ex > 0]] = True
    gap[first] = rng.integers(0, max(1, max_spacer), size=int(first.sum()))
    iv_start = np.cumsum(gap + iv_len) - iv_len  # every interval starts `gap` after the previous one's end
    reverse = (rng.random(n) < reverse_frac).astype(np.uint8)
    offsets = offsets_from_lengths(lengths)
    coverage_len = int(iv_start[-1] + iv_len[-1]) + 16 if n_iv else 16
    return iv_start.astype(np.int64), iv_len.astype(np.int32), orf_iv, reverse, offsets, coverage_len


def synth_nested_layout(n_orfs: int, seed: int, n_groups: int = 48):
    """The NESTED candidate-index law of scripts/gen_big_index.cpp without the text round trip: transcripts of 1-8
    exons (60-420 nt, introns 80-3000) with 1-10 candidate ORFs each -- nested sub-ranges of the transcript, as
    prepare-orfs emits them, so the ORFs of a transcript share coverage --, 65 % of the ORFs 60-150 nt, the others to
    1 800 nt; consecutive transcripts sit on different (chromosome, strand) groups (``n_groups`` of them, laid one
    after the other in the dense coverage: gigabytes apart).  Returns the fields of ``gather.IntervalTable`` plus the
    coverage length, like :func:`synth_exon_layout`.  Vectorised numpy, ~1.5 s per million ORFs."""
    rng = np.random.default_rng(seed + 31)
    n_tx = max(1, int(n_orfs / 5.3) + 64)  # (1-10 ORFs per transcript, a few dropped as too short: ~5.4 kept on average)
    while True:
        n_ex = rng.integers(1, 9, size=n_tx)
        n_here = rng.integers(1, 11, size=n_tx)
        if int(n_here.sum()) >= int(n_orfs * 1.02) + 16:
            break
        n_tx = int(n_tx * 1.1) + 16
    ex_tx = np.repeat(np.arange(n_tx), n_ex)
    ex_first = np.zeros(n_tx + 1, np.int64)
    np.cumsum(n_ex, out=ex_first[1:])
    ex_len = rng.integers(60, 421, size=ex_tx.size).astype(np.int64)
    intron = rng.integers(80, 3001, size=ex_tx.size).astype(np.int64)
    tlen = np.add.reduceat(ex_len, ex_first[:-1])
    # transcript coordinates, all transcripts in ONE ascending space: exon k covers [e_t0[k], e_t0[k] + ex_len[k])
    e_t0 = np.cumsum(ex_len) - ex_len
    tx_t0 = e_t0[ex_first[:-1]]
    # genomic (coverage) coordinates: transcript t lies in group t % n_groups, behind the group's earlier transcripts
    span = np.add.reduceat(ex_len + intron, ex_first[:-1]) + rng.integers(200, 5001, size=n_tx)
    group = np.arange(n_tx) % n_groups
    order = np.argsort(group, kind="stable")  # group-major order of the transcripts = their order in the coverage
    start_in_cov = np.empty(n_tx, np.int64)
    start_in_cov[order] = np.cumsum(span[order]) - span[order]
    within = np.cumsum(ex_len + intron) - (ex_len + intron)  # exon start relative to ... (made per transcript below)
    within -= within[ex_first[:-1]][ex_tx]
    e_g0 = start_in_cov[ex_tx] + within + 16
    coverage_len = int(span.sum()) + 64
    # candidate ORFs: [a, a + len) in the transcript's own coordinates
    orf_tx = np.repeat(np.arange(n_tx), n_here)
    short = rng.random(orf_tx.size) < 0.65
    length = 3 * np.where(short, rng.integers(20, 51, size=orf_tx.size), rng.integers(51, 601, size=orf_tx.size)).astype(np.int64)
    length = np.minimum(length, tlen[orf_tx] // 3 * 3)
    keep = length >= 60
    orf_tx, length = orf_tx[keep][:n_orfs], length[keep][:n_orfs]
    if orf_tx.size < n_orfs:
        raise RuntimeError("synth_nested_layout: too few ORFs drawn")
    a = np.floor(rng.random(n_orfs) * (tlen[orf_tx] - length + 1)).astype(np.int64)
    t_a = tx_t0[orf_tx] + a
    t_b = t_a + length
    k0 = np.searchsorted(e_t0, t_a, side="right") - 1
    k1 = np.searchsorted(e_t0, t_b - 1, side="right") - 1
    n_iv = k1 - k0 + 1
    orf_iv = np.zeros(n_orfs + 1, np.int64)
    np.cumsum(n_iv, out=orf_iv[1:])
    iv_orf = np.repeat(np.arange(n_orfs), n_iv)
    iv_k = k0[iv_orf] + (np.arange(int(orf_iv[-1])) - orf_iv[iv_orf])
    lo = np.maximum(t_a[iv_orf], e_t0[iv_k])
    hi = np.minimum(t_b[iv_orf], e_t0[iv_k] + ex_len[iv_k])
    iv_start = e_g0[iv_k] + (lo - e_t0[iv_k])
    iv_len = (hi - lo).astype(np.int32)
    reverse = ((orf_tx // (n_groups // 2)) % 2).astype(np.uint8) if n_groups >= 2 else np.zeros(n_orfs, np.uint8)
    offsets = offsets_from_lengths(length)
    return iv_start.astype(np.int64), iv_len, orf_iv, reverse, offsets, coverage_len


def synth_coverage_device(coverage_len: int, seed: int, device="cuda"):
    """Dense int32 coverage drawn on the device: blocks of ``COVERAGE_BLOCK`` positions share a rate
    from ``LAMBDAS`` (so sparse and dense stretches -- exact frame ties and clear winners -- both
    occur), half of the blocks carry the (2, .5, .5) position-mod-3 weighting."""
    import torch

    dev = torch.device(device)
    n_blocks = (coverage_len + COVERAGE_BLOCK - 1) // COVERAGE_BLOCK
    rng = np.random.default_rng(seed + 21)
    lam = torch.from_numpy(rng.choice(LAMBDAS, size=n_blocks, p=LAMBDA_P).astype(np.float32)).to(dev)
    framed = torch.from_numpy(rng.random(n_blocks) < 0.5).to(dev)
    cov = torch.empty(coverage_len, dtype=torch.int32, device=dev)
    for c, lo in enumerate(range(0, coverage_len, COVERAGE_CHUNK)):
        hi = min(coverage_len, lo + COVERAGE_CHUNK)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed + 23 + 7919 * c)
        pos = torch.arange(lo, hi, device=dev)
        blk = pos // COVERAGE_BLOCK
        w = torch.where(framed[blk], torch.where(pos % 3 == 0, 2.0, 0.5), 1.0)
        cov[lo:hi] = torch.poisson(lam[blk] * w, generator=gen).to(torch.int32)
        del pos, blk, w
    return cov


def profiles_from_coverage(cov_window: np.ndarray, window_lo: int, iv_start, iv_len, orf_iv, reverse, lo: int, hi: int):
    """Host-side gather of ORFs ``lo..hi`` (detect_orfs.py:176-202 on a dense array): the exon
    intervals in ascending order, the whole profile reversed for a '-' strand ORF.  ``cov_window``
    holds coverage positions ``window_lo ..``.  Returns CSR ``(counts int32, offsets int64)``."""
    parts, lens = [], []
    for i in range(lo, hi):
        k0, k1 = int(orf_iv[i]), int(orf_iv[i + 1])
        p = [cov_window[int(iv_start[k]) - window_lo : int(iv_start[k]) - window_lo + int(iv_len[k])] for k in range(k0, k1)]
        prof = np.concatenate(p) if p else np.zeros(0, np.int32)
        if reverse[i]:
            prof = prof[::-1]
        parts.append(prof)
        lens.append(prof.size)
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    return (np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)), off
