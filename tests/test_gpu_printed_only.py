"""RP_FILTER_PRINTED_ONLY (include/ribophase.h; engine.make_filter(printed_only=True)): in the reference's default mode only
translating rows are printed (detect_orfs.py:301-302), so a too-close-to-call ORF that NO resolution of its frame decision
could make translating is left at its fp32 result instead of being re-walked in float64 / replayed.  The contract tested
here: the status of every ORF, and every output of every ORF that is translating or unflagged, are bit for bit what the
full resolution gives; the ORFs left open (RP_FLAG_UNRESOLVED) are exactly the too-close-to-call ones whose status is 0
under every outcome (detect_orfs.py:289-299 applied to the per-frame scores and N of statistics.py:67-113)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("phase", "valid", "read_count", "min_codon_cov", "status")
THRESHOLDS = {
    "default": dict(),
    "strict": dict(phase_score_cutoff=0.3, min_valid_codons=8, min_reads_per_codon=0, min_valid_codons_ratio=0.1, min_density_over_orf=0.02),
    "lenient": dict(phase_score_cutoff=0.05, min_valid_codons=1),
}


def one_exon_plan(counts, offsets):
    """Every ORF one forward exon of a coverage that IS the counts array (the fused path on CSR data)."""
    from ribotricer_amd.gather import GatherPlan, IntervalTable

    n = offsets.size - 1
    table = IntervalTable(offsets[:-1].copy(), np.diff(offsets).astype(np.int32), np.arange(n + 1, dtype=np.int64), np.zeros(n, np.uint8), offsets)
    return GatherPlan(table, counts.size)


def cannot_translate_bounds(frames, res, lengths, kw, slack):
    """The rule of rp_device.hpp::cannot_be_translating on float64 per-frame scores, with the phase bound moved by +-slack:
    (must, may) -- ORFs that are out for certain even with a tighter margin / that could be out with a looser one."""
    from ribotricer_amd.const import CUTOFF, MINIMUM_VALID_CODONS

    cutoff = kw.get("phase_score_cutoff", CUTOFF)
    n_max = frames["n"].max(axis=1)
    n_codons = np.maximum(1, lengths // 3)
    by_ints = (n_max < kw.get("min_valid_codons", MINIMUM_VALID_CODONS)) | (res["min_codon_cov"] < kw.get("min_reads_per_codon", 0))
    by_ints |= (n_max / n_codons < kw.get("min_valid_codons_ratio", 0)) | (res["read_count"] / n_codons < kw.get("min_density_over_orf", 0.0))
    phase_max = np.sqrt(np.nan_to_num(frames["score"], nan=0.0).max(axis=1))
    return by_ints | (phase_max + 2e-5 + slack < cutoff), by_ints | (phase_max + 2e-5 - slack < cutoff)


@pytest.mark.parametrize("name", list(THRESHOLDS))
@pytest.mark.parametrize("path", ["csr", "fused"])
def test_printed_only_leaves_open_exactly_what_cannot_translate(name, path):
    import torch

    from ribotricer_amd import _lib
    from ribotricer_amd.engine import get_engine, make_filter
    from ribotricer_amd.synth import synth_csr_host

    counts, offsets = synth_csr_host(400_000, seed=606, cfg="cfg3")
    keep = np.diff(offsets) > 0
    assert keep.all()
    eng = get_engine("cuda:0")
    kw = THRESHOLDS[name]
    d_counts = torch.from_numpy(counts).cuda()
    d_offsets = torch.from_numpy(offsets).cuda()
    out = {}
    for tag, printed_only in (("full", False), ("printed", True)):
        thr = make_filter(**kw, printed_only=printed_only)
        if path == "csr":
            r = eng.score(d_counts, d_offsets, thresholds=thr, algo="tile")
        else:
            r = eng.score_coverage(d_counts, one_exon_plan(counts, offsets), thresholds=thr)
        out[tag] = r.cpu_numpy()
        torch.cuda.synchronize()
    full, printed = out["full"], out["printed"]
    fr = eng.frames(d_counts, d_offsets)
    frames = {"score": fr.score.cpu().numpy(), "n": fr.n.cpu().numpy()}
    open_ = (printed["flags"] & _lib.FLAG_UNRESOLVED) != 0
    assert not (full["flags"] & _lib.FLAG_UNRESOLVED).any()  # (never without the filter flag)
    # 1. the status of EVERY ORF, and everything about the ORFs not left open, bit for bit
    assert np.array_equal(full["status"], printed["status"])
    for k in KEYS:
        assert np.array_equal(full[k][~open_], printed[k][~open_]), k
    assert np.array_equal(full["flags"][~open_], printed["flags"][~open_])
    # 2. the ORFs left open: nontranslating, their integers exact, their phase the fp32 tile sums' (well within 1e-5)
    assert open_.any() and not printed["status"][open_].any() and not full["status"][open_].any()
    for k in ("read_count", "min_codon_cov"):
        assert np.array_equal(full[k][open_], printed[k][open_])
    assert np.abs(full["phase"][open_] - printed["phase"][open_]).max() <= 1e-5
    assert not (printed["flags"][open_] & (_lib.FLAG_RECHECK64 | _lib.FLAG_REPLAY)).any()
    # 3. exactly the too-close-to-call ORFs (re-walked in the full run) that cannot translate under any outcome
    rewalked = (full["flags"] & _lib.FLAG_RECHECK64) != 0
    must, may = cannot_translate_bounds(frames, full, np.diff(offsets), kw, slack=2e-6)
    assert not (open_ & ~rewalked).any()
    assert not (rewalked & must & ~open_).any(), "a re-walk was spent on an ORF that cannot translate"
    assert not (open_ & ~may).any(), "an ORF that might translate was left open"
    # (how much of the finish pass's tail this removes: recorded in DESIGN.md from bench.py's `quality`)
    assert open_.sum() >= (0.5 if name != "lenient" else 0.0) * rewalked.sum()


def test_default_mode_export_is_the_same_file_with_and_without_the_shortcut(tmp_path, monkeypatch):
    """export_orf_coverages in default mode sets RP_FILTER_PRINTED_ONLY; RIBOTRICER_AMD_PRINTED_ONLY=0 switches it off.  Same
    bytes on a 40 000-ORF random index (one GPU and three slices), and `unresolved_orfs` says the shortcut was taken."""
    from test_gpu_export import _random_index_and_columns

    from ribotricer_amd import detect_orfs as d

    index = str(tmp_path / "rnd_candidate_orfs.tsv")
    cols = _random_index_and_columns(index, seed=2026)
    texts, open_counts = {}, {}
    for tag, env, devices in (("on", "1", None), ("off", "0", None), ("on3", "1", [0, 0, 0]), ("off3", "0", [0, 0, 0])):
        monkeypatch.setenv("RIBOTRICER_AMD_PRINTED_ONLY", env)
        d.forget_indexes()
        timings = {}
        d.export_orf_coverages(index, cols, str(tmp_path / tag), devices=devices, timings=timings)
        texts[tag] = open(str(tmp_path / tag) + "_translating_ORFs.tsv", "rb").read()
        open_counts[tag] = timings["unresolved_orfs"]
    d.forget_indexes()
    assert texts["on"] == texts["off"] and texts["on3"] == texts["off3"]
    assert open_counts["off"] == 0 and open_counts["off3"] == 0 and open_counts["on"] > 0 and open_counts["on3"] > 0
    # report_all never takes it: every row is printed
    timings = {}
    d.export_orf_coverages(index, cols, str(tmp_path / "all"), report_all=True, timings=timings)
    assert timings["unresolved_orfs"] == 0
    d.forget_indexes()
