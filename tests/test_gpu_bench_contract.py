"""bench.py prints ONE JSON line with the fields the driver and the judge read."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 8192  # the driver keeps an 8 KB tail of stdout: the whole line must fit


def test_bench_line_has_the_contract_fields(tmp_path):
    detail_path = str(tmp_path / "detail.json")
    out = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--orfs", "60000", "--steps", "3", "--warmup", "1", "--cpu-sample", "100",
         "--cpu-cores", "4", "--detail", detail_path],
        capture_output=True, text=True, timeout=600, cwd=REPO,
    )
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "ORFs/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] == 4 and c["value"] > 0
    assert d["cpu_baseline_1core"]["cores"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "configs[2]" in d["config"]["workload"] and d["config"]["orfs_total"] == 60000
    assert d["roofline"]["kernel"] == "rp::k_tile_score"
    assert d["roofline"]["stream_read_GBps"] > 0 and 0 < d["roofline"]["frac_of_stream_read"] < 1.5  # the live yardstick: a plain read of the same buffer
    # the numbers a reader of the driver's flattened record needs, as top-level scalars
    for key in ("kernel_ms", "finish_ms", "step_frac", "fused_step_frac", "fused_nested_step_frac", "fused_finish_ms", "fused_nested_finish_ms",
                "fused_step_frac_all_resolved", "projected_efficiency_g2", "projected_efficiency_g4", "projected_efficiency_g8", "value_single_sample"):
        assert isinstance(d[key], float) and d[key] > 0, key
    assert d["kernel_ms"] == r["kernel_ms"] and d["finish_ms"] == r["finish_ms"] and d["step_frac"] == r["step_frac"]
    # after the timed region: head / middle / tail slices against the oracle, the fused sections, the single-sample rate
    v = d["verify"]
    assert v["ok"] is True and v["orfs_checked"] >= 60000 and v["max_abs_dphase"] <= 1e-6 and v["read_count_checksum_ok"] is True
    for tag in ("fused", "fused_nested"):  # the export's default mode (RP_FILTER_PRINTED_ONLY), all-resolved beside it, each checked
        f = d[tag]
        assert f["kernel_ms"] > 0 and 0 < f["frac"] < 1 and f["verify"]["ok"] is True and f["verify"]["orfs_checked"] >= 60000
        assert f["printed_only_check_ok"] is True and 0 < f["left_open"] <= f["rewalked_when_all_resolved"]
        assert f["step_frac"] == d[f"{tag}_step_frac"] and f["all_resolved"]["step_frac"] == d[f"{tag}_step_frac_all_resolved"]
    assert d["quality"]["left_open"] == 0  # (the headline resolves every ORF)
    assert [row["gpus"] for row in d["slice_projection"]] == [2, 4, 8]
    # ---- the full record (--detail): every block with its explanations
    full = json.load(open(detail_path))
    assert full["value"] == d["value"] and full["roofline"]["stream_read"]["GBps"] == d["roofline"]["stream_read_GBps"]
    assert full["single_sample"]["ms_per_step"] > 0  # (a 60 000-ORF set is launch-bound: no ordering claim)
    sp = full["slice_projection"]  # rank 0's slice of the 2 / 4 / 8-GPU runs, timed on this GPU: the projected scaling curve
    assert [row["gpus"] for row in sp["slices"]] == [2, 4, 8] and sp["step_ms_1gpu"] > 0
    for row in sp["slices"]:
        for key in ("orfs", "nt", "step_ms", "kernel_ms", "finish_ms", "step_frac", "projected_value", "projected_efficiency"):
            assert row[key] is not None and row[key] > 0, (row["gpus"], key)
        assert abs(row["projected_value"] - 60000 / (row["step_ms"] * 1e-3)) < 1e-6 * row["projected_value"]
        assert abs(row["projected_efficiency"] - sp["step_ms_1gpu"] / (row["gpus"] * row["step_ms"])) < 1e-9
        assert row["integers_equal_headline"] is True and row["max_abs_dphase_vs_headline"] <= 1e-6
        assert abs(row["nt"] - d["config"]["nt_total"] / row["gpus"]) <= 0.02 * d["config"]["nt_total"]  # nt-balanced
        assert d[f"projected_efficiency_g{row['gpus']}"] == row["projected_efficiency"]
    fn = full["fused_nested"]  # the nested-index law (transcripts on different chromosomes: pieces of a tile gigabytes apart)
    assert fn["gather_plan"]["slow_tiles"] <= 0.01 * fn["gather_plan"]["tiles"] + 1 and "nested" in fn["workload"]
    assert fn["printed_only_check"]["ok"] is True and fn["verify"]["slices"]


def test_two_ranks_shard_one_set_and_concat_equals_whole():
    """BASELINE configs[3] in miniature: `bench.py --gpus 2` as two ranks (sharing the one GPU
    of this box, gloo for the control traffic) cut ONE seeded set with sharding.slice_bounds,
    score their slices, and rank 0 checks concat == whole on the device."""
    env = dict(os.environ, RP_BENCH_BACKEND="gloo")
    port = 29600 + os.getpid() % 300
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--orfs", "300000", "--steps", "3",
         "--warmup", "1"],
        capture_output=True, text=True, timeout=900, cwd=REPO, env=env,
    )
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["verify"]["ok"] is True and d["verify"]["orfs_checked"] == 300000 and d["verify"]["max_abs_dphase"] <= 1e-6
    assert len(d["per_rank"]) == 2 and sum(r["orfs"] for r in d["per_rank"]) == 300000
    nts = [r["nt"] for r in d["per_rank"]]
    assert abs(nts[0] - nts[1]) <= 0.02 * sum(nts)  # nt-balanced
    assert "configs[3]" in d["config"]["workload"] and "cpu_baseline" not in d


def _bare_bench(n, orfs, extra=(), env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--orfs", str(orfs), "--steps", "3", "--warmup", "1", *extra],
        capture_output=True, text=True, timeout=1500, cwd=REPO, env=env,
    )
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines  # stdout is the ONE JSON line (RCCL's banner and the like go to stderr)
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    return json.loads(lines[0])


def test_a_failing_rccl_probe_falls_back_to_gloo():
    """RCCL forced on two ranks that share the one GPU ("Duplicate GPU detected"): the probe fails on both ranks, they
    agree over gloo, and the run goes on -- the scaling line is still produced and says what happened.  (A node whose
    RCCL cannot come up must not cost the 1/2/4/8 curve.)"""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs ranks sharing a GPU to make RCCL fail")
    d = _bare_bench(2, 200000, env_extra={"RP_BENCH_BACKEND": "nccl"})
    assert d["n_gpus"] == 2 and d["verify"]["ok"] is True and d["verify"]["orfs_checked"] == 200000
    assert d["config"]["control_backend"].startswith("gloo (RCCL probe failed")


@pytest.mark.parametrize("n", [2, 8])
def test_bare_command_runs_n_ranks(n):
    """`python bench.py --gpus N` with NO launcher (the shape of the driver's N = 1 command): bench.py starts its
    own N ranks before touching the GPU; on this one-GPU box they share the device (gloo control traffic, the line
    says so).  Checks what the 8-GPU run will be judged on: n_gpus, nt balance, concat == whole, a roofline block."""
    import torch

    d = _bare_bench(n, 400000)
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["config"]["launcher"] == "bench.py self_launch"
    phys = torch.cuda.device_count()
    assert d["config"]["physical_gpus"] == phys and d["config"]["ranks_share_gpus"] == (n > phys)
    assert d["config"]["control_backend"] == ("nccl" if phys >= n else "gloo")  # (RCCL is probed first; "gloo (RCCL probe failed ...)" would say so)
    ranks = d["per_rank"]
    assert [r["rank"] for r in ranks] == list(range(n)) and sum(r["orfs"] for r in ranks) == 400000
    nts = [r["nt"] for r in ranks]
    assert sum(nts) == d["config"]["nt_total"] and max(nts) - min(nts) <= 0.02 * sum(nts) / n + 100000
    v = d["verify"]
    assert v["ok"] is True and v["orfs_checked"] == 400000 and v["max_abs_dphase"] <= 1e-6
    r = d["roofline"]  # the slowest rank's kernel, per launch and per GPU
    slow = max(ranks, key=lambda x: x["kernel_ms"])
    assert r["rank"] == slow["rank"] and r["kernel_ms"] == slow["kernel_ms"] and r["bound"] == "hbm"
    assert r["algorithmic_bytes_per_launch"] == 4 * slow["nt"] + 8 * (slow["orfs"] + 1) + 24 * slow["orfs"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0 and r["node_achieved"] >= r["achieved"]
    assert d["value"] == pytest.approx(400000 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3), rel=1e-9)
    assert "cpu_baseline" not in d


def test_one_rank_has_the_same_value_definition_with_and_without_a_process_group():
    """N = 1 plain, and N = 1 with the N-rank control flow forced on (RP_BENCH_FORCE_DIST=1: RCCL init, barrier and
    max-over-ranks on the one GPU -- the branch the 8-GPU run takes): `value` = ORFs of the whole job * steps /
    max-over-ranks wall time of the K steps in both."""
    plain = _bare_bench(1, 300000, extra=("--cpu-sample", "0", "--no-fused"))
    forced = _bare_bench(1, 300000, extra=("--cpu-sample", "0", "--no-fused"), env_extra={"RP_BENCH_FORCE_DIST": "1"})
    for d in (plain, forced):
        assert d["n_gpus"] == 1 and d["config"]["orfs_total"] == 300000 and d["verify"]["ok"] is True
        assert d["value"] == pytest.approx(300000 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3), rel=1e-9)
        assert d["roofline"]["rank"] == 0 and d["roofline"]["kernel_ms"] == d["per_rank"][0]["kernel_ms"]
    assert plain["config"]["control_backend"] is None and forced["config"]["control_backend"] == "nccl"
    assert forced["verify"]["concat_equals_whole"]["ok"] is True and "concat_equals_whole" not in plain["verify"]


def test_pipelined_and_first_allocation_values_are_reported_beside_value(tmp_path):
    """A set large enough for the placement search and the two-stream section.  `value` is timed with the record workspace
    placed (engine.tune_workspace: the engine's own buffer) and the counts WHERE THE CALLER PUT THEM; beside it, as extra
    fields: `value_first_allocation` (the step before the search), `value_source_placed` (the counts moved as well,
    engine.tune_source -- timed AFTER the headline, no product path does that), `value_pipelined` (two streams: finish(k)
    beside score(k + 1)) and `value_one_stream_repeat` (the same protocol on one stream right after it): both timings
    reported, neither chosen; the pipelined results equal the headline's bit for bit."""
    detail = str(tmp_path / "detail.json")
    d = _bare_bench(1, 600000, extra=("--cpu-sample", "0", "--no-fused", "--detail", detail))
    assert d["value"] == pytest.approx(600000 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3), rel=1e-9)
    assert d["value_first_allocation"] > 0 and d["first_allocation_ms_per_step"] > 0
    assert d["config"]["workspace_placement"]["searched"] is True
    full = json.load(open(detail))
    wp = full["config"]["workspace_placement"]
    assert wp["released_to_driver"] is True and wp["spacers"] == len(wp["step_ms"]) - 1
    assert wp["copies"] == 2  # (the second workspace of the chosen block serves the other stream of the two-stream trial)
    assert "source_placement" not in wp  # (nothing moved the counts before the headline)
    sp = full["source_placed"]["search"]  # (engine.tune_source: the synthetic counts placed like the workspace; same bytes)
    assert len(sp["step_ms"]) == len(sp["kernel_gbps"]) == sp["spacers"] + 1 <= 4 and 0 <= sp["chosen"] < len(sp["step_ms"])
    assert sp["step_ms"][sp["chosen"]] <= sp["step_ms"][0]
    if sp["chosen"]:
        assert d["value_source_placed"] == pytest.approx(600000 / (full["source_placed"]["ms_per_step"] * 1e-3), rel=1e-9)
    else:
        assert d["value_source_placed"] is None  # (the first place was the fast one already: nothing else to time)
    p = full["pipelined"]
    assert d["value_pipelined"] == pytest.approx(600000 / (p["two_streams_ms_per_step"] * 1e-3), rel=1e-9)
    assert d["value_one_stream_repeat"] == pytest.approx(600000 / (p["one_stream_ms_per_step"] * 1e-3), rel=1e-9)
    assert p["results_equal_headline"] is True and d["verify"]["ok"] is True and "ms_per_step" not in p  # (no best-of)
    assert p["workspaces_from_the_headline_placement"] == 2
