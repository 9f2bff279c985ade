"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, and fails loudly (no fallback) when no HIP device is present.  No compute
call is made here."""

import ctypes
import os
import re

import pytest

from conftest import REPO
from ribotricer_amd import _lib

HEADER = os.path.join(REPO, "include", "ribophase.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `make -C ribotricer_amd/csrc` (build() does)"
    assert os.path.dirname(_lib.LIB_PATH).startswith(REPO)


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = header_symbols()
    assert declared, "no rp_* declarations found in the header"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ribophase.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "Python binding table and header disagree"


def test_stream_probe_is_a_separate_library_that_only_measuring_code_loads():
    """csrc/stream_probe.hip (bench.py's yardstick) is built beside the product library, exports its three
    entry points, and no product module refers to it."""
    path = os.path.join(REPO, "ribotricer_amd", "csrc", "libstreamprobe.so")
    assert os.path.exists(path), "run `make -C ribotricer_amd/csrc` (build() does)"
    lib = ctypes.CDLL(path)
    for name in ("sp_stream_read", "sp_stream_read_lds", "sp_stream_rw"):
        assert hasattr(lib, name), name
    pkg = os.path.join(REPO, "ribotricer_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f != "_probe.py":
            text = open(os.path.join(pkg, f)).read()
            assert not re.search(r"(from\s+\.?_probe\b|from\s+\.\s+import\s+[^\n]*\b_probe\b|import\s+[^\n]*\b_probe\b|libstreamprobe)", text), \
                f"{f} must not use the measuring stick"


def test_version_and_status_strings():
    import ribotricer_amd

    assert _lib.version() == "0.4.0" == ribotricer_amd.__version__  # one version string: the header's
    lib = _lib.load()
    assert lib.rp_status_string(0) == b"ok"
    assert lib.rp_status_string(-3) == b"bad CSR offsets"


def test_filter_defaults_match_reference_constants():
    fp = _lib.filter_defaults()
    # ribotricer/const.py:20-39
    assert fp.phase_score_cutoff == 0.428571428571
    assert fp.min_valid_codons == 5
    assert fp.min_reads_per_codon == 0 and fp.min_valid_codons_ratio == 0 and fp.min_density_over_orf == 0.0


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.load()
    assert lib.rp_workspace_bytes(1, 1, 0, None) == -1
    out = ctypes.c_size_t()
    assert lib.rp_workspace_bytes(-1, 1, 0, ctypes.byref(out)) == -2
    assert lib.rp_workspace_bytes(1, 1, 99, ctypes.byref(out)) == -8
    assert b"unknown algo" in lib.rp_last_error()
    assert lib.rp_workspace_bytes(10, 3000, _lib.RP_ALGO_WAVE, ctypes.byref(out)) == 0 and out.value == 0
    assert lib.rp_workspace_bytes(10, 3000, _lib.RP_ALGO_TILE, ctypes.byref(out)) == 0 and out.value > 0
    assert lib.rp_device_count(None) == -1
    assert lib.rp_workspace_bytes(1, 1, 3, ctypes.byref(out)) == -8  # the round-1 RP_ALGO_TILE_PIPE is gone
    assert lib.rp_plan_bytes(10, 3000, ctypes.byref(out)) == 0 and out.value >= 128 + 2 * 8
    assert lib.rp_plan_bytes(10, 3000, None) == -1
    handle = ctypes.c_void_p()
    assert lib.rp_plan_create_dev(0, None, 10, 3000, 0, None, 0, None, ctypes.byref(handle)) == -1
    assert lib.rp_phase_score_csr_plan_dev(None, None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    lib.rp_plan_free(None)
    # the compact coverage's block map: blocks of 1 ... 64 positions, powers of two; one bit per block + 8 bytes per 64 blocks
    assert lib.rp_coverage_map_bytes(1 << 20, 64, ctypes.byref(out)) == 0
    coarse = out.value
    assert lib.rp_coverage_map_bytes(1 << 20, 1, ctypes.byref(out)) == 0
    assert 60 * coarse > out.value > 20 * coarse and out.value >= (1 << 20) // 4
    for bad in (0, 3, 128, -1):
        assert lib.rp_coverage_map_bytes(1 << 20, bad, ctypes.byref(out)) == -8 and b"power of two" in lib.rp_last_error()
    assert lib.rp_coverage_map_bytes(-1, 1, ctypes.byref(out)) == -2
    assert lib.rp_coverage_map_bytes(1 << 20, 1, None) == -1


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.RibophaseError) as e:
        _lib.device_count()
    assert e.value.status == -6
    from ribotricer_amd import engine

    with pytest.raises(_lib.RibophaseError):
        engine.phase_score_csr([1, 0, 0], [0, 3])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "ribotricer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "phase_oracle" not in text, f


def test_header_is_plain_c(tmp_path):
    """include/ribophase.h must compile as C (no C++/HIP/torch types in the boundary)."""
    import subprocess

    src = tmp_path / "t.c"
    src.write_text('#include "ribophase.h"\nint main(void){ rp_filter_params p; (void)p; return RP_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(REPO, "include"),
                           "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_tile_positions_rule():
    """rp_tile_positions (no device needed): short-ORF indexes get the smaller tile."""
    from ribotricer_amd import _lib

    assert _lib.tile_positions(1000, 100_000) == 6144      # mean 100 nt
    assert _lib.tile_positions(1000, 179_999) == 6144
    assert _lib.tile_positions(1000, 180_000) == 7936      # mean 180 nt and up
    assert _lib.tile_positions(11_000_000, 3_966_674_436) == 7936
    assert _lib.tile_positions(0, 0) == 7936
    # the sizing functions follow the same rule: a short-ORF index needs the bigger plan per nucleotide
    assert _lib.plan_bytes(1000, 100_000) > 0 and _lib.workspace_bytes(1000, 100_000, _lib.RP_ALGO_TILE) > 0


def test_bench_line_of_the_full_size_record_fits_the_drivers_tail():
    """bench.compact_line on the FULL record of the round's evidence run (profiles/r06_bench_default_detail.json: 11 M ORFs, every
    section on): the one JSON line must stay under the 8 KB the driver keeps of stdout, and carry the contract fields and the
    top-level scalars a reader of the flattened record needs.  (The GPU contract test checks the same on small runs.)"""
    import importlib.util
    import json
    import os

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(repo, "profiles", "r06_bench_default_detail.json")) as fh:
        full = json.load(fh)
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 8192, len(line)
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "kernel_ms", "finish_ms", "step_frac", "fused_step_frac", "fused_nested_step_frac",
                "projected_efficiency_g8", "value_first_allocation", "value_source_placed", "value_pipelined", "value_one_stream_repeat",
                "value_single_sample", "verify", "quality"):
        assert key in d, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "fetch_factor"):
        assert key in d["roofline"], key
    assert d["value"] == full["value"] and d["roofline"]["frac"] == full["roofline"]["frac"] and "workload" in d["config"]
