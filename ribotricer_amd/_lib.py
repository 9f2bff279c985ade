"""ctypes binding of libribophase.so (C ABI: include/ribophase.h).

The shared library is the product: there is no Python stand-in for it (the GPU-less backend,
``backend.py``, is the library's own ``*_host`` entry points).  Importing this module without a built
library raises immediately with the build command; calling a ``*_dev`` entry point without a usable HIP
device raises ``RibophaseError`` carrying the library's own message.
"""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RIBOPHASE_LIB selects another build of the same library (kernel A/B runs: scripts/ab_variants.py)
LIB_PATH = os.environ.get("RIBOPHASE_LIB") or os.path.join(_HERE, "csrc", "libribophase.so")

RP_OK = 0
RP_ALGO_AUTO, RP_ALGO_WAVE, RP_ALGO_TILE = 0, 1, 2
ALGOS = {"auto": RP_ALGO_AUTO, "wave": RP_ALGO_WAVE, "tile": RP_ALGO_TILE}

FLAG_TIE = 0x01
FLAG_RECHECK64 = 0x02
FLAG_SPLIT = 0x04
FLAG_REPLAY = 0x08
FLAG_BIGTIE = 0x10
FLAG_BIGCOUNT = 0x20
FLAG_UNRESOLVED = 0x40  # only with FILTER_PRINTED_ONLY: too close to call, left at the fp32 result because status is 0 for certain
FILTER_PRINTED_ONLY = 0x1  # rp_filter_params.flags: the caller prints translating ORFs only (include/ribophase.h)
MIN_CODON_COV_EMPTY = 2147483647
MAX_COUNT = 16777215
ERR_INTERVALS = -12
ERR_SIZE = -2


class RibophaseError(RuntimeError):
    """A libribophase entry point returned a negative rp_status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libribophase: {message} (status {status})")
        self.status = status


class IndexCoordinateError(RibophaseError, ValueError):
    """A coordinate field that does not parse (status RP_ERR_INDEX_COORD).  Also a ``ValueError``: that is what the
    reference raises for the same line (``start, end = group.split("-")`` / ``int(start)``, orf.py:165-168)."""


class FilterParams(ctypes.Structure):
    """rp_filter_params -- thresholds of detect_orfs.py:289-299 (defaults const.py:20-39)."""

    _fields_ = [
        ("phase_score_cutoff", ctypes.c_double),
        ("min_valid_codons_ratio", ctypes.c_double),
        ("min_density_over_orf", ctypes.c_double),
        ("min_reads_per_codon", ctypes.c_double),
        ("min_valid_codons", ctypes.c_int32),
        ("flags", ctypes.c_int32),  # FILTER_* bits
    ]


from .const import (  # noqa: E402
    CUTOFF,
    MINIMUM_DENSITY_OVER_ORF,
    MINIMUM_READS_PER_CODON,
    MINIMUM_VALID_CODONS,
    MINIMUM_VALID_CODONS_RATIO,
)


def make_filter(
    phase_score_cutoff: float = CUTOFF,
    min_valid_codons: int = MINIMUM_VALID_CODONS,
    min_reads_per_codon: float = MINIMUM_READS_PER_CODON,
    min_valid_codons_ratio: float = MINIMUM_VALID_CODONS_RATIO,
    min_density_over_orf: float = MINIMUM_DENSITY_OVER_ORF,
    printed_only: bool = False,
) -> FilterParams:
    """Thresholds in the argument order of export_orf_coverages (detect_orfs.py:206-216).

    ``printed_only`` (``RP_FILTER_PRINTED_ONLY``): the caller prints translating ORFs only (the reference's default mode,
    detect_orfs.py:301-302).  Too-close-to-call ORFs that no resolution could make translating are then left at their
    fp32 result (``FLAG_UNRESOLVED``, status 0) instead of being re-walked in float64; every ORF with status 1 and every
    unflagged ORF is unchanged.  Off by default: scores of ALL ORFs then carry the full resolution."""
    fp = FilterParams()
    fp.flags = FILTER_PRINTED_ONLY if printed_only else 0
    fp.phase_score_cutoff = float(phase_score_cutoff)
    fp.min_valid_codons = int(min_valid_codons)
    fp.min_reads_per_codon = float(min_reads_per_codon)
    fp.min_valid_codons_ratio = float(min_valid_codons_ratio)
    fp.min_density_over_orf = float(min_density_over_orf)
    return fp


# every symbol include/ribophase.h declares: name -> (restype, argtypes)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_SCORE_ARGS = [_int, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(FilterParams), _vp, ctypes.c_size_t, _int, _vp]
SYMBOLS = {
    "rp_version": (ctypes.c_char_p, []),
    "rp_last_error": (ctypes.c_char_p, []),
    "rp_status_string": (ctypes.c_char_p, [_int]),
    "rp_device_count": (_int, [ctypes.POINTER(_int)]),
    "rp_filter_defaults": (_int, [ctypes.POINTER(FilterParams)]),
    "rp_workspace_bytes": (_int, [_i64, _i64, _int, ctypes.POINTER(ctypes.c_size_t)]),
    "rp_phase_score_csr_dev": (_int, _SCORE_ARGS),
    "rp_phase_score_csr_dev_timed": (_int, _SCORE_ARGS[:-1] + [_vp, _vp, ctypes.POINTER(ctypes.c_float * 4)]),
    "rp_tile_positions": (_int, [_i64, _i64, ctypes.POINTER(ctypes.c_int32)]),
    "rp_plan_bytes": (_int, [_i64, _i64, ctypes.POINTER(ctypes.c_size_t)]),
    "rp_plan_create_dev": (_int, [_int, _vp, _i64, _i64, _int, _vp, ctypes.c_size_t, _vp, ctypes.POINTER(_vp)]),
    "rp_plan_free": (None, [_vp]),
    "rp_phase_score_csr_plan_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(FilterParams), _vp, ctypes.c_size_t, _vp]),
    "rp_phase_score_frames_dev": (_int, [_int, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "rp_phase_score_f64_csr_dev": (_int, [_int, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    # host side (no GPU): exact-tie replay with this host's libm
    "rp_tie_replay_host": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "rp_tie_replay_f64_host": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "rp_phase_score_csr_host": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(FilterParams), _int]),
    "rp_gather_profiles_dev": (_int, [_int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "rp_gather_plan_bytes": (_int, [_i64, _i64, _i64, ctypes.POINTER(ctypes.c_size_t)]),
    "rp_gather_plan_create_dev": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp, ctypes.POINTER(_vp)]),
    "rp_gather_plan_free": (None, [_vp]),
    "rp_gather_profiles_plan_dev": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "rp_gather_selected_plan_dev": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "rp_phase_score_coverage_dev": (_int, [_int, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(FilterParams),
                                           _vp, ctypes.c_size_t, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_float * 4)]),
    "rp_validate_csr_dev": (_int, [_int, _vp, _vp, _i64, _i64, _vp]),
    "rp_metagene_dev": (_int, [_int, _vp, _vp, _i64, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    "rp_metagene_host": (_int, [_vp, _vp, _i64, ctypes.c_int32, _vp, _vp, _vp]),
    "rp_coverage_build_dev": (_int, [_int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, ctypes.c_int32, _vp, _i64, _vp, ctypes.POINTER(ctypes.c_int32)]),
    "rp_coverage_build_rows_dev": (_int, [_int, _vp, _vp, _vp, _vp, _i64, _vp, ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, _vp, _i64, _vp,
                                          ctypes.POINTER(ctypes.c_int32), _vp, _i64, ctypes.c_int32]),
    "rp_coverage_map_bytes": (_int, [_i64, ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)]),
    "rp_coverage_map_create_dev": (_int, [_int, _vp, _vp, _i64, _i64, ctypes.c_int32, _vp, ctypes.c_size_t, _vp, ctypes.POINTER(_i64)]),
    "rp_coverage_map_remap_dev": (_int, [_int, _vp, _i64, _vp, _i64, ctypes.c_int32, _vp]),
    "rp_coverage_big_positions_dev": (_int, [_int, _vp, _i64, _vp, _i64, ctypes.POINTER(_i64), _vp]),
    # host side (no GPU): TSV row rendering
    "rp_format_rows_host": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp,
                                   ctypes.c_size_t, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_size_t)]),
    "rp_index_parse_host": (_int, [_vp, ctypes.c_size_t, _int, ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "rp_index_view_host": (_int, [_vp, _vp]),
    "rp_index_free": (None, [_vp]),
    "rp_interval_table_host": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "rp_gather_profiles_host": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _int]),
    "rp_select_profiles_host": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, ctypes.POINTER(_i64)]),
    "rp_coverage_windows_host": (_int, [_vp, _vp, _i64, ctypes.c_int32, _vp, _vp, _vp, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64), _vp]),
    "rp_bam_split_host": (_int, [ctypes.c_char_p, _int, _vp, ctypes.c_int32, ctypes.POINTER(_vp)]),
    "rp_bam_view_host": (_int, [_vp, _vp]),
    "rp_bam_free": (None, [_vp]),
    "rp_format_double_repr": (_int, [ctypes.c_double, _vp]),
    "rp_format_int_list": (ctypes.c_size_t, [_vp, _i64, _vp]),
    "rp_format_wig_rows_host": (ctypes.c_size_t, [_vp, _vp, _i64, _vp]),
    "rp_wig_pack_host": (_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_int32, _vp, ctypes.c_int32, _vp, ctypes.POINTER(_i64)]),
    "rp_wig_render_host": (ctypes.c_size_t, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "rp_measurement_tag": (ctypes.c_int, [ctypes.c_int]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libribophase.so once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing. Build the HIP extension first: "
            "`make -C ribotricer_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "Both backends (hip and cpu) live in that library."
        )
    # PyTorch ships its own HIP / HSA runtime libraries; this library is linked against the system's (/opt/rocm).  Whichever
    # is loaded FIRST serves the whole process (same sonames), and a process that starts on the system's runtime and then
    # imports torch loses its device ("no ROCm-capable device is detected": seen with __graft_entry__.build() + smoke() in one
    # process).  So torch, where it is installed, goes first -- as it does in every other entry point of the package.
    try:
        import torch  # noqa: F401
    except ImportError:  # (a host without torch: the *_host entry points need neither)
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != RP_OK:
        raise RibophaseError(status, load().rp_last_error().decode("utf-8", "replace"))


def version() -> str:
    return load().rp_version().decode()


def device_count() -> int:
    n = _int(0)
    check(load().rp_device_count(ctypes.byref(n)))
    return n.value


def filter_defaults() -> FilterParams:
    fp = FilterParams()
    check(load().rp_filter_defaults(ctypes.byref(fp)))
    return fp


def tile_positions(n_orfs: int, total_nt: int) -> int:
    """Positions per tile the tile path uses for such an index (7 936, or 6 144 for short ORFs)."""
    out = ctypes.c_int32(0)
    check(load().rp_tile_positions(n_orfs, total_nt, ctypes.byref(out)))
    return out.value


def plan_bytes(n_orfs: int, total_nt: int) -> int:
    out = ctypes.c_size_t(0)
    check(load().rp_plan_bytes(n_orfs, total_nt, ctypes.byref(out)))
    return out.value


def gather_plan_bytes(n_orfs: int, n_intervals: int, total_nt: int) -> int:
    out = ctypes.c_size_t(0)
    check(load().rp_gather_plan_bytes(n_orfs, n_intervals, total_nt, ctypes.byref(out)))
    return out.value


def workspace_bytes(n_orfs: int, total_nt: int, algo: int) -> int:
    out = ctypes.c_size_t(0)
    check(load().rp_workspace_bytes(n_orfs, total_nt, algo, ctypes.byref(out)))
    return out.value


def tie_replay_host(values, offsets):
    """``(phase float64[n], valid int32[n])`` of every profile of a host CSR batch, with the
    reference's own float64 operations and this host's libm ``pow`` (``rp_tie_replay_host`` for
    integer profiles, ``rp_tie_replay_f64_host`` for float64 ones).  For the handful of exact
    frame ties the device cannot finish with the reference's bits; not a scoring path."""
    import numpy as np

    values = np.ascontiguousarray(values)
    is_float = values.dtype.kind == "f"
    values = np.ascontiguousarray(values, dtype=np.float64 if is_float else np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    phase = np.empty(n, np.float64)
    valid = np.empty(n, np.int32)
    fn = load().rp_tie_replay_f64_host if is_float else load().rp_tie_replay_host
    check(fn(values.ctypes.data if values.size else None, offsets.ctypes.data, n, phase.ctypes.data, valid.ctypes.data))
    return phase, valid


def usable_cores() -> int:
    """Cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota
    (a container may show 256 CPUs and own 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def phase_score_csr_host(counts, offsets, thresholds=None, n_threads: int = 0) -> dict:
    """``rp_phase_score_csr_host``: the per-ORF loop body on the HOST in the reference's own float64
    arithmetic (SURVEY.md 8(b) lists a host entry point beside the device one): the scoring step of the
    GPU-less backend (``backend.py``, RIBOTRICER_AMD_BACKEND=cpu) and of GPU-free cross-checks; never a
    fallback behind a failing device call.  Returns a dict of numpy arrays like ``PhaseScores.cpu_numpy()``."""
    import numpy as np

    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    out = {"phase": np.empty(n, np.float64), "valid": np.empty(n, np.int32), "read_count": np.empty(n, np.int64),
           "min_codon_cov": np.empty(n, np.int32), "flags": np.empty(n, np.uint8),
           "status": np.empty(n, np.uint8) if thresholds is not None else None}
    check(load().rp_phase_score_csr_host(
        counts.ctypes.data if counts.size else None, offsets.ctypes.data, n, out["phase"].ctypes.data, out["valid"].ctypes.data,
        out["read_count"].ctypes.data, out["min_codon_cov"].ctypes.data, out["flags"].ctypes.data,
        out["status"].ctypes.data if out["status"] is not None else None,
        ctypes.byref(thresholds) if thresholds is not None else None, int(n_threads)))
    return out
