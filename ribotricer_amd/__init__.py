"""ribotricer_amd -- MI355X-native phase-score engine for ribotricer's detect-orfs path.

Scope: the per-ORF periodicity loop of
ribotricer/detect_orfs.py:274-324 + ribotricer/statistics.py:48-115, as
hand-written gfx950 HIP kernels behind a C ABI (include/ribophase.h), and the
host-side mirror of the reference interface for that path:

    ribotricer_amd.statistics.phasescore            <- ribotricer.statistics.phasescore
    ribotricer_amd.detect_orfs.export_orf_coverages <- ribotricer.detect_orfs.export_orf_coverages
    ribotricer_amd.engine.phase_score_csr           (batch seam the GPU path sits behind)

The HIP library must be built (``make -C ribotricer_amd/csrc``).  ``RIBOTRICER_AMD_BACKEND`` = ``hip`` | ``cpu`` |
``auto`` (default: hip when a HIP device is visible, else cpu -- decided from what is visible before any work, never
behind a failing device call; ``backend.py``).  The cpu backend runs export_orf_coverages / phasescore / detect_orfs
through the library's ``*_host`` entry points: the reference's own float64 arithmetic in C++.
"""



def _abi_version() -> str:
    """ONE version string: the C ABI's (``RP_VERSION_STRING`` in include/ribophase.h, what ``rp_version()``
    returns); read from the header so that importing the package does not need the built library."""
    import os
    import re

    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ribophase.h")
    with open(header) as fh:
        return re.search(r'#define\s+RP_VERSION_STRING\s+"([^"]+)"', fh.read()).group(1)


__version__ = _abi_version()
