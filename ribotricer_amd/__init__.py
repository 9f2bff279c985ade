"""ribotricer_amd -- MI355X-native phase-score engine for ribotricer's detect-orfs path.

Scope: the per-ORF periodicity loop of
ribotricer/detect_orfs.py:274-324 + ribotricer/statistics.py:48-115, as
hand-written gfx950 HIP kernels behind a C ABI (include/ribophase.h), and the
host-side mirror of the reference interface for that path:

    ribotricer_amd.statistics.phasescore            <- ribotricer.statistics.phasescore
    ribotricer_amd.detect_orfs.export_orf_coverages <- ribotricer.detect_orfs.export_orf_coverages
    ribotricer_amd.engine.phase_score_csr           (batch seam the GPU path sits behind)

No CPU fallback: the HIP library must be built (``make -C ribotricer_amd/csrc``)
and a GPU must be present for any compute call.
"""

__version__ = "0.1.0"
