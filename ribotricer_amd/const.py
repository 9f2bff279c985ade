"""Thresholds of the detect-orfs status predicate.

Same names and values as ribotricer/const.py:20-39 of the reference so that callers
of ``export_orf_coverages`` see identical defaults.
"""

CUTOFF = 0.428571428571  # const.py:20
MINIMUM_VALID_CODONS = 5  # const.py:27
MINIMUM_READS_PER_CODON = 0  # const.py:32
MINIMUM_VALID_CODONS_RATIO = 0  # const.py:35
MINIMUM_DENSITY_OVER_ORF = 0.0  # const.py:39
TYPICAL_OFFSET = 12  # const.py:23
META_MIN_READS = 100000  # const.py:42
