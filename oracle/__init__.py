"""CPU oracle for the phase-score path -- TEST INFRASTRUCTURE ONLY.

Nothing in ``ribotricer_amd/`` (the product path) may import this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.

Contents
--------
phasescore_literal   scalar restatement of ribotricer/statistics.py:48-115 that
                     issues the identical ``scipy.signal.coherence`` call
                     (parity pinned against the reference itself through the
                     golden fixtures in tests/golden/, see make_golden.py).
phase_oracle.c       plain-C restatement of the closed form over CSR arrays,
                     built by oracle/Makefile into oracle/_build/.
c_oracle             ctypes loader for the C restatement.
"""
