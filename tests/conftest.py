"""Shared pytest configuration.

Markers
-------
gpu   needs a real MI355X (run with ``-m gpu`` on the GPU box through the C ABI).
Everything else runs on CPU: oracle vs. golden fixtures, host logic, ABI symbols,
world_size-2 gloo sharding.
"""

import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (HIP path through the C ABI)")


@pytest.fixture(autouse=True)
def _gpu_tests_insist_on_the_hip_backend(request, monkeypatch):
    """A test marked gpu must run the HIP path or fail: never the host backend that ``auto`` would pick on a box whose
    GPU is not visible."""
    if request.node.get_closest_marker("gpu") is not None:
        monkeypatch.setenv("RIBOTRICER_AMD_BACKEND", "hip")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def g1():
    return load_json("g1_known_answers.json")["vectors"]


@pytest.fixture(scope="session")
def g2():
    return load_npz("g2_poisson.npz")


@pytest.fixture(scope="session")
def g3():
    return load_npz("g3_adversarial.npz")


@pytest.fixture(scope="session")
def g4():
    return load_npz("g4_long.npz")


@pytest.fixture(scope="session")
def g5():
    return load_json("g5_float.json")["vectors"]


@pytest.fixture(scope="session")
def g8():
    return load_npz("g8_bigties.npz")


@pytest.fixture(scope="session")
def g10():
    return load_npz("g10_bigcounts.npz")


@pytest.fixture(scope="session")
def g8f():
    return load_json("g8_float_ties.json")["vectors"]


def split_csr(counts, offsets):
    return [counts[offsets[i] : offsets[i + 1]] for i in range(len(offsets) - 1)]
