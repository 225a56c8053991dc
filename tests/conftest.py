import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODEL_PREFIX = os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights():
    from roomnet_amd.tf_bundle import BundleReader
    return BundleReader(MODEL_PREFIX).load_all()


@pytest.fixture(scope="session")
def parity_images():
    from roomnet_amd.synth import parity_batch
    return parity_batch(224, seed=1)


@pytest.fixture(scope="session")
def golden_parity():
    return np.load(os.path.join(GOLDEN, "parity_224.npz"))


@pytest.fixture(scope="session")
def golden_taps():
    return np.load(os.path.join(GOLDEN, "taps_224.npz"))


def sample_positions(size, k=16):
    """Same deterministic positions tools/make_golden.py sampled."""
    rng = np.random.default_rng(size)
    return np.sort(rng.integers(0, size, k))
