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


def class_fields():
    """The 24 coarse colour grids (4 per class of infer.py:22) found by tools/search_class_images.py."""
    return np.load(os.path.join(GOLDEN, "class_fields.npz"))["fields_u8"]


def parity_set_of(side):
    """The parity images of a side: 40 seeded images + the 24 class-covering field images (rows of parity_<side>.npz refer to it)."""
    from roomnet_amd.synth import parity_set
    return parity_set(side, class_fields())


@pytest.fixture(scope="session")
def parity_images():
    return parity_set_of(224)


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


# ---- parity evidence on file -------------------------------------------------------------------------------------------
# GPU parity tests record the numbers that bound the 16-bit shortcuts (per-stage relative errors, max |dlogit|, class-id
# disagreements and their margins) through `parity_record`; at the end of a session that recorded anything they are
# written as JSON to $RN_PARITY_REPORT (default gpurun_out/parity_report.json).  tools/parity_report.py runs the
# recording tests and puts the file under profiles/.
_PARITY = {}


def parity_record(section, key, value):
    _PARITY.setdefault(section, {})[key] = value


@pytest.fixture(scope="session")
def record():
    return parity_record


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    path = os.environ.get("RN_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    doc = {"what": "GPU parity evidence recorded by tests/ (HIP path through the C ABI vs the oracle / fp64 goldens / float32 HIP path)",
           "tolerances": {"logits_abs_16bit": 0.1, "id_margin_16bit": 0.2, "stage_rel_bf16": 0.0125, "stage_rel_f16": 0.006,
                          "note": "stage_rel = max |got - oracle| / absmax(oracle tensor); oracle parity with TensorFlow itself is unpinned (SURVEY 8c)"},
           "pytest_exitstatus": int(exitstatus)}
    doc.update(_PARITY)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)


def with_gammas(weights, bn2_frozen, bn6_frozen, seed):
    """The shipped checkpoint with OTHER frozen channels: every gamma of batch_normalization_2 (stage 2's BN) and _6 (stage 5's
    first BN) set to a trained-looking value, the listed channels to 1e-24 (their betas to a visible constant)."""
    rng = np.random.default_rng(seed)
    w = dict(weights)
    for name, frozen in (("batch_normalization_2", bn2_frozen), ("batch_normalization_6", bn6_frozen)):
        n = len(w[name + "/gamma"])
        g = rng.uniform(0.05, 0.4, n).astype(np.float32) * rng.choice([-1.0, 1.0], n).astype(np.float32)
        b = w[name + "/beta"].copy()
        for c in frozen:
            g[c] = np.float32(1e-24)
            b[c] = np.float32(rng.uniform(0.002, 0.02))
        w[name + "/gamma"] = g
        w[name + "/beta"] = b
        w[name + "/moving_variance"] = np.maximum(w[name + "/moving_variance"], np.float32(0.05))
    return w
