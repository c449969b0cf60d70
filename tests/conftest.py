import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (meta dict, {key: torch tensor}) from tests/golden/<name>.npz."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        meta = json.loads(bytes(z["meta"]).decode())
        arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrays


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden
