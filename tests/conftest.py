import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    d = {k: z[k] for k in z.files}
    if "cfg" in d:
        d["cfg"] = json.loads(bytes(d["cfg"]).decode())
    if "seed" in d:
        d["seed"] = int(d["seed"])
    return d


@pytest.fixture(scope="session")
def golden():
    return load_golden
