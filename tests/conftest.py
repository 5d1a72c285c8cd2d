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


def pytest_sessionstart(session):
    """The native pieces are built in-tree by __graft_entry__.build(); build them here if a fresh checkout has not yet
    (hipcc cross-compiles gfx950 without a GPU, ~1 min; gcc for the oracle)."""
    from vocoder_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        _lib.build()
    from oracle import oracle as orc
    orc.build()


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
