import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("tetra-nerf_amd.scenes")


@pytest.fixture(scope="session")
def oracle():
    from oracle import tn_oracle

    tn_oracle.build()
    return tn_oracle


@pytest.fixture(scope="session")
def bottle():
    """The reference's test mesh (tests/assets/bottle.ply) as committed golden arrays."""
    z = np.load(ROOT / "tests" / "golden" / "bottle_mesh.npz")
    return {"vertices": z["vertices"], "cells": z["cells"]}


@pytest.fixture(scope="session")
def tn():
    """The product package (loads libtetranerf_hip.so; fails loudly if it is missing)."""
    return importlib.import_module("tetra-nerf_amd")


@pytest.fixture(scope="session")
def device():
    import torch

    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    return torch.device("cuda:0")
