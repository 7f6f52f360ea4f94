"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    gdir = ROOT / "tests" / "golden"

    def load(name):
        return np.load(gdir / name, allow_pickle=False)

    return load
