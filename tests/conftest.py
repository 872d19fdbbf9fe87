import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def at_scale(monkeypatch):
    """the at-scale launch forms (32-row tiles, length-class attention lists, scatter / owner jobs in k_wgrad) whatever the batch's
    expected token count: the library picks its regime from the plan's expected_tokens hint (csrc/kernels.h), which would send the
    smaller at-scale test batches down the latency forms"""
    monkeypatch.setenv("DR4SR_FORCE_SCALE", "1")
