import os
import sys

import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)
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


def _reload_lib_env():
    """libdr4sr_hip.so caches every DR4SR_* switch per process; dr4sr_reload_env() (include/dr4sr_hip_hooks.h) makes it read them again"""
    from dr4sr_amd import _lib
    if _lib._lib is not None:                      # only a library that is already loaded holds cached values
        _lib._lib.dr4sr_reload_env()


@pytest.fixture(autouse=True)
def _dr4sr_env_switches_follow_monkeypatch(monkeypatch):
    """Every test starts from the process environment as it is NOW (the previous test's monkeypatch undo has already happened), and
    every monkeypatch.setenv / delenv of a DR4SR_* name inside the test reaches the library at once — the cross-check switches of
    DESIGN.md 5a are tested in-process instead of in a fresh interpreter per switch."""
    _reload_lib_env()
    real_set, real_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        real_set(name, value, prepend)
        if name.startswith("DR4SR_"):
            _reload_lib_env()

    def delenv(name, raising=True):
        real_del(name, raising)
        if name.startswith("DR4SR_"):
            _reload_lib_env()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.setenv, monkeypatch.delenv = real_set, real_del
