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
    config.addinivalue_line("markers", "slow: GPU tests with minutes of CPU oracle time; not part of -m gpu (run -m 'gpu or slow')")
    config.addinivalue_line("markers", "transport: launches child processes (torch.distributed.run / bench.py); collected last")


def pytest_collection_modifyitems(config, items):
    """Transport / subprocess tests run LAST: they launch `torch.distributed.run` children or whole bench.py
    processes, and under the driver's `pytest -x` one failure there used to hide every parity module that sorts behind it alphabetically
    (round 5: all of test_gpu_trained.py and test_gpu_xcd_order.py).  A test is a transport test when it is marked `transport` or its code
    names one of the launch helpers (tests/_launch.py torchrun, subprocess)."""
    def is_transport(item):
        if item.get_closest_marker("transport") is not None:
            return True
        fn = getattr(item, "function", None)
        names = set(getattr(getattr(fn, "__code__", None), "co_names", ()))
        return bool(names & {"torchrun", "subprocess"})
    first = [it for it in items if not is_transport(it)]
    last = [it for it in items if is_transport(it)]
    items[:] = first + last


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def at_scale(monkeypatch):
    """the at-scale launch forms (32-row tiles, length-class attention lists, scatter / owner jobs in k_wgrad) whatever the batch's
    expected token count: the library picks its regime from the plan's expected_tokens hint (csrc/kernels.h), which would send the
    smaller at-scale test batches down the latency forms"""
    monkeypatch.setenv("DR4SR_FORCE_SCALE", "1")


def _experiment_switches():
    """the DR4SR_* names the library reads through DR4SR_XENV (csrc/common.h): experiment / tuning switches, compiled out of the shipped build"""
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(ROOT, "dr4sr_amd", "csrc", "*.h*")):
        names |= set(re.findall(r'DR4SR_XENV\("(DR4SR_[A-Z0-9_]+)"\)', open(f).read()))
    return names


EXPERIMENT_SWITCHES = _experiment_switches()
SESSION_DETERMINISTIC = os.environ.get("DR4SR_DETERMINISTIC")          # set by whoever started pytest, not by a test


def _experiments_build() -> bool:
    from dr4sr_amd import _lib
    return bool(_lib.load().dr4sr_build_flags() & 1)


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
    # a model built with train.deterministic in an EARLIER test turned the process-wide mode on (dr4sr_amd/model/basemodel.py): a test starts
    # from the default mode unless it asks for the other one itself
    # (a DR4SR_DETERMINISTIC exported for the whole session — the forced-mode run of the suite, DESIGN §4f — stays what the session was given)
    bm = sys.modules.get("dr4sr_amd.model.basemodel")
    if bm is not None and bm.BaseModel._det_set_by_model:
        bm.BaseModel._det_set_by_model = False
        os.environ.pop("DR4SR_DETERMINISTIC", None)
    if SESSION_DETERMINISTIC is not None:                  # (also after a test that cleared the switch by hand: _lib.set_env(..., None))
        os.environ["DR4SR_DETERMINISTIC"] = SESSION_DETERMINISTIC
    _reload_lib_env()
    real_set, real_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        if name in EXPERIMENT_SWITCHES and not _experiments_build():
            pytest.skip("%s is an experiment / tuning switch: compiled out of the shipped library (csrc/common.h DR4SR_XENV); build "
                        "`make -C dr4sr_amd/csrc EXPERIMENTS=1` and run with DR4SR_LIB_PATH=.../libdr4sr_hip_exp.so" % name)
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


@pytest.fixture(scope="session", autouse=True)
def _memoise_oracle_gradients():
    """oracle.sasrec_oracle.grads_of(params, batch, ...) of a full-size batch takes seconds of CPU autograd, and the switch matrix
    (tests/test_gpu_r2_paths.py) asks for the same (parameters, batch) pairs once per switch: keep the results for the session, keyed by
    the CONTENT of every input tensor (blake2b of the bytes) — a changed parameter or batch is a different key."""
    import hashlib
    import torch
    from oracle import sasrec_oracle as O
    real = O.grads_of
    memo = {}

    def digest(d):
        h = hashlib.blake2b(digest_size=16)
        for k in sorted(d):
            t = d[k].detach().contiguous().cpu()
            h.update(k.encode())
            h.update(str(tuple(t.shape)).encode())
            h.update(t.numpy().tobytes())
        return h.hexdigest()

    def grads_of(params, batch, *args, **kw):
        if kw.get("masks") is not None or any(torch.is_tensor(a) or isinstance(a, dict) for a in args):
            return real(params, batch, *args, **kw)          # explicit dropout masks etc.: not memoised
        key = (digest(params), digest(batch), tuple(args), tuple(sorted(kw.items())))
        if key not in memo:
            if len(memo) > 64:
                memo.clear()
            memo[key] = real(params, batch, *args, **kw)
        loss, q, grads = memo[key]
        return loss, q, dict(grads)
    O.grads_of = grads_of
    from oracle import gru4rec_oracle as GO
    real_gru, memo_gru = GO.grads_of, {}

    def gru_grads_of(params, batch, *args, **kw):
        key = (digest(params), digest(batch), tuple(args), tuple(sorted(kw.items())))
        if key not in memo_gru:
            if len(memo_gru) > 32:
                memo_gru.clear()
            memo_gru[key] = real_gru(params, batch, *args, **kw)
        r = memo_gru[key]
        return r[0], r[1], dict(r[2])
    GO.grads_of = gru_grads_of
    yield
    O.grads_of, GO.grads_of = real, real_gru
