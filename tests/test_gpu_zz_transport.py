"""Round-6 transport tests (VERDICT r5 "Next round" 1 + 2).  Collected LAST (tests/conftest.py: every test that launches child processes):

  * the library's own RCCL communicator through the C ABI, no torch.distributed in the process (tools/comm_check.py): collectives,
    error codes, the asynchronous form as a parallel branch of a captured graph replayed 200 times;
  * the data-parallel k-step graph with its collective(s) captured, FIVE fresh processes in a row (tools/dp_graph_check.py) — the form
    that aborted once on the round-5 driver box under ProcessGroupNCCL; `tools/dp_graph_loop.sh` runs the long loop whose log is kept
    under profiles/;
  * bench.py's one JSON line survives a SIGABRT injected into a later leg (arm_crash_line / dr4sr_crash_line_set): single process and
    under torch.distributed.run with the in-graph collective leg as the victim.

The reference has no distributed path (/root/reference/utils/callbacks.py:130 is its TODO); the contract is SURVEY.md section 8(b) last
table row (`allreduce_flat(buf)` (RCCL)) + section 8(e)."""
import json
import os
import subprocess
import sys

import pytest

from _launch import ROOT, report, torchrun

pytestmark = [pytest.mark.gpu, pytest.mark.transport]


def test_native_rccl_communicator_through_the_c_abi():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "comm_check.py")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert out.returncode == 0 and "COMM_CHECK_OK" in out.stdout, report(out)
    print([l for l in out.stdout.splitlines() if l.startswith("COMM_CHECK ")][0])


@pytest.mark.parametrize("B,buckets", [(256, 1), (8192, 2)])
def test_rccl_in_graph_step_five_fresh_processes_in_a_row(B, buckets):
    """each launch: rendezvous, communicator, capture of a 4-step DP graph with its all-reduce(s), 30 replays, parity against the
    un-captured single-GPU loop, teardown — the whole life cycle, five times"""
    env = {"DP_GRAPH_B": B, "DP_GRAPH_EXPECT_BUCKETS": buckets, "DR4SR_DP_BUCKETS": buckets, "DP_GRAPH_REPLAYS": 30 if B == 256 else 8}
    for i in range(5):
        out = torchrun(1, "tools/dp_graph_check.py", env, timeout=600)
        assert out.returncode == 0 and "DP_GRAPH_OK" in out.stdout, "launch %d\n" % i + report(out)
    print([l for l in out.stdout.splitlines() if l.startswith("DP_GRAPH ")][0])


def _one_line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, report(out)
    return json.loads(lines[0])


def test_bench_line_survives_an_abort_in_a_later_leg():
    """single process: SIGABRT at the start of the first extra leg — the headline measurement is complete, the line must appear, marked"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--repeats", "3", "--no-cpu-baseline",
                          "--no-strong"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, DR4SR_BENCH_INJECT_ABORT="throughput_mode"), cwd=ROOT)
    j = _one_line(out)
    assert out.returncode == 0, report(out)
    assert j["aborted_during"] == "throughput_mode" and j["value"] > 0 and j["roofline"]["frac"] > 0 and "throughput_mode" not in j
    assert j["metric"].startswith("training sequences/sec") and j["n_gpus"] == 1


def test_bench_line_survives_an_abort_in_the_in_graph_leg():
    """under torch.distributed.run, the library's RCCL communicator with its one rank: the host-form line is complete when the in-graph
    leg starts; an abort there (what a c10d thread did to the round-5 suite) still yields ONE line with collective_forms.in_graph_error"""
    out = torchrun(1, "bench.py", {"DR4SR_BENCH_FORCE_DP": "1", "DR4SR_BENCH_INJECT_ABORT": "in_graph"},
                   args=["--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "3", "--no-throughput-mode", "--no-strong", "--no-cpu-baseline"],
                   timeout=600)
    j = _one_line(out)
    assert j["aborted_during"] == "in_graph" and j["value"] > 0
    cf = j["collective_forms"]
    assert cf["in_graph"] is None and "signal" in cf["in_graph_error"] and cf["host"] == j["ms_per_step"]
    assert "rccl all-reduce launched by the host" in j["config"]["collective"]


def test_bench_says_so_when_a_multi_rank_run_dies_before_its_first_measurement():
    """a data-parallel run killed before ANY measurement exists (a wedged first collective + the watchdog, a launcher's SIGTERM): rank 0 leaves one
    parseable line with value null and the reason, and the process exits non-zero — not silence"""
    out = torchrun(1, "bench.py", {"DR4SR_BENCH_FORCE_DP": "1", "DR4SR_BENCH_INJECT_ABORT": "headline"},
                   args=["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-throughput-mode", "--no-strong", "--no-cpu-baseline"], timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode != 0 and len(lines) == 1, report(out)
    j = json.loads(lines[0])
    assert j["value"] is None and "no measurement completed" in j["aborted_during"] and j["n_gpus"] == 1


def test_bench_falls_back_to_the_staged_data_plane_when_rccl_refuses():
    """two ranks on ONE GPU with the RCCL data plane asked for explicitly: RCCL refuses a second rank on a device, every rank sees the failure
    over the control plane (parallel.init_distributed(allow_fallback=True): all_ok), all of them drop to the host-staged gloo data plane and
    rank 0 still prints ONE line — flagged with `transport_fallback`, never silently.  (fit() / run.py do NOT fall back: they raise.)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--repeats", "2",
                          "--no-throughput-mode", "--no-strong"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, DR4SR_BENCH_SHARE_GPU="1", DR4SR_DP_BACKEND="rccl"), cwd=ROOT)
    j = _one_line(out)
    assert out.returncode == 0, report(out)
    assert j["n_gpus"] == 2 and "could not be created" in j["transport_fallback"] and "gloo" in j["transport_fallback"]
    assert j["value"] > 0 and j["collective_forms"]["in_graph"] is None


def test_a_wedged_communicator_bootstrap_costs_the_timeout_not_the_run(monkeypatch):
    """parallel.init_distributed(allow_fallback=True, init_timeout=...) — what bench.py calls: the RCCL bootstrap runs in a helper thread; when it
    does not return in time (here: the entry point replaced by one that sleeps) the rank counts as failed, the ranks agree over the control plane
    and take the staged gloo data plane with the reason recorded, instead of hanging the launcher"""
    import time
    import torch
    from _launch import free_port
    from dr4sr_amd import _lib, parallel
    lib = _lib.load()
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(free_port())), ("DR4SR_BENCH_FORCE_DP", "1")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("DR4SR_DP_BACKEND", raising=False)
    real = lib.dr4sr_comm_init_rank
    calls = []

    def stalls(*a):
        calls.append(1)
        time.sleep(3.0)
        return -1
    lib.dr4sr_comm_init_rank = stalls
    try:
        t0 = time.time()
        assert parallel.init_distributed(torch.device("cuda", 0), allow_fallback=True, init_timeout=0.5)
        took = time.time() - t0
        assert calls and took < 2.5 and not parallel.can_capture()
        assert "did not return within" in parallel.FALLBACK_REASON and "gloo" in parallel.FALLBACK_REASON
        g = torch.ones(1000, device="cuda")
        parallel.allreduce_flat(g)                               # the staged plane carries the collective (one rank: identity)
        assert bool((g == 1).all())
    finally:
        lib.dr4sr_comm_init_rank = real
        parallel.shutdown()
        parallel.FALLBACK_REASON = None
        time.sleep(3.0)                                        # let the helper thread's sleep end before the next test
