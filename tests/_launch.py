"""Subprocess launches of the multi-process (transport) tests: `python -m torch.distributed.run` on a FREE port — a socket left behind by
an earlier crashed launch on the same box must not fail the next test for an unrelated reason — with the children's full stderr kept for
the assertion message (Next #1c)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def report(out, n=6000) -> str:
    """what an assertion on a launch prints: return code + the tails of both streams (the c10d / RCCL message is in stderr)"""
    return "rc=%s\n---- stdout tail ----\n%s\n---- stderr tail ----\n%s" % (out.returncode, out.stdout[-n:], out.stderr[-n:])


def torchrun(nproc, script, env=None, args=(), timeout=900):
    """run `script` (a path under the repo root, or absolute) as `nproc` ranks on this node; returns the CompletedProcess"""
    path = script if os.path.isabs(script) else os.path.join(ROOT, script)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), path] + [str(a) for a in args]
    full = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", TORCH_SHOW_CPP_STACKTRACES="1")
    full.update({k: str(v) for k, v in (env or {}).items()})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=full, cwd=ROOT)
