"""GPU tests that take minutes of CPU oracle time: marked `slow` only (NOT `gpu`), so the driver's `pytest -m gpu` leaves them out and
`pytest -m "gpu or slow"` on an MI355X box runs everything; skipped where there is no GPU."""
import pytest
import torch

pytestmark = [pytest.mark.slow, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs the MI355X")]


def test_sasrec_strong_scaling_batch_size_131072_vs_chunked_oracle():
    """bench.py's strong[2] size — more than 65 536 sequences per grid, the weight-gradient split cap, several carried rounds per workgroup
    in the two-phase next-step prep — against the oracle summed over 16 chunks of 8 192 rows, and train_steps == repeated steps"""
    from test_gpu_r3_paths import _strong_scaling_batch_vs_chunked_oracle
    _strong_scaling_batch_vs_chunked_oracle(131072)
