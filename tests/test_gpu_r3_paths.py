"""Round-3 parity tests (VERDICT r2 "thin spots"):

  * the bench's own strong-scaling batch sizes — B = 32 768 and B = 131 072 toys-shaped rows per launch (two-phase next-step prep
    with several carried rounds per workgroup, weight-gradient split cap, more than 65 536 sequences per grid) — against the ORACLE
    run in chunks of 8 192 rows: the loss sum and every gradient are additive over sequences, so the chunks' un-normalised sums add
    up to the full batch's;
  * GRU4Rec at BASELINE configs[2] exactly: B = 256, N = 12 102 (amazon-beauty's item count);
  * MetaModel under data parallelism: one outer (hyper-gradient) step on 2 ranks equals the single-rank step.

Reference arithmetic: /root/reference model/sasrec.py:39-75 + model/basemodel.py:193-214 + model/loss_func.py:9-38 (SASRec step),
module/layers.py:117-136 (GRU), model/metamodel.py:123-166 + utils/utils.py:145-252 (outer loop)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sasrec_oracle as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


def test_sasrec_strong_scaling_batch_size_32768_vs_chunked_oracle():
    """B = 32 768 (bench.py's strong[1]); B = 131 072 (strong[2]: 16 chunks of CPU autograd, 150-300 s) is tests/test_slow_gpu.py,
    selected by `-m slow` (`-m "gpu or slow"` = everything)"""
    _strong_scaling_batch_vs_chunked_oracle(32768)


def _strong_scaling_batch_vs_chunked_oracle(B):
    """bench.py's `strong[1..2]` sizes (16-18 M seq/s claims).  Part 1: dr4sr_sasrec_fwd_bwd on a batch selected on the device from a
    permutation of a U = B + 1000 row dataset — loss and EVERY gradient against the oracle's autograd summed over B / 8192 chunks.
    Part 2: dr4sr_sasrec_train_steps (the optimizer launch prepares the next step in two phases spread over its grid) against
    repeated fwd_bwd + adam_step with the stand-alone prep launch, two steps, the second batch wrapping the permutation."""
    from test_gpu_parity import _random_params
    from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
    from dr4sr_amd.engine import SasrecEngine
    dev = torch.device("cuda", 0)
    L, N, U, CH = 50, TOYS_N_ITEMS, B + 1000, 8192
    rows = make_rows(n_rows=U, n_items=N, seed=31)
    data_h = {k: torch.from_numpy(rows[k]) for k in ("in_item_id", "item_id", "seqlen")}
    data = {k: v.to(dev) for k, v in data_h.items()}
    perm_h = torch.from_numpy(np.random.default_rng(5).permutation(U))
    perm = perm_h.to(dev)
    negs_h = torch.randint(1, N, (B, L), generator=torch.Generator().manual_seed(32))          # by batch slot
    params = _random_params(N, 64, 128, 2, seed=6)

    def make(lr):
        eng = SasrecEngine(N, L, 64, 2, 128, 2, 1e-12, 0.0, B, dev, seed=9, lr=lr)
        eng.load_named(params)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        log = torch.zeros(8, dtype=torch.float32, device=dev)
        plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=torch.zeros(B, dtype=torch.int64, device=dev),
                             neg_item=negs_h.to(dev).view(-1), sample_neg=False, perm_sel=(perm, B, 0, counter), loss_log=log)
        return eng, plan, counter, log

    # ---- part 1: one fwd_bwd at this size vs the chunked oracle
    eng, plan, counter, log = make(1e-3)
    eng.fwd_bwd(plan)
    torch.cuda.synchronize()
    g_dev = {k: v.clone() for k, v in eng.normalized_grads().items()}
    loss, n = eng.loss_and_count()
    sel = perm_h[:B]
    acc, loss_sum, n_sum = None, 0.0, 0
    for c in range(0, B, CH):
        r = sel[c:c + CH]
        b = {"in_item_id": data_h["in_item_id"][r], "item_id": data_h["item_id"][r], "seqlen": data_h["seqlen"][r],
             "neg_item": negs_h[c:c + CH].unsqueeze(-1)}
        nc = int((b["item_id"] != 0).sum())
        lo, _, go = O.grads_of(params, b, 2, 2, 1e-12)
        loss_sum += float(lo) * nc
        n_sum += nc
        if acc is None:
            acc = {k: v.double() * nc for k, v in go.items()}
        else:
            for k, v in go.items():
                acc[k] += v.double() * nc
    assert n == n_sum, (n, n_sum)
    assert abs(loss - loss_sum / n_sum) < 2e-5, (loss, loss_sum / n_sum)
    worst = 0.0
    for k, gv in g_dev.items():
        e = relerr(gv, acc[k] / n_sum)
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    print("B=%d chunked-oracle parity: %d valid targets, worst grad relerr %.2e" % (B, n_sum, worst))

    # ---- part 2: train_steps (two-phase next-step prep inside the optimizer launch) == fwd_bwd + adam_step with the prep launch
    eng.adam_step(plan)
    eng.fwd_bwd(plan)
    eng.adam_step(plan)
    torch.cuda.synchronize()
    p_ref, log_ref = eng.params.clone(), log.clone()
    assert int(counter) == 2
    del eng, plan
    torch.cuda.empty_cache()
    eng2, plan2, counter2, log2 = make(1e-3)
    eng2.train_steps(plan2, 2)
    torch.cuda.synchronize()
    assert int(counter2) == 2 and int(eng2.state[0]) == 2
    assert torch.allclose(log_ref[:2], log2[:2], rtol=1e-5, atol=1e-6), (log_ref[:2], log2[:2])
    assert float(log2[0]) > 0 and abs(float(log2[0]) - loss) < 2e-5
    assert float((p_ref - eng2.params).abs().max()) < 2e-4          # fp32 atomics order of the weight gradients, amplified by Adam


def test_gru4rec_baseline_config2_exact_size_vs_oracle():
    """BASELINE configs[2] exactly: GRU4Rec, B = 256, amazon-beauty's N = 12 102 items, hidden 256, 2 layers — loss and every gradient
    against oracle/gru4rec_oracle.py (cooperative recurrence, 16 slices per group)"""
    from test_gpu_r2_paths import _gru_vs_oracle
    eng, worst = _gru_vs_oracle(256, 256, 2, N=12102)
    assert eng.uses_cooperative(256)
    assert eng.uses_wavefront(256) == (os.environ.get("DR4SR_GRU_NOWAVE") is None)
    print("GRU4Rec B=256 N=12102 worst grad relerr %.2e" % worst)


def test_gru4rec_partial_batch_in_the_workspace_of_a_larger_one():
    """An engine built for batches of 300 (8 slices per group, no wavefront) must run a batch of 256 (16 slices per group + the
    wavefront's per-step slots: a LARGER exchange area) in the same workspace — the last batch of an epoch; the reservation is an upper
    bound over every batch size up to the engine's (csrc/gru_coop.hip gru_xch_words)"""
    from dr4sr_amd.gru_engine import GruEngine, gru_param_names, gru_param_shapes
    from test_gpu_r2_paths import _gru_batch, relerr
    N, L, H, NL = 2000, 50, 256, 2
    gen = torch.Generator().manual_seed(3)
    params = {n: 0.08 * torch.randn(s_, generator=gen) for n, s_ in zip(gru_param_names(NL), gru_param_shapes(N, 64, H, NL))}
    params["item_embedding.weight"][0] = 0
    out = {}
    for maxb in (300, 256):
        eng = GruEngine(N, L, 64, H, NL, 0.0, maxb, "cuda", seed=5)
        eng.load_named(params)
        for B in ((300, 256) if maxb == 300 else (256,)):
            b, _ = _gru_batch(B, N, L, 11)
            dev = eng.device
            plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                                 neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
            eng.fwd_bwd(plan)
            eng.check_device_error()
            out[(maxb, B)] = {k: v.detach().cpu().clone() for k, v in eng.normalized_grads().items()}
    assert eng.uses_wavefront(256) and not eng.uses_wavefront(300)
    for k, v in out[(256, 256)].items():
        assert relerr(out[(300, 256)][k], v) < 2e-5, k


@pytest.mark.parametrize("env", [{"DR4SR_GRU_NOWAVE": "1"}, {"DR4SR_GRU_WAVE_BWD": "1"}, {"DR4SR_GRU_NOWAVE": "1", "DR4SR_GRU_BWD_F32": "1", "DR4SR_GRU_FWD_F32": "1"},
                                 {"DR4SR_GRU_NOFUSE_GLUE": "1"}, {"DR4SR_GRU_WAVE_ORDER": "0"}, {"DR4SR_WGRAD_F32": "1", "DR4SR_GRU_NO_SPLITK": "1"}],
                         ids=["one-launch-per-layer", "backward-wavefront", "round-2-kernels", "separate-glue-launches", "round-3-role-order",
                              "fp32-weight-gradients"])
def test_gru4rec_wavefront_switches_vs_oracle(env, monkeypatch):
    """Two-layer plans run both forward recurrences in one launch by default (layer wavefront, csrc/gru_coop.hip); DR4SR_GRU_NOWAVE = the
    one-launch-per-layer form, DR4SR_GRU_WAVE_BWD = the (opt-in, not faster) one-launch backward, DR4SR_GRU_BWD_F32 = the fp32-MFMA BPTT with
    W_hh in LDS instead of the bf16x3 one with W_hh in registers (k_gru_bwd_coop_bf; DR4SR_GRU_FWD_F32 likewise for the single-layer forward).
    The oracle tests that reach them — BASELINE configs[2] exactly, odd batch sizes, chunks of 256 with a ragged last chunk, the second bank
    of cooperative groups — re-run in this process with the switch set (dr4sr_reload_env through conftest's monkeypatch hook).
    Round 4: DR4SR_GRU_NOFUSE_GLUE = the separate embed / projection / scorer / scatter launches instead of k_gru_embed_gi, k_gru_mid,
    k_gru_dx_embed; DR4SR_GRU_WAVE_ORDER=0 = round 3's block -> role order of the forward wavefront; DR4SR_WGRAD_F32 = k_wgrad64 on the fp32
    matrix cores instead of the bf16x3 split, DR4SR_GRU_NO_SPLITK = the one-chain K = 3H data-gradient GEMM."""
    import test_gpu_gru as G
    import test_gpu_r2_paths as R2
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if "DR4SR_GRU_NOFUSE_GLUE" in env:                        # round 4: the fused glue launches are the default; their dropout masks too
        G.test_gru4rec_full_size_vs_oracle(False)
    test_gru4rec_baseline_config2_exact_size_vs_oracle()
    for B in (1, 17, 100):
        G.test_gru4rec_odd_batch_sizes_vs_oracle(B)
    R2.test_gru4rec_chunked_cooperative_recurrence_vs_oracle(400, 256, True)
    R2.test_gru4rec_cooperative_second_bank_vs_oracle(130)


def test_metamodel_outer_step_two_ranks_equal_single_rank():
    """SURVEY §8(e) last row: the outer loop's all-reduces (d L_val / dW, six Hessian-vector probes, two mixed-derivative probes of both
    flat buffers) leave every rank with the single-rank hyper-gradient and meta-module step (tools/dp_meta_check.py: 2 ranks sharing
    cuda:0 over the gloo transport of dr4sr_amd/parallel.py, explicit Gumbel noise, dropout 0)"""
    from _launch import report, torchrun
    out = torchrun(2, "tools/dp_meta_check.py", {"DR4SR_DP_BACKEND": "gloo"}, timeout=600)
    assert out.returncode == 0, report(out)
    line = [l for l in out.stdout.splitlines() if l.startswith("DP_META")]
    assert line and "replicas identical: True" in line[0], out.stdout[-2000:]
    print(line[0])


def test_rccl_allreduce_inside_k_step_graph_replayed_120_times():
    """the opt-in data-parallel form (RCCL all-reduce captured inside the k-step graph, dr4sr_amd/model/basemodel.py:_step_graph and
    bench.py) with the one RCCL rank a 1-GPU box has: 30 replays of a 4-step graph = 120 steps, loss log and parameters against the
    un-captured single-GPU loop (tools/dp_graph_check.py)"""
    from _launch import report, torchrun
    out = torchrun(1, "tools/dp_graph_check.py", timeout=600)
    assert out.returncode == 0 and "DP_GRAPH_OK" in out.stdout, report(out)
    print([l for l in out.stdout.splitlines() if l.startswith("DP_GRAPH ")][0])
