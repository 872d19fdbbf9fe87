"""GPU parity of the FMLP path (dr4sr_fmlp_* through the C ABI) vs golden vectors from the reference and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fmlp_oracle as FO  # noqa: E402

REL = 3e-4


def relerr(a, b):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "fmlp_d64.npz"))
    g = {k: z[k] for k in z.files}
    params = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    batch = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    return g, params, batch


def engine(g, params, B, p=0.0, **kw):
    from dr4sr_amd.fmlp_engine import FmlpEngine
    eng = FmlpEngine(int(g["meta.num_items"]), 50, 64, 256, int(g["meta.layer_num"]), 1e-12, p, B, "cuda", **kw)
    eng.load_named(params)
    return eng


def test_fmlp_encode_fwd_bwd_adam_vs_golden(golden_dir):
    g, params, b = load(golden_dir)
    B = b["in_item_id"].shape[0]
    eng = engine(g, params, B, lr=float(g["meta.lr"]))
    dev = eng.device
    idx, tgt = b["in_item_id"].to(dev), b["item_id"].to(dev)
    neg = b["neg_item"].view(-1).contiguous().to(dev)
    plan = eng.make_plan(idx, tgt, neg_item=neg, sample_neg=False)
    q = eng.encode(plan, False)
    assert relerr(q, g["out.query"]) < REL
    # eval rows (initial weights): query at the last position
    ev = eng.encode(eng.make_plan(torch.from_numpy(g["eval.in_item_id"]).to(dev), None), False)
    assert relerr(ev, g["eval.query_last"]) < REL
    eng.fwd_bwd(plan)
    loss, n = eng.loss_and_count()
    assert n == B and abs(loss - float(g["out.loss"])) < 1e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, g["grad." + k]) < REL, k
    eng.adam_step()
    for k, v in eng.views.items():
        well = np.abs(g["grad." + k]) > 1e-5
        d = v.cpu().numpy() - g["adam1." + k]
        assert np.abs(d[well]).max(initial=0) < 1e-5 and np.abs(d).max() < 1.1e-3, k
    eng.fwd_bwd(plan)
    assert abs(eng.loss_and_count()[0] - float(g["out.loss_step2"])) < 3e-5


def test_fmlp_dropout_masks_match_oracle(golden_dir):
    g, params, b = load(golden_dir)
    B, L = b["in_item_id"].shape
    p = 0.5
    eng = engine(g, params, B, p=p, seed=77)
    dev = eng.device
    neg = b["neg_item"].view(-1).contiguous().to(dev)
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), neg_item=neg, sample_neg=False)
    eng.fwd_bwd(plan)
    step = int(eng.state[3])
    masks = {FO.SITE_EMB: eng.dropout_mask(B * L * 64, 0, step).view(B, L, 64).cpu()}
    for l in range(2):
        masks[FO.site(FO.SITE_FILT, l)] = eng.dropout_mask(B * L * 64, 1 + 2 * l, step).view(B, L, 64).cpu()
        masks[FO.site(FO.SITE_FFN, l)] = eng.dropout_mask(B * L * 64, 2 + 2 * l, step).view(B, L, 64).cpu()
    loss_o, _, grads_o = FO.grads_of(params, b, 2, masks=masks, pdrop=p)
    loss, n = eng.loss_and_count()
    assert abs(loss - float(loss_o)) < 3e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, grads_o[k]) < REL, k


def test_fmlp_full_size_and_training_decreases_loss():
    from dr4sr_amd.data.synthetic import TOYS_N_ITEMS
    from dr4sr_amd.fmlp_engine import FmlpEngine, fmlp_param_names, fmlp_param_shapes
    B, L, N = 256, 50, TOYS_N_ITEMS
    gen = torch.Generator().manual_seed(0)
    idx = torch.zeros(B, L, dtype=torch.long)
    for i in range(B):
        n = int(torch.randint(1, L + 1, (1,), generator=gen))
        idx[i, L - n:] = torch.randint(1, N, (n,), generator=gen)          # left-padded prefixes
    tgt = torch.randint(1, N, (B,), generator=gen)
    tgt[3] = 0                                                             # one masked row
    neg = torch.randint(1, N, (B, 1), generator=gen)
    params = {}
    for nme, shp in zip(fmlp_param_names(2), fmlp_param_shapes(N, L, 64, 256, 2)):
        params[nme] = (1.0 if nme.endswith("LayerNorm.weight") else 0.0) + 0.05 * torch.randn(shp, generator=gen)
    params["item_embedding.weight"][0] = 0
    eng = FmlpEngine(N, L, 64, 256, 2, 1e-12, 0.0, B, "cuda")
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(idx.to(dev), tgt.to(dev), neg_item=neg.view(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    batch = {"in_item_id": idx, "item_id": tgt, "neg_item": neg}
    loss_o, _, grads_o = FO.grads_of(params, batch, 2)
    loss, n = eng.loss_and_count()
    assert n == B - 1 and abs(loss - float(loss_o)) < 3e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, grads_o[k]) < REL, k
    first = loss
    for _ in range(30):
        eng.train_step(plan)
    assert eng.loss_and_count()[0] < first - 0.05 and int(eng.state[0]) == 30
    # in-kernel negatives stay in range
    plan2 = eng.make_plan(idx.to(dev), tgt.to(dev))
    eng.fwd_bwd(plan2)
    assert int(eng.neg_scratch[:B].min()) >= 1 and int(eng.neg_scratch[:B].max()) < N


@pytest.mark.parametrize("B", [1, 3, 33, 100])
def test_fmlp_odd_batch_sizes_vs_oracle(B):
    """batch sizes that do not fill the 32-row token tiles / the 256 partial-sum blocks of the filter backward"""
    _fmlp_case(B, 2, 50)


@pytest.mark.parametrize("NL,L", [(1, 50), (3, 50), (2, 20), (4, 8), (2, 2)])
def test_fmlp_every_accepted_shape_vs_oracle(NL, L):
    """1-4 filter layers, even sequence lengths from 2 to 50 (the plan check of csrc/fmlp.hip): loss and every gradient vs the oracle"""
    _fmlp_case(29, NL, L)


@pytest.mark.parametrize("L", [49, 64])
def test_fmlp_unsupported_lengths_are_refused(L):
    from dr4sr_amd.fmlp_engine import FmlpEngine
    from dr4sr_amd import _lib
    with pytest.raises(_lib.Dr4srError):
        FmlpEngine(150, L, 64, 256, 2, 1e-12, 0.0, 8, "cuda")


def _fmlp_case(B, NL, L):
    from dr4sr_amd.fmlp_engine import FmlpEngine, fmlp_param_names, fmlp_param_shapes
    N = 150
    gen = torch.Generator().manual_seed(100 + B)
    idx = torch.zeros(B, L, dtype=torch.long)
    for i in range(B):
        n = int(torch.randint(1, L + 1, (1,), generator=gen))
        idx[i, L - n:] = torch.randint(1, N, (n,), generator=gen)
    tgt = torch.randint(1, N, (B,), generator=gen)
    neg = torch.randint(1, N, (B, 1), generator=gen)
    params = {}
    for nme, shp in zip(fmlp_param_names(NL), fmlp_param_shapes(N, L, 64, 256, NL)):
        params[nme] = (1.0 if nme.endswith("LayerNorm.weight") else 0.0) + 0.05 * torch.randn(shp, generator=gen)
    params["item_embedding.weight"][0] = 0
    eng = FmlpEngine(N, L, 64, 256, NL, 1e-12, 0.0, B, "cuda")
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(idx.to(dev), tgt.to(dev), neg_item=neg.view(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    loss_o, _, grads_o = FO.grads_of(params, {"in_item_id": idx, "item_id": tgt, "neg_item": neg}, NL)
    loss, n = eng.loss_and_count()
    assert n == B and abs(loss - float(loss_o)) < 3e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, grads_o[k]) < REL, k


@pytest.mark.parametrize("env", [{"DR4SR_WGRAD_F32": "1"}, {"DR4SR_FMLP_WGRAD_WIDE": "1"}], ids=["fp32-weight-gradients", "whole-jobs-bf16x3"])
def test_fmlp_weight_gradient_switches_vs_oracle(env, golden_dir, monkeypatch):
    """round 4: the FFN weight gradients run as 64 x 64 blocks on the bf16 matrix cores (3-term split, k_fmlp_wgrad_bf64) by default;
    DR4SR_WGRAD_F32 = the fp32-MFMA kernel, DR4SR_FMLP_WGRAD_WIDE = the split on the two whole jobs per layer.  The golden and
    oracle tests re-run in this process with the switch set."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    test_fmlp_encode_fwd_bwd_adam_vs_golden(golden_dir)
    for B in (1, 37):
        test_fmlp_odd_batch_sizes_vs_oracle(B)


def test_fmlp_bf16x3_intermediate_kernels_match_fp32_mfma(monkeypatch):
    """round 5: the Intermediate-block GEMMs as a 3-term bf16 split from fragment-major weight images (csrc/common.h tile_mma_xwT_bf3; images
    written by extra blocks of k_fmlp_prep) against the fp32-MFMA kernels of the same step (DR4SR_TILE_F32=1): loss and every gradient"""
    from dr4sr_amd.data.synthetic import make_rows
    from dr4sr_amd.fmlp_engine import FmlpEngine
    dev = torch.device("cuda", 0)
    B, L, N = 256, 50, 997
    rows = make_rows(n_rows=B, n_items=N, seed=23)
    hist, sl = torch.from_numpy(rows["in_item_id"]).to(dev), torch.from_numpy(rows["seqlen"]).to(dev)
    ar = torch.arange(L, device=dev).view(1, -1)
    shift = (L - sl).view(-1, 1)
    ids = torch.where(ar >= shift, hist.gather(1, (ar - shift) % L), torch.zeros_like(hist)).contiguous()      # left-padded prefixes
    tgt = torch.from_numpy(rows["item_id"]).to(dev).gather(1, (sl - 1).clamp(min=0).view(-1, 1)).squeeze(1).contiguous()
    neg = torch.randint(1, N, (B,), generator=torch.Generator().manual_seed(2)).to(dev)
    out = {}
    for f32 in (False, True):
        if f32:
            monkeypatch.setenv("DR4SR_TILE_F32", "1")
        eng = FmlpEngine(N, L, 64, 256, 2, 1e-12, 0.0, B, dev, seed=3, lr=1e-3)
        g = torch.Generator().manual_seed(5)
        for k, v in eng.views.items():
            v.copy_(torch.ones(v.shape) if k.endswith("LayerNorm.weight") else (torch.zeros(v.shape) if k.endswith("bias") else 0.05 * torch.randn(v.shape, generator=g)))
        eng.views["item_embedding.weight"][0] = 0
        plan = eng.make_plan(ids, tgt, neg_item=neg, sample_neg=False)
        eng.fwd_bwd(plan)
        torch.cuda.synchronize()
        out[f32] = (eng.loss_and_count(), eng.grads.clone())
    (l0, n0), g0 = out[False]
    (l1, n1), g1 = out[True]
    assert n0 == n1 and abs(l0 - l1) < 1e-5
    assert float((g0 - g1).abs().max()) <= 5e-5 * float(g1.abs().max())


def test_fmlp_launch_kernel_hook_replays_the_steps_launches(golden_dir):
    """the measurement hook of bench.py's FMLP roofline (include/dr4sr_hip_hooks.h: dr4sr_fmlp_launch_kernel): every launch it re-enqueues
    on the state a fwd_bwd left reproduces what that step wrote (dropout off: the launches are pure functions of the workspace); the
    weight-gradient launch ACCUMULATES (the step zeroes the gradients in its first launch), so one replay doubles the GEMM gradients"""
    import ctypes as C
    from dr4sr_amd import _lib
    g, params, b = load(golden_dir)
    B = b["in_item_id"].shape[0]
    eng = engine(g, params, B)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), neg_item=b["neg_item"].view(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    torch.cuda.synchronize()
    ws0 = eng.workspace.clone()
    n0 = eng.loss_and_count()[1]
    g0 = {k: v.clone() for k, v in eng.normalized_grads().items()}
    lib = eng.lib
    for name in ("filter_fwd", "ffn_fwd", "ffn_bwd", "filter_bwd"):
        for layer in range(eng.n_layer):
            _lib.check(lib.dr4sr_fmlp_launch_kernel(C.byref(plan), _lib.FMLP_KERNEL_IDS[name], layer, _lib.cur_stream()), name)
            torch.cuda.synchronize()
            assert torch.equal(eng.workspace, ws0), (name, layer)
    _lib.check(lib.dr4sr_fmlp_launch_kernel(C.byref(plan), _lib.FMLP_KERNEL_IDS["wgrad"], 0, _lib.cur_stream()), "wgrad")
    g1, n1 = eng.normalized_grads(), eng.loss_and_count()[1]        # (the launch's reduce blocks add the scorer's counts to the tail again)
    k = "item_encoder.layer.0.intermediate.dense_1.weight"
    assert relerr(g1[k] * n1, (2 * n0 * g0[k]).cpu()) < 1e-5
    assert lib.dr4sr_fmlp_launch_kernel(C.byref(plan), 99, 0, _lib.cur_stream()) != 0
    assert lib.dr4sr_fmlp_launch_kernel(C.byref(plan), 0, eng.n_layer, _lib.cur_stream()) != 0
