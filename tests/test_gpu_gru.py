"""GPU parity of the GRU4Rec path (dr4sr_gru4rec_* through the C ABI) vs golden vectors from the reference and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gru4rec_oracle as GO  # noqa: E402

REL = 3e-4


def relerr(a, b):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


def test_gru4rec_vs_golden(golden_dir):
    from dr4sr_amd import _lib
    from dr4sr_amd.gru_engine import GruEngine
    z = np.load(os.path.join(golden_dir, "gru4rec_d64.npz"))
    g = {k: z[k] for k in z.files}
    params = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    b = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    B = b["in_item_id"].shape[0]
    eng = GruEngine(int(g["meta.num_items"]), 50, 64, int(g["meta.hidden_size"]), int(g["meta.layer_num"]), 0.0, B, "cuda",
                    lr=float(g["meta.lr"]), weight_decay=float(g["meta.weight_decay"]))
    eng.load_named(params)
    dev = eng.device
    idx, tgt, sl = b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev)
    neg = b["neg_item"].squeeze(-1).contiguous().to(dev)
    plan = eng.make_plan(idx, tgt, sl, neg_item=neg, sample_neg=False)
    q = eng.encode(plan, False, _lib.POOL_ORIGIN)
    assert relerr(q, g["out.query"]) < REL
    ql = eng.encode(eng.make_plan(torch.from_numpy(g["eval.in_item_id"]).to(dev), None, torch.from_numpy(g["eval.seqlen"]).to(dev)),
                    False, _lib.POOL_LAST)
    assert relerr(ql, g["eval.query_last"]) < REL
    eng.fwd_bwd(plan)
    loss, n = eng.loss_and_count()
    assert n == int((b["item_id"] != 0).sum()) and abs(loss - float(g["out.loss"])) < 1e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, g["grad." + k]) < REL, k
    eng.adam_step()
    for k, v in eng.views.items():
        well = np.abs(g["grad." + k]) > 1e-4
        d = v.cpu().numpy() - g["adam1." + k]
        assert np.abs(d[well]).max(initial=0) < 1e-5 and np.abs(d).max() < 1.1e-3, k


@pytest.mark.parametrize("dense", [False, True])
def test_gru4rec_full_size_vs_oracle(dense):
    """BASELINE config 3 shapes: H = 256, 2 layers, d = 64, beauty-sized table, B = 64 rows (oracle BPTT on CPU stays fast)"""
    from dr4sr_amd.data.synthetic import make_rows
    from dr4sr_amd.gru_engine import GruEngine, gru_param_names, gru_param_shapes
    N, B, H = 12102, 64, 256
    rows = make_rows(n_rows=B, n_items=N, seed=3, dense=dense)
    b = {k: torch.from_numpy(rows[k]) for k in ("in_item_id", "item_id", "seqlen")}
    gen = torch.Generator().manual_seed(1)
    b["neg_item"] = torch.randint(1, N, (B, 50, 1), generator=gen)
    params = {}
    for nme, shp in zip(gru_param_names(2), gru_param_shapes(N, 64, H, 2)):
        params[nme] = 0.08 * torch.randn(shp, generator=gen)
    params["item_embedding.weight"][0] = 0
    p = 0.2
    eng = GruEngine(N, 50, 64, H, 2, p, B, "cuda", seed=5)
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    step = int(eng.state[3])
    mask = eng.dropout_mask(B * 50 * 64, 0, step).view(B, 50, 64).cpu()
    op = dict(params)
    op["query_encoder.0.1.weight"] = params["item_embedding.weight"]
    loss_o, _, grads_o = GO.grads_of(op, b, 2, mask=mask, pdrop=p)
    loss, n = eng.loss_and_count()
    assert n == int((b["item_id"] != 0).sum()) and abs(loss - float(loss_o)) < 3e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, grads_o[k]) < REL, k
    first = loss
    for _ in range(20):
        eng.train_step(plan)
    assert eng.loss_and_count()[0] < first and int(eng.state[0]) == 20


@pytest.mark.parametrize("B", [1, 3, 17, 40, 100])
def test_gru4rec_odd_batch_sizes_vs_oracle(B):
    """batch sizes that are not multiples of the 16-sequence groups of the cooperative recurrence (partial last group, single
    sequence), random lengths including 1 and L"""
    _gru_case(B, 256, 2, 50)


@pytest.mark.parametrize("H,NL,L", [(128, 2, 50), (256, 1, 50), (256, 3, 20), (128, 1, 64), (256, 2, 1), (128, 3, 7)])
def test_gru4rec_every_accepted_shape_vs_oracle(H, NL, L):
    """hidden 128 / 256, 1-3 stacked layers, L from 1 to 64: whatever the plan check of csrc/gru.hip accepts must reproduce the oracle"""
    _gru_case(37, H, NL, L)


@pytest.mark.parametrize("H,NL,L", [(64, 2, 50), (512, 2, 50), (256, 2, 100)])
def test_gru4rec_unsupported_shapes_are_refused(H, NL, L):
    from dr4sr_amd.gru_engine import GruEngine
    from dr4sr_amd import _lib
    with pytest.raises(_lib.Dr4srError):
        GruEngine(200, L, 64, H, NL, 0.0, 8, "cuda", seed=5)


def _gru_case(B, H, NL, L):
    from dr4sr_amd.gru_engine import GruEngine, gru_param_names, gru_param_shapes
    N = 200
    gen = torch.Generator().manual_seed(B)
    sl = torch.randint(1, L + 1, (B,), generator=gen)
    sl[0] = 1
    sl[-1] = L
    inp = torch.zeros(B, L, dtype=torch.int64)
    tgt = torch.zeros(B, L, dtype=torch.int64)
    for r in range(B):
        n = int(sl[r])
        inp[r, :n] = torch.randint(1, N, (n,), generator=gen)
        tgt[r, :n] = torch.randint(0, N, (n,), generator=gen)
    b = {"in_item_id": inp, "item_id": tgt, "seqlen": sl, "neg_item": torch.randint(1, N, (B, L, 1), generator=gen)}
    params = {}
    for nme, shp in zip(gru_param_names(NL), gru_param_shapes(N, 64, H, NL)):
        params[nme] = 0.08 * torch.randn(shp, generator=gen)
    params["item_embedding.weight"][0] = 0
    eng = GruEngine(N, L, 64, H, NL, 0.0, B, "cuda", seed=5)
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    op = dict(params)
    op["query_encoder.0.1.weight"] = params["item_embedding.weight"]
    loss_o, _, grads_o = GO.grads_of(op, b, NL)
    loss, n = eng.loss_and_count()
    assert n == int((b["item_id"] != 0).sum()) and abs(loss - float(loss_o)) < 3e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, grads_o[k]) < REL, k


@pytest.mark.parametrize("nofuse", [False, True])
def test_gru4rec_train_steps_equal_repeated_single_steps(monkeypatch, nofuse):
    """round 4: dr4sr_gru4rec_train_steps — device-side batch selection from the epoch permutation (plan->perm), ONE prep launch, every
    optimizer launch preparing the next step, per-step loss log — against n single dr4sr_gru4rec_train_step calls on the same plan
    (dropout 0.2, in-kernel negatives: same Philox streams); DR4SR_NO_PREP_FUSE = a prep launch per step inside train_steps"""
    from dr4sr_amd.data.synthetic import make_rows
    from dr4sr_amd.gru_engine import GruEngine, gru_param_names, gru_param_shapes
    if nofuse:
        monkeypatch.setenv("DR4SR_NO_PREP_FUSE", "1")
    N, U, B, H, n = 3000, 700, 96, 256, 5
    rows = make_rows(n_rows=U, n_items=N, seed=8)
    gen = torch.Generator().manual_seed(2)
    params = {nme: 0.08 * torch.randn(shp, generator=gen) for nme, shp in zip(gru_param_names(2), gru_param_shapes(N, 64, H, 2))}
    params["item_embedding.weight"][0] = 0
    out = []
    for mode in ("steps", "single"):
        eng = GruEngine(N, 50, 64, H, 2, 0.2, B, "cuda", seed=5, lr=1e-3, weight_decay=1e-4)
        eng.load_named(params)
        dev = eng.device
        data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
        perm = torch.randperm(U, generator=torch.Generator().manual_seed(3)).to(dev)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        rows_buf = torch.zeros(B, dtype=torch.int64, device=dev)
        log = torch.zeros(16, dtype=torch.float32, device=dev)
        plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=rows_buf, sample_neg=True,
                             perm_sel=(perm, B, 0, counter), loss_log=log)
        if mode == "steps":
            eng.train_steps(plan, n)
        else:
            for _ in range(n):
                eng.train_step(plan)
        eng.check_device_error()
        assert int(counter) == n and int(eng.state[0]) == n
        assert torch.equal(rows_buf, perm[(n - 1) * B:n * B])              # the last step's batch
        out.append((eng.params.clone(), log.clone()))
    (pa, la), (pb, lb) = out
    assert float(la[:n].min()) > 0 and torch.allclose(la, lb, rtol=1e-5, atol=1e-6), (la, lb)
    assert float((pa - pb).abs().max()) < 2e-5 * float(pb.abs().max())      # fp32 atomics order only
