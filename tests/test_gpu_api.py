"""GPU tests of the drop-in surface: dr4sr_amd.model.sasrec.SASRec behind the reference's BaseModel protocol."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sasrec_oracle as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_config(dropout=0.0, n_rows=600, n_items=300, batch=64, epochs=2):
    return {
        "data": {"dataset": "synthetic-toys", "domain_name_list": ["toy"], "max_seq_len": 50, "dataset_class": "synthetic",
                 "train_file": "", "n_items": n_items, "n_rows": n_rows, "n_eval_rows": 128, "seed": 5},
        "model": {"model": "SASRec", "embed_dim": 64, "loss_fn": "bce", "hidden_size": 128, "layer_num": 2, "head_num": 2,
                  "dropout_rate": dropout, "activation": "gelu", "layer_norm_eps": 1e-12},
        "train": {"batch_size": batch, "early_stop_mode": "max", "early_stop_patience": 20, "epochs": epochs, "device": "cuda",
                  "optimizer": "adam", "learning_rate": 0.001, "weight_decay": 0, "num_neg": 1, "seed": 2023, "hip_graph": True},
        "eval": {"batch_size": 128, "cutoff": [20, 10], "val_metrics": ["ndcg", "recall"], "test_metrics": ["ndcg", "recall"],
                 "topk": 100, "save_path": "./saved/"},
    }


def build(config):
    from dr4sr_amd.utils import prepare_datasets, prepare_model, seed_everything
    seed_everything(config["train"]["seed"])
    ds = prepare_datasets(config)
    model = prepare_model(config, ds)
    return ds, model


def test_state_dict_names_and_golden_checkpoint_loads(golden_dir):
    z = np.load(os.path.join(golden_dir, "sasrec_d64.npz"))
    cfg = make_config(n_items=int(z["meta.num_items"]))
    ds, model = build(cfg)
    ref = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    assert set(model.state_dict()) == set(ref)
    model.load_state_dict(ref, strict=True)
    assert model.item_embedding.weight.data_ptr() == model.query_encoder.item_encoder.weight.data_ptr()   # tied table
    # the API forward reproduces the reference's query and eval rows
    batch = {k[6:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("batch.")}
    model.train()
    q = model.forward(batch)
    assert float((q.cpu() - torch.from_numpy(z["out.query"])).abs().max()) < 2e-4 * float(np.abs(z["out.query"]).max())
    model.eval()
    ev = {k[5:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("eval.") and k[5:] in
          ("in_item_id", "seqlen", "item_id", "user_hist", "user_id", "label")}
    model.set_eval_domain("toy")
    with torch.no_grad():
        score, items = model.topk(ev, 20, ev["user_hist"])
    assert (items.cpu().numpy() == z["eval.topk_items"]).mean() > 0.99
    np.testing.assert_allclose(score.cpu().numpy(), z["eval.topk_score"], rtol=2e-4, atol=2e-5)


def test_api_training_step_backward_and_optimizer_match_reference(golden_dir):
    """loss = model.training_step(batch); loss.backward(); optimizer.step()  ==  the reference's numbers"""
    z = np.load(os.path.join(golden_dir, "sasrec_d64.npz"))
    cfg = make_config(n_items=int(z["meta.num_items"]))
    ds, model = build(cfg)
    model._init_model(ds[0])
    model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})
    batch = {k[6:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("batch.")}
    model.train()
    model.optimizer.zero_grad()
    loss, query = model.training_step(batch, reduce=True, return_query=True)
    loss.backward()
    assert abs(float(loss) - float(z["out.loss"])) < 2e-6
    for n, p in model.named_parameters():
        ref = z["grad." + n]
        assert float(np.abs(p.grad.cpu().numpy() - ref).max()) < 2e-4 * max(1e-8, float(np.abs(ref).max())), n
    with torch.no_grad():
        lnr = model.training_step(batch, reduce=False)
    np.testing.assert_allclose(lnr.cpu().numpy(), z["out.loss_noreduce"], rtol=2e-4, atol=1e-7)
    model.optimizer.step()
    for n, p in model.named_parameters():
        well = np.abs(z["grad." + n]) > 1e-5
        d = p.detach().cpu().numpy() - z["adam1." + n]
        assert np.abs(d[well]).max(initial=0) < 5e-6 and np.abs(d).max() < 2e-4, n
    neg = model._neg_sampling(batch)
    assert neg.shape == batch["neg_item"].shape and int(neg.min()) >= 1 and int(neg.max()) < model.num_items


def test_fast_path_epoch_equals_api_path_step(golden_dir):
    """one fused-graph step == one API-path step on the same rows and the same negatives (dropout 0)"""
    cfg = make_config(dropout=0.0, n_rows=64, batch=64)
    ds, ma = build(cfg)
    ma._init_model(ds[0])
    _, mb = build(copy.deepcopy(cfg))
    mb._init_model(ds[0])
    mb.load_state_dict(ma.state_dict())
    loader = ds[0].get_loader(shuffle=False)
    ma.train()
    out = ma._fused_epoch(loader)                      # 1 batch of 64 rows, negatives drawn in-kernel
    neg = ma._neg_buf.view(64, 50, 1).clone()
    batch = next(iter(loader))
    batch["neg_item"] = neg
    mb.train()
    mb.optimizer.zero_grad()
    loss = mb.training_step(batch)
    loss.backward()
    mb.optimizer.step()
    assert abs(float(out[0]["loss_0"][0]) - float(loss)) < 2e-6
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        d = (pa - pb).abs()           # Adam's first step is lr*g/(|g|+eps): elements with |g| ~ eps amplify fp32 atomic-order noise
        assert float(d.max()) < 2e-4 and float(d.mean()) < 2e-7, n


def test_fit_and_evaluate_end_to_end(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from dr4sr_amd import quickstart
    cfg = make_config(dropout=0.2, n_rows=1500, n_items=200, batch=128, epochs=4)
    out = quickstart.run(cfg)
    assert {"ndcg@20", "recall@20", "ndcg@10", "recall@10", "toy_ndcg@20"} <= set(out)
    assert all(np.isfinite(v) for v in out.values())
    logs = [os.path.join(dp, f) for dp, _, fs in os.walk("log") for f in fs]
    ckpts = [os.path.join(dp, f) for dp, _, fs in os.walk("saved") for f in fs]
    assert len(logs) == 1 and len(ckpts) == 1 and ckpts[0].startswith("saved/SASRec/synthetic-toys/")
    ck = torch.load(ckpts[0], weights_only=False)
    assert set(ck) == {"config", "model", "epoch", "parameters", "metric"} and "ndcg@20" in ck["metric"]
    text = open(logs[0]).read()
    losses = [float(x) for x in __import__("re").findall(r"'train_loss_0': tensor\(([0-9.]+)", text)]
    assert len(losses) == 4 and losses[-1] < losses[0]            # it learns


def test_last_partial_batch_and_graph_cache(tmp_path):
    cfg = make_config(dropout=0.5, n_rows=150, batch=64, epochs=1)
    ds, model = build(cfg)
    model._init_model(ds[0])
    model.train()
    for _ in range(2):
        out = model._fused_epoch(ds[0].get_loader())
        assert out[0]["loss_0"].shape == (3,) and torch.isfinite(out[0]["loss_0"]).all()    # 64 + 64 + 22
    assert len(model._graphs) == 2                                 # one graph per distinct batch length, reused
    assert int(model.engine.state[0]) == 6


def test_fmlp_model_api_and_fast_path(golden_dir, tmp_path, monkeypatch):
    """dr4sr_amd.model.fmlp.FMLP: reference state-dict names, API-path loss/backward == reference, fit() end to end"""
    z = np.load(os.path.join(golden_dir, "fmlp_d64.npz"))
    cfg = make_config(n_items=int(z["meta.num_items"]))
    cfg["model"].update({"model": "FMLP", "layer_num": 2})
    cfg["data"]["prefix_rows"] = True
    ds, model = build(cfg)
    model._init_model(ds[0])
    ref = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    assert set(model.state_dict()) == set(ref)
    model.load_state_dict(ref, strict=True)
    model.engine.p_drop = 0.0                                   # fixture was generated with the hard-coded dropout disabled
    batch = {k[6:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("batch.")}
    model.train()
    model.optimizer.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    assert abs(float(loss) - float(z["out.loss"])) < 3e-6
    for n, p in model.named_parameters():
        r = z["grad." + n]
        assert float(np.abs(p.grad.cpu().numpy() - r).max()) < 3e-4 * max(1e-8, float(np.abs(r).max())), n
    model.optimizer.step()
    for n, p in model.named_parameters():
        well = np.abs(z["grad." + n]) > 1e-5
        d = p.detach().cpu().numpy() - z["adam1." + n]
        assert np.abs(d[well]).max(initial=0) < 1e-5, n
    # end-to-end fit on synthetic prefix rows with the fused-graph fast path
    monkeypatch.chdir(tmp_path)
    from dr4sr_amd import quickstart
    cfg2 = make_config(n_rows=1200, n_items=150, batch=128, epochs=3)
    cfg2["model"].update({"model": "FMLP", "layer_num": 2})
    cfg2["data"]["prefix_rows"] = True
    out = quickstart.run(cfg2)
    assert {"ndcg@20", "recall@20"} <= set(out) and all(np.isfinite(v) for v in out.values())


def test_gru4rec_model_api_and_fast_path(golden_dir, tmp_path, monkeypatch):
    """dr4sr_amd.model.gru4rec.GRU4Rec: reference state-dict names, API-path loss/backward == reference, fit() end to end"""
    z = np.load(os.path.join(golden_dir, "gru4rec_d64.npz"))
    cfg = make_config(n_items=int(z["meta.num_items"]))
    cfg["model"].update({"model": "GRU4Rec", "hidden_size": int(z["meta.hidden_size"]), "layer_num": 2, "dropout_rate": 0.0})
    cfg["train"]["weight_decay"] = 1e-4
    ds, model = build(cfg)
    model._init_model(ds[0])
    ref = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    assert set(model.state_dict()) == set(ref)
    model.load_state_dict(ref, strict=True)
    batch = {k[6:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("batch.")}
    model.train()
    model.optimizer.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    assert abs(float(loss) - float(z["out.loss"])) < 3e-6
    for n, p in model.named_parameters():
        r = z["grad." + n]
        assert float(np.abs(p.grad.cpu().numpy() - r).max()) < 3e-4 * max(1e-8, float(np.abs(r).max())), n
    model.optimizer.step()
    for n, p in model.named_parameters():
        well = np.abs(z["grad." + n]) > 1e-4
        d = p.detach().cpu().numpy() - z["adam1." + n]
        assert np.abs(d[well]).max(initial=0) < 1e-5, n
    monkeypatch.chdir(tmp_path)
    from dr4sr_amd import quickstart
    cfg2 = make_config(n_rows=1000, n_items=150, batch=128, epochs=3)
    cfg2["model"].update({"model": "GRU4Rec", "hidden_size": 256, "layer_num": 2, "dropout_rate": 0.2})
    cfg2["train"]["weight_decay"] = 1e-4
    out = quickstart.run(cfg2)
    assert {"ndcg@20", "recall@20"} <= set(out) and all(np.isfinite(v) for v in out.values())


@pytest.mark.gpu
@pytest.mark.parametrize("B,U", [(64, 640), (2048, 19412), (9000, 19412)])
def test_train_steps_equals_repeated_train_step(B, U):
    """dr4sr_sasrec_train_steps (one prep, optimizer launches prepare the next step) == k x dr4sr_sasrec_train_step, over
    consecutive batches of a permutation, with dropout and in-kernel negatives (same RNG steps on both sides).  B = 64: the next
    step's prep is an extra workgroup of the optimizer launch; B = 2048 / 9000 (> 1024; 9000 > one 8192-sequence chunk and wraps the
    permutation): the two-phase form — selection spread over the optimizer launch's workgroups, scan by the last one to finish."""
    import numpy as np
    from dr4sr_amd.engine import SasrecEngine
    from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
    dev = torch.device("cuda", 0)
    L, k = 50, 4
    rows = make_rows(n_rows=U, n_items=TOYS_N_ITEMS, seed=11)
    data = {n: torch.from_numpy(rows[n]).to(dev) for n in ("in_item_id", "item_id", "seqlen")}
    perm = torch.from_numpy(np.random.default_rng(3).permutation(U)).to(dev)
    out = []
    for fused in (False, True, "split"):
        eng = SasrecEngine(TOYS_N_ITEMS, L, 64, 2, 128, 2, 1e-12, 0.5, B, dev, seed=77, lr=1e-3)
        g = torch.Generator().manual_seed(5)
        for kname, v in eng.views.items():
            v.copy_(torch.ones(v.shape) if "norm" in kname and kname.endswith("weight") else 0.05 * torch.randn(v.shape, generator=g))
        eng.views["item_embedding.weight"][0] = 0
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        log = torch.zeros(16, dtype=torch.float32, device=dev)
        plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=torch.zeros(B, dtype=torch.int64, device=dev),
                             neg_item=torch.zeros(B, L, dtype=torch.int64, device=dev), sample_neg=True,
                             perm_sel=(perm, B, 0, counter), loss_log=log)
        if fused == "split" and B > 1024:                   # (the split form prepares inside the optimizer launch up to B = 1024 only)
            for _ in range(k + 1):
                eng.fwd_bwd(plan)
                eng.adam_step(plan)
        elif fused == "split":                              # the data-parallel form: halves of a step around (here: no) all-reduce
            eng.fwd_bwd(plan)
            for _ in range(k):
                eng.adam_step_prepare_next(plan)
                eng.fwd_bwd_prepared(plan)
            eng.adam_step(plan)
        elif fused:
            eng.train_steps(plan, k)
            eng.train_steps(plan, 1)
        else:
            for _ in range(k + 1):
                eng.train_step(plan)
        torch.cuda.synchronize()
        assert int(counter) == k + 1 and int(eng.state[0]) == k + 1 and int(eng.state[3]) == k + 1
        out.append((eng.params.clone(), log.clone(), eng.grads.clone()))
    (p0, l0, g0), (p1, l1, g1), (p2, l2, g2) = out
    assert torch.allclose(l0, l2, rtol=1e-5, atol=1e-6) and float((p0 - p2).abs().max()) < 2e-4
    assert float(l0[:k + 1].min()) > 0 and torch.allclose(l0, l1, rtol=1e-5, atol=1e-6)
    assert float((p0 - p1).abs().max()) < 2e-4           # same trajectory: fp32 atomics order only, amplified by Adam (lr 1e-3 per step)
    assert float((g0 - g1).abs().max()) <= 1e-4 * float(g0.abs().max())


@pytest.mark.gpu
def test_data_parallel_ranks_equal_single_rank():
    """2 ranks (gloo, sharing cuda:0) sharding every global batch + one sum-all-reduce of the flat gradient and its {n_valid, loss}
    tail == a single rank on the full batches (tools/dp_check.py); replicas stay bit-identical."""
    from _launch import report, torchrun
    out = torchrun(2, "tools/dp_check.py", {"DR4SR_DP_BACKEND": "gloo"}, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_CHECK")]
    assert out.returncode == 0 and len(lines) == 2, report(out)
    assert "replica checksums equal: True" in lines[1]


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["GRU4Rec", "FMLP", "MetaModel", "SASRec-d128", "CL4SRec", "MetaModel-CL4SRec"])
def test_data_parallel_fit_other_models(tmp_path, model_name):
    """the same end-to-end run (quickstart.run under 2 ranks sharing the GPU, uneven tail batch) for the other DP-capable models:
    GRU4Rec / FMLP step through their engines' two-graph form, MetaModel all-reduces both flat buffers and runs its outer loop;
    CL4SRec (round 4) all-gathers the pooled views of the global batch inside the step (CL4SRec._cl_term);
    SASRec at d = 128 (BASELINE configs[3]'s width: DR4SR_EMBED_DIM, the override utils/config.py:load_config honours)"""
    extra = {}
    if model_name == "SASRec-d128":
        model_name, extra = "SASRec", {"DR4SR_EMBED_DIM": "128"}
    if model_name == "MetaModel-CL4SRec":                 # round 4: the tuple-loss sub-model, its contrastive term over the gathered global batch
        model_name, extra = "MetaModel", {"SUB_MODEL": "CL4SRec"}
    from _launch import report, torchrun
    out = torchrun(2, "tools/dp_fit_check.py", dict(DR4SR_DP_BACKEND="gloo", DP_FIT_DIR=str(tmp_path), MODEL=model_name, **extra), timeout=500)
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_FIT ")]
    err = out.stdout[out.stdout.find("DP_FIT_ERROR"):][:3000] if "DP_FIT_ERROR" in out.stdout else report(out)
    assert out.returncode == 0 and len(lines) == 1, err
    assert "replicas identical: True; finite: True" in lines[0] and "one ckpt stem: True" in lines[0], lines[0]


@pytest.mark.gpu
def test_data_parallel_fit_end_to_end(tmp_path):
    """fit() under 2 ranks (gloo, sharing cuda:0) on a dataset whose tail batch splits unevenly (13 rows: rank 0 gets a new slice size,
    rank 1 an empty one): every rank must enter the same collectives (graph warm-ups stay local) and the replicas stay bit-identical"""
    from _launch import report, torchrun
    out = torchrun(2, "tools/dp_fit_check.py", dict(DR4SR_DP_BACKEND="gloo", DP_FIT_DIR=str(tmp_path)), timeout=400)
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_FIT")]
    assert out.returncode == 0 and len(lines) == 1, report(out)
    assert "replicas identical: True; finite: True" in lines[0] and "steps=27" in lines[0]
    # quickstart.run under W ranks: ONE log / checkpoint stem (rank 0's), every rank loaded rank 0's best checkpoint in evaluate()
    assert "one ckpt stem: True" in lines[0]
    assert len(list((tmp_path / "saved" / "SASRec" / "synthetic-toys").glob("*.ckpt"))) == 1


@pytest.mark.gpu
def test_bench_self_launches_n_ranks_on_a_shared_gpu():
    """`python bench.py --gpus 2` stand-alone: re-executes itself under torch.distributed.run, both ranks take the data-parallel
    step (DR4SR_BENCH_SHARE_GPU: one GPU, gloo transport — RCCL refuses two ranks on one device), rank 0 prints ONE JSON line with
    n_gpus = 2, a weak-scaled headline value and the strong-scaling object"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                          "--no-throughput-mode", "--strong-global-batch", "512"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, DR4SR_BENCH_SHARE_GPU="1"), cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, "rc=%s\n" % out.returncode + out.stdout[-3000:] + out.stderr[-6000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 512 and j["config"]["parallelism"] == "dp2"
    assert "gloo" in j["config"]["collective"] and j["value"] > 0 and j["scaling"] == "weak"
    st = j["strong"][0]
    assert st["global_batch"] == 512 and st["per_gpu_batch"] == 256 and st["n_gpus"] == 2 and st["speedup"] > 0


@pytest.mark.gpu
def test_bench_single_rank_rccl_in_graph_allreduce():
    """the default N-rank form — RCCL all-reduce captured inside the k-step graph — exercised with the one rank a 1-GPU box has
    (DR4SR_BENCH_FORCE_DP); asserts that RCCL, not a fallback, carried the reduce"""
    import json
    from _launch import report, torchrun
    out = torchrun(1, "bench.py", {"DR4SR_BENCH_FORCE_DP": "1"}, args=["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-throughput-mode",
                                                                     "--no-strong", "--no-cpu-baseline"], timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, report(out)
    j = json.loads(lines[0])
    assert "rccl all-reduce captured in the step graph" in j["config"]["collective"], j["config"]
    assert j["final_loss"] == j["final_loss"] and 0.5 < j["final_loss"] < 2.0


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["SASRec", "GRU4Rec", "FMLP", "MetaModel", "CL4SRec"])
def test_fit_learns_a_sequential_signal(tmp_path, monkeypatch, model_name):
    """end-to-end sanity of the whole loop (targets shifted by one, masks, negatives, optimizer, top-k with history masking, metrics):
    on data whose next item follows the current one through a fixed map 90 % of the time, a few epochs must lift recall@20 far above
    the popularity baseline (20 / 300 items = 0.07)"""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd import quickstart
    from dr4sr_amd.utils import load_config
    cfg = load_config({"model": model_name, "dataset": "synthetic-toys"})
    cfg["data"].update({"n_items": 300, "n_rows": 4000, "n_eval_rows": 512, "markov": 0.9, "seed": 3})
    cfg["train"].update({"device": "cuda", "epochs": 12 if model_name in ("SASRec", "CL4SRec") else 30, "batch_size": 128})
    cfg["eval"]["batch_size"] = 512
    if model_name in ("FMLP", "MetaModel"):                 # FMLP (MetaModel's default sub-model) keeps one query per row: prefix-row format
        cfg["data"]["prefix_rows"] = True
    out = quickstart.run(cfg)
    assert out["recall@20"] > 0.5 and out["ndcg@20"] > 0.2, out


@pytest.mark.gpu
@pytest.mark.parametrize("model_name,prefix", [("FMLP", False), ("MetaModel", False), ("SASRec", True), ("GRU4Rec", True)])
def test_models_reject_the_other_target_format(tmp_path, monkeypatch, model_name, prefix):
    """FMLP keeps one query per row (model/fmlp.py:38), SASRec / GRU4Rec one per position (pooling 'origin'); against the other target
    format the reference's (query * item_embedding(target)).sum(-1) (basemodel.py:182) cannot broadcast.  Here it must fail loudly,
    not read the target table with the wrong stride"""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd import quickstart, _lib
    from dr4sr_amd.utils import load_config
    cfg = load_config({"model": model_name, "dataset": "synthetic-toys"})
    cfg["data"].update({"n_items": 300, "n_rows": 512, "n_eval_rows": 128, "prefix_rows": prefix})
    cfg["train"].update({"device": "cuda", "epochs": 2, "batch_size": 128, "warmup_epoch": -1})
    with pytest.raises(_lib.Dr4srError, match="one query per|one-query-per-row"):
        quickstart.run(cfg)


@pytest.mark.gpu
def test_loss_modules_on_score_tensors_match_reference(golden_dir):
    """model.loss_func.{BinaryCrossEntropyLoss, BPRLoss}.forward on score tensors (dr4sr_loss_from_scores_*) vs the reference's two
    loss classes run on the same scores (tests/golden/loss_modules.npz): loss, d pos, d neg; 1-D / 2-D positives, K = 1 and 3"""
    from dr4sr_amd.model.loss_func import BinaryCrossEntropyLoss, BPRLoss
    z = np.load(os.path.join(golden_dir, "loss_modules.npz"))
    for tag in ("a", "b", "c", "d"):                     # d: the plain-mean branch (loss_func.py:32-33), BCE only
        for name in (("bce", "bce_nr") if tag == "d" else ("bce", "bce_nr", "bpr")):
            p = torch.from_numpy(z[f"{tag}.pos"]).cuda().requires_grad_(True)
            n = torch.from_numpy(z[f"{tag}.neg"]).cuda().requires_grad_(True)
            loss = BPRLoss()(p, n) if name == "bpr" else BinaryCrossEntropyLoss()(p, n, reduce=(name == "bce"))
            np.testing.assert_allclose(loss.detach().cpu().numpy(), z[f"{tag}.{name}.loss"], rtol=1e-5, atol=1e-7)
            (loss * torch.from_numpy(z[f"{tag}.{name}.up"]).cuda()).sum().backward()
            np.testing.assert_allclose(p.grad.cpu().numpy(), z[f"{tag}.{name}.dpos"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(n.grad.cpu().numpy(), z[f"{tag}.{name}.dneg"], rtol=1e-5, atol=1e-7)
    with pytest.raises(TypeError):
        BPRLoss()(torch.zeros(2).cuda(), torch.zeros(2, 1).cuda(), reduce=True)       # loss_func.py:44: no such parameter
    # reduced-precision score tensors: the kernels compute in fp32, the gradients come back in the inputs' dtypes (autograd contract)
    for neg_shape in ((4, 5, 2), (4, 3)):                   # masked branch / plain-mean branch
        ph = torch.randn(4, 5, device="cuda").to(torch.bfloat16).requires_grad_(True)
        nh = torch.randn(*neg_shape, device="cuda").to(torch.float16).requires_grad_(True)
        BinaryCrossEntropyLoss()(ph, nh).backward()
        assert ph.grad.dtype == torch.bfloat16 and nh.grad.dtype == torch.float16 and bool(torch.isfinite(nh.grad.float()).all())


@pytest.mark.gpu
def test_bpr_loss_training_step_matches_oracle_and_trains(golden_dir):
    """a8: loss_fn 'bpr' (basemodel.py:103-104).  training_step = encoder + tied scorer + BPRLoss (loss_func.py:44-49): loss and every
    parameter gradient vs the oracle (whose BPR restatement is pinned on the reference's BPRLoss, test_oracle_loss_modules_match_reference),
    reduce=False keeps the reference's TypeError, and fit() runs on the API path"""
    z = np.load(os.path.join(golden_dir, "sasrec_d64.npz"))
    cfg = make_config(n_items=int(z["meta.num_items"]))
    cfg["model"]["loss_fn"] = "bpr"
    ds, model = build(cfg)
    model._init_model(ds[0])
    ref = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    model.load_state_dict(ref, strict=True)
    batch_cpu = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch.")}
    batch = {k: v.cuda() for k, v in batch_cpu.items()}
    model.train()
    model.optimizer.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    loss_o, _, grads_o = O.grads_of(ref, batch_cpu, int(z["meta.head_num"]), int(z["meta.layer_num"]), float(z["meta.layer_norm_eps"]),
                                    loss_fn="bpr")
    assert abs(float(loss) - float(loss_o)) < 2e-6 * max(1.0, abs(float(loss_o)))
    for k, g in model.engine.grad_views.items():
        ref_g = grads_o[k]
        err = float((g.cpu() - ref_g).abs().max() / max(1e-12, float(ref_g.abs().max())))
        assert err < 2e-4, (k, err)
    with pytest.raises(TypeError):
        model.training_step(batch, reduce=False)
    # a few optimizer steps through the API loop reduce the loss
    first = float(loss)
    for _ in range(15):
        model.optimizer.zero_grad()
        l2 = model.training_step(batch)
        l2.backward()
        model.optimizer.step()
    assert float(l2) < first


@pytest.mark.gpu
def test_torch_custom_ops_match_oracle_and_autograd(golden_dir):
    """torch.ops.dr4sr_hip.* (dr4sr_amd/ops.py): the registered ops run the HIP kernels and differentiate through
    torch.autograd like the reference's plain-torch expressions (basemodel.py:204-214, loss_func.py): gather bit-exact, scorer
    loss / d query / d E vs the oracle, top-k vs the oracle, fused Adam vs the oracle's Adam step"""
    import dr4sr_amd.ops  # noqa: F401  (registers the ops)
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    N, D, B, L = 300, 64, 9, 50
    E = (0.1 * torch.randn(N, D, generator=g))
    E[0] = 0
    P = 0.1 * torch.randn(L, D, generator=g)
    idx = torch.randint(0, N, (B, L), generator=g)
    x = torch.ops.dr4sr_hip.embed_gather_posadd(E.to(dev), P.to(dev), idx.to(dev))
    assert torch.equal(x.cpu(), E[idx] + P.unsqueeze(0))                      # one IEEE add: bit-exact
    q = torch.randn(B, L, D, generator=g)
    tgt = torch.randint(0, N, (B, L), generator=g)
    neg = torch.randint(1, N, (B, L), generator=g)
    for name, ofn in (("score_bce", lambda qq, ee: O.score_bce(qq, ee, tgt, neg.unsqueeze(-1), True)[0]),
                      ("score_bpr", lambda qq, ee: O.score_bpr(qq, ee, tgt, neg.unsqueeze(-1))[0])):
        qd, Ed = q.to(dev).requires_grad_(True), E.to(dev).requires_grad_(True)
        lp, st = getattr(torch.ops.dr4sr_hip, name)(qd, Ed, tgt.to(dev), neg.to(dev))
        loss = lp.sum() / st[0]
        loss.backward()
        qo, Eo = q.clone().requires_grad_(True), E.clone().requires_grad_(True)
        lo = ofn(qo, Eo)
        lo.backward()
        assert abs(float(loss) - float(lo)) < 2e-6 * max(1.0, abs(float(lo))), name
        assert float((qd.grad.cpu() - qo.grad).abs().max()) < 2e-6 * float(qo.grad.abs().max()) + 1e-9, name
        assert float((Ed.grad.cpu() - Eo.grad).abs().max()) < 2e-5 * float(Eo.grad.abs().max()) + 1e-9, name
    hist = torch.randint(0, N, (B, 7), generator=g)
    sc, it = torch.ops.dr4sr_hip.full_score_topk(q[:, 0].contiguous().to(dev), E.to(dev), hist.to(dev), None, 20)
    rs, ri = O.full_score_topk(q[:, 0], E, hist, 20)
    assert float((sc.cpu() - rs).abs().max()) < 1e-5 and float((it.cpu() == ri).float().mean()) > 0.99
    ids = torch.ops.dr4sr_hip.neg_sample(4096, N, 7, 3, E.to(dev))
    assert int(ids.min()) >= 1 and int(ids.max()) <= N - 1
    # fused Adam on flat buffers vs the oracle's torch.optim.Adam formula (two steps)
    n = 1024
    p0 = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g)
    pd, m, v = p0.to(dev).clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    gbuf = torch.cat([gr, torch.tensor([1.0, 0.0, 0.0, 0.0])]).to(dev)
    state = torch.zeros(16, dtype=torch.int32, device=dev)
    po, mo, vo = {"w": p0.clone()}, {"w": torch.zeros(n)}, {"w": torch.zeros(n)}
    for t in (1, 2):
        torch.ops.dr4sr_hip.fused_adam_(pd, gbuf, m, v, state, 1e-3, 0.9, 0.999, 1e-8, 0.0)
        po = O.adam_step(po, {"w": gr}, mo, vo, t)
    assert int(state[0]) == 2 and float((pd.cpu() - po["w"]).abs().max()) < 1e-6


def _write_reference_files(root, n_users=300, n_items=120, L=50, seed=3):
    """a dataset directory in the REFERENCE's on-disk row format (data/dataset.py:56-65: torch-pickled row lists + inter.csv),
    toys-like lengths; returns the rows"""
    d = os.path.join(root, "dataset", "disk-toys", "toy")
    os.makedirs(d)
    rng = np.random.default_rng(seed)
    pad = lambda s: list(s) + [0] * (L - len(s))
    train, val, test = [], [], []
    for u in range(1, n_users + 1):
        sl = int(min(L, 1 + rng.geometric(0.18)))
        full = rng.integers(1, n_items, sl + 3).tolist()
        train.append([u, pad(full[:sl]), pad(full[1:sl + 1]), sl, [1] * sl + [0] * (L - sl), [0] * L])
        hv = full[:min(L, sl + 1)]
        val.append([u, pad(hv), full[len(hv)], len(hv), 1, [0] * L, pad(hv)])
        ht = full[:min(L, sl + 2)]
        test.append([u, pad(ht), full[len(ht)], len(ht), 1, [0] * L, pad(ht)])
    torch.save(train, os.path.join(d, "train_ori.pth"))
    torch.save(val, os.path.join(d, "val.pth"))
    torch.save(test, os.path.join(d, "test.pth"))
    with open(os.path.join(d, "inter.csv"), "w") as f:
        f.write("user_id,item_id,rating,timestamp,domain\n")
        for i in range(1, n_items):
            f.write(f"{(i - 1) % n_users + 1},{i},1.0,{i},0\n")
    return train, val, test


@pytest.mark.parametrize("model_name", ["SASRec", "GRU4Rec"])
def test_fit_from_reference_on_disk_files(tmp_path, monkeypatch, model_name):
    """§8(f)-2 on the GPU: `quickstart.run` on a dataset directory in the reference's own file format (`dataset_class: general`,
    train_ori.pth / val.pth / test.pth / inter.csv) — first run parses the pickles and writes the packed images, second run trains
    from the images; both see the same rows (same seed -> the same metrics up to the fp32 order of the table-gradient atomics),
    checkpoints land where the reference puts them"""
    from dr4sr_amd.quickstart import run
    from dr4sr_amd.utils import load_config
    train, val, test = _write_reference_files(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))

    def cfg():
        c = load_config({"model": model_name, "dataset": "amazon-toys"})
        c["data"].update({"dataset": "disk-toys", "domain_name_list": ["toy"], "train_file": "_ori"})
        c["train"].update({"epochs": 3, "batch_size": 64, "device": "cuda", "seed": 11})
        c["eval"].update({"batch_size": 128})
        return c

    from dr4sr_amd.utils import seed_everything
    seed_everything(11)                                        # the reference's run.py entry does this (utils/utils.py:14-20)
    out1 = run(cfg())
    d = tmp_path / "dataset" / "disk-toys" / "toy"
    assert (d / "train_ori.pth.dr4srpk").exists() and (d / "val.pth.dr4srpk").exists() and (d / "test.pth.dr4srpk").exists()
    calls = []
    real_load = torch.load
    monkeypatch.setattr(torch, "load", lambda *a, **k: calls.append(str(a[0])) or real_load(*a, **k))
    seed_everything(11)
    out2 = run(cfg())
    assert not [c for c in calls if c.endswith(".pth")], calls           # the second run never unpickles the row lists
    assert set(out1) == set(out2) and all(np.isfinite(v) for v in out1.values())
    assert all(abs(out1[k] - out2[k]) < 0.03 for k in out1), (out1, out2)
    assert len(list((tmp_path / "saved" / model_name / "disk-toys").glob("*.ckpt"))) == 2


def test_regime_follows_the_expected_token_hint(monkeypatch):
    """include/dr4sr_hip.h dr4sr_sasrec_plan.expected_tokens: the launch forms follow the host's estimate of VALID tokens (the measured
    crossovers of tools/regime_sweep.sh: ~7 k for the token-tile kernels, ~14 k for the attention lists), not the capacity B * L;
    0 = unknown keeps the capacity rule (16 384); DR4SR_FORCE_SCALE / DR4SR_FORCE_ATTN_SPLIT override.  The engine fills the hint
    from the seqlen tensor of the plan."""
    import ctypes as C
    from dr4sr_amd import _lib
    from dr4sr_amd.engine import SasrecEngine
    lib = _lib.load()
    eng = SasrecEngine(500, 50, 64, 2, 128, 2, 1e-12, 0.0, 4096, "cuda")
    dev = eng.device
    ids = torch.ones(4096, 50, dtype=torch.int64, device=dev)

    def scale(B, lens, hint=None):
        plan = eng.make_plan(ids[:B], ids[:B], lens[:B].contiguous())
        if hint is not None:
            plan.expected_tokens = hint
        return int(lib.dr4sr_sasrec_at_scale(C.byref(plan)))
    short, full = torch.full((4096,), 5, dtype=torch.int64, device=dev), torch.full((4096,), 50, dtype=torch.int64, device=dev)
    # bit 0: at-scale token-tile kernels (> ~7 k expected tokens), bit 1: length-class attention lists (> ~14 k),
    # bit 2 (round 4): the attention runs inside the 16-token tile kernels — the latency regime (neither of the other two)
    # round 6: short-sequence plans switch tiles AND attention at ~7.7 k expected tokens (wave-per-tile attention, bit 4); the middle regime
    # (at-scale tiles, one attention workgroup per sequence) only exists for the cross-check forms
    assert scale(256, short) == 4 and scale(1400, short) == 4 and scale(1500, short) == 4     # 1 280 / 7 000 / 7 500 expected tokens
    assert scale(1600, short) == 3 | 16 and scale(2000, short) == 3 | 16 and scale(4096, short) == 3 | 16      # 8 000 / 10 000 / 20 480
    monkeypatch.setenv("DR4SR_ATTN_LISTS", "1")                                                # the lists keep the round-4 boundaries
    assert scale(2000, short) == 4 and scale(2100, short) == 1 and scale(2800, short) == 1 and scale(2900, short) == 3
    monkeypatch.delenv("DR4SR_ATTN_LISTS")
    # batches of LONG sequences (expected mean length > 16) never take the attention lists: one 8-wave workgroup per sequence is faster
    # at every size (round 3, tools/regime_sweep3.sh --dense)
    assert scale(120, full) == 4 and scale(128, full) == 1 and scale(160, full) == 1 and scale(300, full) == 1 and scale(4096, full) == 1      # 6 000 / 6 400 / 8 000 / 15 000 / 204 800 tokens (long sequences: boundary 6 144)
    mid = torch.full((4096,), 16, dtype=torch.int64, device=dev)
    assert scale(1000, mid) == 3 | 16                                                            # 16 000 tokens, mean length 16: lists
    assert scale(256, short, hint=0) == 4 and scale(400, short, hint=0) == 3 | 16                   # unknown: capacity 12 800 / 20 000 decides both
    monkeypatch.setenv("DR4SR_FORCE_SCALE", "1")
    assert scale(64, short) == 3 | 16
    monkeypatch.setenv("DR4SR_ATTN_LISTS", "1")                                                # cross-check: the length-class lists
    assert scale(64, short) == 3
    monkeypatch.delenv("DR4SR_ATTN_LISTS")
    monkeypatch.setenv("DR4SR_FORCE_ATTN_SPLIT", "0")
    assert scale(64, short) == 1
    monkeypatch.delenv("DR4SR_FORCE_ATTN_SPLIT")
    monkeypatch.setenv("DR4SR_FORCE_SCALE", "0")
    assert scale(4096, full) == 4
    monkeypatch.setenv("DR4SR_ATTN_SEPARATE", "1")                                             # cross-check: one workgroup per sequence
    assert scale(4096, full) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adam", "sgd", "adagrad", "rmsprop"])
def test_optimizer_choices_match_torch_optim(kind):
    """basemodel.py:79-98 (round 4): the reference's `optimizer` choices through the one fused flat-buffer launch (k_adam<OPT>,
    csrc/step.hip) — dr4sr_optimizer_flat on random buffers, 4 steps with the un-normalised gradient + n_valid tail the training step
    leaves, against oracle/optim_oracle.py (pinned on torch.optim); and through a SASRec engine's plan (plan->optimizer)"""
    from dr4sr_amd import _lib
    from oracle import optim_oracle as OO
    lib = _lib.load()
    code, betas, eps, wd = _lib.optimizer_settings(kind, 1e-2)
    n, lr, nv = 4096 + 8, 1e-2, 37.0
    gen = torch.Generator().manual_seed(7)
    p0 = torch.randn(n, generator=gen)
    P, M, V = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device="cuda")
    p, st = p0.clone(), OO.init_state(p0)
    for i in range(4):
        g = torch.randn(n, generator=gen) * torch.rand(n, generator=gen)
        G = torch.cat([g * nv, torch.tensor([nv, 0.0, 0.0, 0.0])]).cuda()         # un-normalised sum + {n_valid, loss_sum, poison, -}
        _lib.check(lib.dr4sr_optimizer_flat(code, _lib.ptr(P), _lib.ptr(G), _lib.ptr(M), _lib.ptr(V), n, _lib.ptr(state), lr, betas[0], betas[1],
                                            eps, wd, _lib.cur_stream()), "dr4sr_optimizer_flat")
        p = OO.step(kind, p, g, st, lr, wd)
        err = float((P.cpu() - p).abs().max())
        assert err < 3e-6, (kind, i, err)
    assert int(state[_lib.STATE_STEP]) == 4
    if kind == "sgd":
        assert float(M.abs().max()) == 0.0 and float(V.abs().max()) == 0.0        # SGD keeps no moments (and moves no bytes for them)
    # a poisoned step is skipped by every kind
    G[n + 2] = 1.0
    before = P.clone()
    _lib.check(lib.dr4sr_optimizer_flat(code, _lib.ptr(P), _lib.ptr(G), _lib.ptr(M), _lib.ptr(V), n, _lib.ptr(state), lr, betas[0], betas[1], eps, wd,
                                        _lib.cur_stream()), "dr4sr_optimizer_flat")
    assert torch.equal(P, before) and int(state[_lib.STATE_STEP]) == 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sgd", "adagrad", "rmsprop", "lamb"])
def test_model_optimizer_config_trains(tmp_path, monkeypatch, name):
    """`train.optimizer` of the reference's config (basemodel.py:79-98) through fit(): the fused k-step graphs carry the chosen update;
    an unknown name ('lamb') falls back to Adam without weight decay as the reference's else branch; 'sparse_adam' raises what torch
    raises on the reference's dense gradients"""
    monkeypatch.chdir(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(root, "configs"))
    from dr4sr_amd import _lib
    from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
    cfg = load_config({"model": "SASRec", "dataset": "synthetic-toys"})
    cfg["data"].update({"n_items": 200, "n_rows": 512, "n_eval_rows": 64, "seed": 5})
    cfg["train"].update({"batch_size": 64, "epochs": 1, "device": "cuda", "optimizer": name, "weight_decay": 1e-3,
                         "learning_rate": {"sgd": 0.5, "adagrad": 0.05, "rmsprop": 0.003, "lamb": 0.003}[name]})
    seed_everything(cfg["train"]["seed"])
    ds = prepare_datasets(cfg)
    model = prepare_model(cfg, ds)
    model._init_model(ds[0])
    eng = model.engine
    want = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "rmsprop": _lib.OPT_RMSPROP, "lamb": _lib.OPT_ADAM}[name]
    assert eng.optimizer == want and eng.weight_decay == (0.0 if name == "lamb" else 1e-3)
    model.train()
    first = last = None
    for ep in range(6):
        out = model.training_epoch(ep)
        loss = float(torch.stack([o["loss_0"].float().mean() if torch.is_tensor(o["loss_0"]) else torch.tensor(o["loss_0"]) for o in out[0]]).mean())
        first = loss if first is None else first
        last = loss
    assert np.isfinite(last) and last < first, (name, first, last)
    cfg["train"]["optimizer"] = "sparse_adam"
    with pytest.raises(RuntimeError, match="SparseAdam does not support dense gradients"):
        prepare_model(cfg, ds)._init_model(ds[0])


@pytest.mark.gpu
def test_out_of_range_item_ids_raise_like_nn_embedding(monkeypatch):
    """the gather kernels clamp ids outside the table where torch's nn.Embedding (the reference's gather,
    model/sasrec.py:43) raises IndexError.  dr4sr_check_ids counts the offenders: the dense dispatcher op raises per call, the models
    raise once when they are initialised on a malformed dataset — before any step runs"""
    import dr4sr_amd.ops  # noqa: F401
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(root, "configs"))
    E, P = torch.randn(50, 64, device="cuda"), torch.randn(50, 64, device="cuda")
    idx = torch.randint(0, 50, (4, 50), device="cuda")
    torch.ops.dr4sr_hip.embed_gather_posadd(E, P, idx)                   # in range: fine
    for bad in (50, -1, 10 ** 9):
        idx2 = idx.clone()
        idx2[2, 7] = bad
        with pytest.raises(IndexError, match="index out of range in self"):
            torch.ops.dr4sr_hip.embed_gather_posadd(E, P, idx2)
    from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
    cfg = load_config({"model": "SASRec", "dataset": "synthetic-toys"})
    cfg["data"].update({"n_items": 120, "n_rows": 200, "n_eval_rows": 64, "seed": 5})
    cfg["train"].update({"batch_size": 64, "device": "cuda"})
    seed_everything(1)
    ds = prepare_datasets(cfg)
    prepare_model(cfg, ds)._init_model(ds[0])                            # a well-formed dataset passes
    f = ds[0].fields()
    f["in_item_id"][5, 0] = 120                                           # = num_items: one past the table
    with pytest.raises(IndexError, match="index out of range in self"):
        prepare_model(cfg, ds)._init_model(ds[0])
