"""CPU tests of the host-side logic (no GPU): config merge, dataset row format, batch loader, metrics, early
stopping, C-ABI symbol export, parameter layout, and the data-parallel math over gloo with world_size 2."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dr4sr_amd import _lib
    lib = _lib.load()
    # the product ABI and the test / measurement hooks are separate headers; both are exported by the one library
    hdr = open(os.path.join(ROOT, "include", "dr4sr_hip.h")).read()
    hooks = open(os.path.join(ROOT, "include", "dr4sr_hip_hooks.h")).read()
    product, hook_syms = set(re.findall(r"\b(dr4sr_[a-z0-9_]+)\s*\(", hdr)), set(re.findall(r"\b(dr4sr_[a-z0-9_]+)\s*\(", hooks))
    assert product and hook_syms and not (product & hook_syms), "no declarations parsed / a hook declared in the product header"
    assert hook_syms == {"dr4sr_dropout_mask", "dr4sr_sasrec_launch_kernel", "dr4sr_sasrec_launch_kernel_weighted", "dr4sr_gru4rec_launch_kernel",
                         "dr4sr_fmlp_launch_kernel", "dr4sr_reload_env", "dr4sr_crash_line_set", "dr4sr_build_flags"}
    declared = product | hook_syms
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.dr4sr_abi_version() == _lib.ABI_VERSION
    # the ctypes mirror of the plan struct must match the C layout the library was compiled with
    assert C.sizeof(_lib.SasrecPlan) == lib.dr4sr_sasrec_plan_sizeof()


def test_transport_entry_points_bind_rccl_directly():
    """ABI 8 (SURVEY 8(b) `allreduce_flat(buf)` (RCCL), 8(e) ncclAllReduce over the flat gradient): the library itself imports RCCL's
    collectives — no torch.distributed ProcessGroupNCCL (and none of its threads) between a training step and its all-reduce"""
    import subprocess
    so = os.path.join(ROOT, "dr4sr_amd", "csrc", "libdr4sr_hip.so")
    nm = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    undefined = {l.split()[-1] for l in nm.splitlines() if " U " in l}
    assert {"ncclAllReduce", "ncclAllGather", "ncclBroadcast", "ncclCommInitRank", "ncclGetUniqueId", "ncclCommDestroy"} <= undefined
    defined = {l.split()[-1] for l in nm.splitlines() if " T " in l}
    assert {"dr4sr_comm_unique_id", "dr4sr_comm_init_rank", "dr4sr_comm_destroy", "dr4sr_allreduce_f32", "dr4sr_allreduce_f32_async",
            "dr4sr_comm_join", "dr4sr_allgather_bytes", "dr4sr_broadcast_bytes"} <= defined
    from dr4sr_amd import _lib
    lib = _lib.load()
    assert lib.dr4sr_comm_error_string(0) == b"ok" and lib.dr4sr_comm_error_string(-1) == b"argument error"
    assert b"invalid usage" in lib.dr4sr_comm_error_string(-100 - 5)          # DR4SR_E_RCCL_BASE - ncclInvalidUsage
    # argument checks need no GPU
    assert lib.dr4sr_comm_unique_id(None) == -1 and lib.dr4sr_allreduce_f32(None, None, 4, None) == -1 and lib.dr4sr_comm_join(None, None) == -1
    # the Python transport never names c10d's NCCL backend
    src = open(os.path.join(ROOT, "dr4sr_amd", "parallel.py")).read()
    assert 'init_process_group("nccl"' not in src and 'init_process_group("gloo")' in src


def test_crash_line_survives_an_abort(tmp_path):
    """bench.py's abort safety (include/dr4sr_hip_hooks.h dr4sr_crash_line_set): a process that armed a line and is then killed by SIGABRT
    — what an uncaught exception in a foreign thread does, uncatchable from Python — still leaves exactly that line on the given fd and
    exits 0; disarmed, the abort is an abort."""
    import subprocess
    import sys
    code = ("import ctypes as C, os, sys\n"
            "lib = C.CDLL(%r)\n"
            "lib.dr4sr_crash_line_set.argtypes = [C.c_char_p, C.c_int32, C.c_int32]\n"
            "fd = os.dup(1)\n"
            "assert lib.dr4sr_crash_line_set(b'{\"value\": 1, \"aborted_during\": \"x\"}', fd, 0) == 0\n"
            "assert lib.dr4sr_crash_line_set(b'{\"value\": 2, \"aborted_during\": \"y\"}', fd, 0) == 0\n"
            "if sys.argv[1] == 'disarm': lib.dr4sr_crash_line_set(None, 0, 0)\n"
            "if sys.argv[1] == 'thread':\n"
            "    import threading\n"
            "    t = threading.Thread(target=os.abort); t.start(); t.join()\n"
            "if sys.argv[1] == 'term':\n"
            "    import signal; os.kill(os.getpid(), signal.SIGTERM)\n"
            "    import time; time.sleep(5)\n"
            "os.abort()\n") % os.path.join(ROOT, "dr4sr_amd", "csrc", "libdr4sr_hip.so")
    for mode in ("main", "thread", "term"):
        out = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout == '{"value": 2, "aborted_during": "y"}\n', (mode, out.returncode, out.stdout, out.stderr[-500:])
    out = subprocess.run([sys.executable, "-c", code, "disarm"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and out.stdout == ""


def test_param_layout_matches_reference_state_dict(golden_dir):
    from dr4sr_amd import _lib
    from dr4sr_amd.engine import param_names, param_shapes
    lib = _lib.load()
    z = np.load(os.path.join(golden_dir, "sasrec_d64.npz"))
    N = int(z["meta.num_items"])
    names, shapes = param_names(2), param_shapes(N, 50, 64, 128, 2)
    off = (C.c_int64 * 26)()
    n = lib.dr4sr_sasrec_param_layout(N, 50, 64, 128, 2, off)
    assert n == sum(int(np.prod(s)) for s in shapes)
    o = 0
    for i, (nm, sh) in enumerate(zip(names, shapes)):
        assert off[i] == o and tuple(z["param." + nm].shape) == tuple(sh), nm
        o += int(np.prod(sh))
    ref_keys = {k[6:] for k in z.files if k.startswith("param.")}
    assert ref_keys == set(names) | {"query_encoder.item_encoder.weight"}
    assert lib.dr4sr_sasrec_param_layout(11925, 50, 64, 128, 2, None) == 833344


def test_plan_validation_errors():
    from dr4sr_amd import _lib
    lib = _lib.load()
    p = _lib.SasrecPlan()
    assert lib.dr4sr_sasrec_fwd_bwd(C.byref(p), None) == -1            # abi_version 0
    p.abi_version = _lib.ABI_VERSION
    p.B, p.L, p.D, p.H, p.F, p.n_layer, p.n_items = 4, 50, 96, 2, 128, 2, 100
    p.params = p.state = p.in_item_id = p.seqlen = 1                 # never dereferenced: shape check comes first
    assert lib.dr4sr_sasrec_fwd_bwd(C.byref(p), None) == -2            # unsupported D
    p.D, p.L = 64, 80
    assert lib.dr4sr_sasrec_fwd_bwd(C.byref(p), None) == -2            # L > 64
    p.L = 50
    assert lib.dr4sr_sasrec_workspace_bytes(C.byref(p)) > 0
    assert lib.dr4sr_sasrec_fwd_bwd(C.byref(p), None) == -1            # no workspace
    assert lib.dr4sr_neg_sample(None, 4, 10, 0, 0, None) == -1
    assert lib.dr4sr_dropout_mask(None, 8, 0.5, 0, 0, 0, None) == -1
    # fused batch selection needs a writable rows buffer and a counter
    p.workspace, p.workspace_bytes, p.grads, p.item_id, p.neg_item = 1, 1 << 40, 1, 1, 1
    p.n_params = lib.dr4sr_sasrec_param_layout(100, 50, 64, 128, 2, None)
    p.perm, p.n_perm = 1, 10
    assert lib.dr4sr_sasrec_fwd_bwd(C.byref(p), None) == -1            # perm without rows / perm_counter


def test_argument_validation_of_the_widening_entry_points():
    """MetaModel / CL4SRec / GRU entry points reject bad arguments before touching the device (ctypes, no GPU needed)"""
    from dr4sr_amd import _lib
    lib = _lib.load()
    one = C.c_void_p(16)
    assert lib.dr4sr_meta_param_count(64) == 64 * 64 + 64 + 2 * 64 + 2 and lib.dr4sr_meta_param_count(96) == -2
    assert lib.dr4sr_meta_select_workspace_floats(12800) > 0
    assert lib.dr4sr_meta_select_fwd(None, one, None, 0, 0, None, 1.0, None, one, 4, 50, 64, None, None, one, None) == -1      # no query
    assert lib.dr4sr_meta_select_fwd(one, one, None, 0, 0, None, 0.0, None, one, 4, 50, 64, None, None, one, None) == -1       # tau <= 0
    assert lib.dr4sr_meta_select_fwd(one, one, None, 0, 0, None, 1.0, None, one, 4, 50, 128, None, None, one, None) == -2      # D != 64
    assert lib.dr4sr_meta_select_bwd(one, one, None, 0, 0, None, 1.0, None, one, 4, 50, 64, None, one, None, None, one, None, None) == -1
    assert lib.dr4sr_fd_step_size(None, one, 8, 1e-3, one, None) == -1 and lib.dr4sr_fd_shift(one, one, one, None, 1.0, 8, None) == -1
    assert lib.dr4sr_meta_sgd_step(one, one, one, 0, 1e-3, 0.9, 0.0, 10.0, one, None, None) == -1
    assert lib.dr4sr_cl_augment(one, one, one, one, 4, 80, 0, 0.2, 0.7, 0.2, 10, 0, 0, None) == -1                              # L > 64
    assert lib.dr4sr_cl_augment(one, one, one, one, 4, 50, 7, 0.2, 0.7, 0.2, 10, 0, 0, None) == -1                              # bad mode
    assert lib.dr4sr_cl_augment(one, one, one, one, 0, 50, 0, 0.2, 0.7, 0.2, 10, 0, 0, None) == 0                               # empty batch
    assert lib.dr4sr_infonce_fwd(one, one, None, 8, 96, 1.0, one, one, one, None) == -2                                         # D
    assert lib.dr4sr_infonce_fwd(one, one, None, 8, 64, 0.0, one, one, one, None) == -1                                         # temperature
    assert lib.dr4sr_infonce_bwd(one, one, None, 8, 64, 1.0, None, None, one, one, None) == -1
    assert lib.dr4sr_neg_sample_dev(one, 8, 10, 0, None, None) == -1
    g = _lib.GruPlan()
    assert lib.dr4sr_gru4rec_fwd_bwd(C.byref(g), None) == -1
    f = _lib.FmlpPlan()
    assert lib.dr4sr_fmlp_fwd_bwd(C.byref(f), None) == -1


def test_load_config_three_way_merge(tmp_path, monkeypatch):
    from dr4sr_amd.utils import load_config
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    cfg = load_config({"model": "SASRec", "dataset": "amazon-toys"})
    assert set(cfg) == {"model", "data", "train", "eval"}
    assert cfg["data"]["dataset"] == "amazon-toys" and cfg["data"]["domain_name_list"] == ["toy"]
    assert cfg["data"]["max_seq_len"] == 50 and cfg["data"]["dataset_class"] == "general" and cfg["data"]["train_file"] == "_regen"
    assert cfg["train"]["batch_size"] == 256 and cfg["train"]["learning_rate"] == 0.001 and cfg["train"]["seed"] == 2023
    m = cfg["model"]
    assert (m["embed_dim"], m["hidden_size"], m["layer_num"], m["head_num"], m["dropout_rate"]) == (64, 128, 2, 2, 0.5)
    assert m["activation"] == "gelu" and m["layer_norm_eps"] == 1e-12 and m["loss_fn"] == "bce" and m["model"] == "SASRec"
    assert cfg["eval"]["batch_size"] == 2048 and cfg["eval"]["cutoff"] == [20, 10] and cfg["eval"]["topk"] == 100
    monkeypatch.setenv("DR4SR_EMBED_DIM", "128")
    assert load_config({"model": "SASRec", "dataset": "yelp"})["model"]["embed_dim"] == 128


def _write_dataset(root, n_items=40, seqlens=(1, 3, 50, 7, 2)):
    d = os.path.join(root, "dataset", "tiny", "toy")
    os.makedirs(d)
    rng = np.random.default_rng(0)
    pad = lambda s: list(s) + [0] * (50 - len(s))
    train, val = [], []
    for u, sl in enumerate(seqlens, 1):
        full = rng.integers(1, n_items, sl + 2).tolist()
        train.append([u, pad(full[:sl]), pad(full[1:sl + 1]), sl, [1] * sl + [0] * (50 - sl), [0] * 50])
        val.append([u, pad(full[:sl]), full[sl], sl, 1, [0] * 50, pad(full[:sl])])
    torch.save(train, os.path.join(d, "train_ori.pth"))
    torch.save(val, os.path.join(d, "val.pth"))
    torch.save(val, os.path.join(d, "test.pth"))
    with open(os.path.join(d, "inter.csv"), "w") as f:
        f.write("user_id,item_id,rating,timestamp,domain\n")
        for i in range(1, n_items):
            f.write(f"{(i - 1) % len(seqlens) + 1},{i},1.0,{i},0\n")
    return train, val


def test_separate_dataset_reads_reference_row_format(tmp_path, monkeypatch):
    from dr4sr_amd.data.dataset import SeparateDataset
    train, val = _write_dataset(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    cfg = {"data": {"dataset": "tiny", "domain_name_list": ["toy"], "max_seq_len": 50, "train_file": "_ori"},
           "train": {"device": "cpu", "batch_size": 2}, "eval": {"batch_size": 4}}
    tr = SeparateDataset(cfg, "train")
    tr.build()
    assert tr.num_items == 40 and tr.num_users == 6 and len(tr) == 5
    f = tr.fields()
    assert f["in_item_id"].shape == (5, 50) and f["in_item_id"].dtype == torch.int64
    assert f["seqlen"].tolist() == [1, 3, 50, 7, 2]
    assert f["item_id"][1, :3].tolist() == train[1][2][:3] and f["label"][2].sum() == 50
    loader = tr.get_loader()
    assert len(loader) == 3                                   # no drop_last: 2 + 2 + 1
    seen = []
    for b in loader:
        assert set(b) == {"user_id", "in_item_id", "item_id", "seqlen", "label", "domain_id", "index"}
        assert torch.equal(b["in_item_id"], f["in_item_id"][b["index"]])
        seen += b["index"].tolist()
    assert sorted(seen) == [0, 1, 2, 3, 4]                    # one epoch = one permutation
    va = SeparateDataset(cfg, "val")
    va.build()
    vb = next(iter(va.get_loader()))
    assert vb["item_id"].shape == (4,) and torch.equal(vb["user_hist"], vb["in_item_id"])
    assert tr[1]["seqlen"] == 3 and tr[1]["index"] == 1      # per-sample access still works


def test_packed_row_cache_roundtrip_and_staleness(tmp_path, monkeypatch):
    """data/packed.py: the packed image reproduces the .pth rows bit for bit, is reused while fresh, rebuilt when the source changes"""
    import time as _time
    from dr4sr_amd.data import packed
    from dr4sr_amd.data.dataset import SeparateDataset
    train, val = _write_dataset(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    cfg = {"data": {"dataset": "tiny", "domain_name_list": ["toy"], "max_seq_len": 50, "train_file": "_ori"},
           "train": {"device": "cpu", "batch_size": 2}, "eval": {"batch_size": 4}}
    d = os.path.join("dataset", "tiny", "toy")
    ref = {}
    for phase in ("train", "val"):                                 # first build: from the pickles (writes the images)
        ds = SeparateDataset(cfg, phase)
        ds.build()
        ref[phase] = {k: v.clone() for k, v in ds.fields().items()}
    pk = os.path.join(d, "train_ori.pth.dr4srpk")
    assert os.path.exists(pk) and os.path.exists(os.path.join(d, "val.pth.dr4srpk"))
    got = packed.read_packed(pk, os.path.join(d, "train_ori.pth"))
    assert got is not None and got["item_id"].shape == (5, 50) and got["seqlen"].dtype == np.int32
    calls = []
    real_load = torch.load
    monkeypatch.setattr(torch, "load", lambda *a, **k: calls.append(a) or real_load(*a, **k))
    for phase in ("train", "val"):                                 # second build: from the images only
        ds = SeparateDataset(cfg, phase)
        ds.build()
        for k, v in ds.fields().items():
            assert v.dtype == torch.int64 and torch.equal(v, ref[phase][k]), (phase, k)
    assert calls == []
    # stale source -> image ignored and rebuilt
    train[0][3] = 1
    train[0][1][0] = 39
    _time.sleep(0.01)
    torch.save(train, os.path.join(d, "train_ori.pth"))
    assert packed.read_packed(pk, os.path.join(d, "train_ori.pth")) is None
    ds = SeparateDataset(cfg, "train")
    ds.build()
    assert len(calls) == 1 and int(ds.fields()["in_item_id"][0, 0]) == 39
    assert packed.read_packed(pk, os.path.join(d, "train_ori.pth")) is not None
    # corrupt / truncated image -> ignored
    with open(pk, "r+b") as f:
        f.truncate(100)
    assert packed.read_packed(pk, os.path.join(d, "train_ori.pth")) is None
    cfg["data"]["packed_cache"] = False                            # opt-out never touches the image
    os.remove(pk)
    SeparateDataset(cfg, "train").build()
    assert not os.path.exists(pk)


def test_metrics_match_golden(golden_dir):
    from dr4sr_amd import evaluation
    z = np.load(os.path.join(golden_dir, "sasrec_d64.npz"))
    hit = torch.from_numpy(z["eval.item_id"]).view(-1, 1) == torch.from_numpy(z["eval.topk_items"])
    tgt = torch.from_numpy(z["eval.label"]).view(-1, 1)
    for k in (20, 10):
        np.testing.assert_allclose(evaluation.ndcg(hit, tgt, k, mean=False).numpy(), z[f"eval.ndcg@{k}"], rtol=1e-6)
        np.testing.assert_allclose(evaluation.recall(hit, tgt, k, mean=False).numpy(), z[f"eval.recall@{k}"], rtol=1e-6)
    assert evaluation.get_eval_metrics(["ndcg", "recall"], [20, 10], validation=True) == ["ndcg@20", "recall@20"]
    assert evaluation.get_eval_metrics(["ndcg", "recall"], [20, 10]) == ["ndcg@20", "recall@20", "ndcg@10", "recall@10"]


def test_early_stopping_and_checkpoint_format(tmp_path, monkeypatch):
    from dr4sr_amd.utils.callbacks import EarlyStopping
    monkeypatch.chdir(tmp_path)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))
            self.config = {"x": 1}
    m = M()
    es = EarlyStopping(m, "ndcg@20", "ds", patience=2)
    assert es(m, 0, {"ndcg@20": 0.1}) is False
    m.w.data.fill_(1.0)
    assert es(m, 1, {"ndcg@20": 0.05}) is False and es(m, 2, {"ndcg@20": 0.02}) is True
    ck = torch.load(os.path.join("saved", es.get_checkpoint_path()), weights_only=False)
    assert set(ck) == {"config", "model", "epoch", "parameters", "metric"} and ck["model"] == "M" and ck["epoch"] == 0
    assert float(ck["parameters"]["w"].sum()) == 0.0           # the BEST state was kept, not the latest


def test_shard_bounds_cover_every_batch_once():
    from dr4sr_amd.parallel import shard_bounds
    n, B = 19412, 256
    for W in (1, 2, 4, 8, 3):
        nb = (n + B - 1) // B
        for i in (0, 1, nb - 1):
            spans = [shard_bounds(i, B, n, W, r) for r in range(W)]
            assert spans[0][0] == i * B and spans[-1][1] == min(n, (i + 1) * B)
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0] and a[0] <= a[1]
    assert shard_bounds(0, 256, 3, 8, 5) == (3, 3)             # more ranks than rows: empty slice


def _dp_worker(rank, world, port, golden_dir, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import sasrec_oracle as O
    from dr4sr_amd.parallel import allreduce_flat, shard_bounds
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    z = np.load(os.path.join(golden_dir, "sasrec_d64.npz"))
    params = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    batch = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch.")}
    B = batch["seqlen"].shape[0]
    lo, hi = shard_bounds(0, B, B, world, rank)
    sl = {k: v[lo:hi] for k, v in batch.items()}
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items() if k != "query_encoder.item_encoder.weight"}
    # un-normalised local gradient: d(sum loss) = d(mean loss * n_local)
    loss, _, _, _ = O.training_step(leaf, sl, 2, 2, 1e-12)
    n_local = (sl["item_id"] != 0).sum()
    (loss * n_local).backward()
    flat = torch.cat([leaf[k].grad.flatten() for k in sorted(leaf)] + [torch.tensor([float(n_local), float(loss * n_local), 0, 0])])
    allreduce_flat(flat)
    if rank == 0:
        q.put(flat.numpy())
    dist.destroy_process_group()


def test_data_parallel_math_gloo_world2(golden_dir):
    """sum-all-reduce of un-normalised shard gradients + {n_valid, loss_sum} tail == the reference's global-batch gradient"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    z = np.load(os.path.join(golden_dir, "sasrec_d64.npz"))
    n, loss_sum = flat[-4], flat[-3]
    assert n == (z["batch.item_id"] != 0).sum()
    np.testing.assert_allclose(loss_sum / n, float(z["out.loss"]), rtol=1e-5)
    keys = sorted(k[5:] for k in z.files if k.startswith("grad."))
    o = 0
    for k in keys:
        ref = z["grad." + k]
        got = flat[o:o + ref.size].reshape(ref.shape) / n
        o += ref.size
        assert np.abs(got - ref).max() <= 2e-4 * max(1e-8, np.abs(ref).max()), k


def _dp_cl_worker(rank, world, port, golden_dir, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import cl4srec_oracle as CO, sasrec_oracle as O
    from dr4sr_amd.parallel import all_gather_flat, allreduce_flat, shard_bounds
    from tests.test_cl_oracle import load_cl
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g, p, batch, views, cfg = load_cl(golden_dir)
    B, D = batch["seqlen"].shape[0], p["item_embedding.weight"].shape[1]
    bounds = [shard_bounds(0, B, B, world, k) for k in range(world)]
    counts = [b - a for a, b in bounds]
    lo, hi = bounds[rank]
    sl = {k: v[lo:hi] for k, v in batch.items()}
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    bce, _, _, _ = O.training_step(leaf, sl, cfg["H"], cfg["n_layer"], cfg["eps"])
    n_local = (sl["item_id"] != 0).sum()
    (vi, li), (vj, lj) = views
    oi = CO.view_mean(leaf, vi[lo:hi], li[lo:hi], cfg["H"], cfg["n_layer"], cfg["eps"])
    oj = CO.view_mean(leaf, vj[lo:hi], lj[lo:hi], cfg["H"], cfg["n_layer"], cfg["eps"])
    # the scheme of CL4SRec._cl_term: [Bmax rows of (q_i | q_j | kept)] + n_valid, all-gathered; own rows keep their graph
    Bmax, RW = max(counts), 2 * D + 1
    send = torch.zeros(Bmax * RW + 1)
    rows = send[:Bmax * RW].view(Bmax, RW)
    rows[:hi - lo, :D], rows[:hi - lo, D:2 * D], rows[:hi - lo, 2 * D] = oi.detach(), oj.detach(), (sl["seqlen"] != 1).float()
    send[-1] = float(n_local)
    got = all_gather_flat(send)
    parts = [got[k, :counts[k] * RW].view(counts[k], RW) for k in range(world)]
    xi = torch.cat([oi if k == rank else parts[k][:, :D] for k in range(world)], 0)
    xj = torch.cat([oj if k == rank else parts[k][:, D:2 * D] for k in range(world)], 0)
    keep = torch.cat([pt[:, 2 * D] for pt in parts], 0) > 0
    nv_glob = got[:, -1].sum()
    cl = CO.infonce(xi[keep], xj[keep], cfg["temperature"])
    # un-normalised local objective: sum of the local BCE terms + the contrastive mean under the optimizer's 1 / n_valid(global)
    (bce * n_local + cfg["cl_weight"] * nv_glob * cl).backward()
    names = sorted(leaf)
    flat = torch.cat([(leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k])).flatten() for k in names]
                     + [torch.tensor([float(n_local), float(bce * n_local + cfg["cl_weight"] * cl * n_local), 0, 0])])
    allreduce_flat(flat)
    if rank == 0:
        q.put((names, flat.numpy()))
    dist.destroy_process_group()


def test_cl4srec_data_parallel_scheme_gloo_world2(golden_dir):
    """CL4SRec under DP (round 4): InfoNCE's negatives are the other rows of the GLOBAL batch.  Every rank all-gathers the pooled views
    (parallel.all_gather_flat), evaluates the global loss with the other ranks' rows as constants and back-propagates its own rows; the
    sum-all-reduce of those local gradients (scaled cl_weight * n_valid(global), as the optimizer divides by the all-reduced n_valid)
    == the reference's single-process gradient of the whole batch (golden vectors of tests/golden/cl4srec_d64.npz)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_cl_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    names, flat = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    z = np.load(os.path.join(golden_dir, "cl4srec_d64.npz"))
    n = flat[-4]
    assert n == (z["batch.item_id"] != 0).sum()
    np.testing.assert_allclose(flat[-3] / n, float(z["out.loss"]), rtol=1e-5)
    o = 0
    for k in names:
        ref = z["grad." + k]
        got = flat[o:o + ref.size].reshape(ref.shape) / n
        o += ref.size
        assert np.abs(got - ref).max() <= 2e-4 * max(1e-8, np.abs(ref).max()), k


def test_torch_ops_are_registered_with_schemas_and_fake_kernels():
    """dr4sr_amd/ops.py: the dense C-ABI entry points are dispatcher ops (torch.ops.dr4sr_hip.*) with schemas and shape-inference
    (fake) kernels — checkable without a GPU; the real kernels refuse CPU tensors (no CPU path)"""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    import dr4sr_amd.ops as ops
    from dr4sr_amd._lib import Dr4srError
    for n in ops.OPS:
        assert hasattr(torch.ops.dr4sr_hip, n), n
    assert "Tensor(a0!) params" in str(torch.ops.dr4sr_hip.fused_adam_.default._schema)
    with FakeTensorMode():
        E, P = torch.empty(100, 64), torch.empty(50, 64)
        idx = torch.empty(8, 50, dtype=torch.int64)
        assert torch.ops.dr4sr_hip.embed_gather_posadd(E, P, idx).shape == (8, 50, 64)
        lp, st = torch.ops.dr4sr_hip.score_bce(torch.empty(8, 50, 64), E, idx, idx)
        assert lp.shape == (8, 50) and st.shape == (2,)
        sc, it = torch.ops.dr4sr_hip.full_score_topk(torch.empty(8, 64), E, idx, None, 20)
        assert sc.shape == (8, 20) and it.dtype == torch.int64
    with pytest.raises(Dr4srError):
        torch.ops.dr4sr_hip.embed_gather_posadd(torch.zeros(10, 64), torch.zeros(50, 64), torch.zeros(2, 50, dtype=torch.int64))


class _FakeEngine:
    """the slice of the engine protocol parallel.dp_backward drives, on CPU tensors: the `gradient` of a rank is a deterministic function of
    its slice of the global batch, phase 1 fills the table bucket, phase 2 the encoder bucket + tail (a short / empty-regime plan fills
    everything in phase 1, like a latency-form plan under a two-bucket decision)"""
    def __init__(self, n, split, two_phase_plan=True):
        self.grads = torch.zeros(n + 4)
        self.n, self.split, self.two = n, split, two_phase_plan
        self.calls = []

    def _contrib(self, plan, lo, hi):
        g = torch.Generator().manual_seed(1234)
        base = torch.randn(self.n + 4, generator=g)
        for r in plan:                                       # plan = the global row ids of this rank's slice
            self.grads[lo:hi] += base[lo:hi] * float(r + 1)

    def fwd_bwd(self, plan):
        self.calls.append("flat")
        self.grads.zero_()
        self._contrib(plan, 0, self.n + 4)
    fwd_bwd_prepared = fwd_bwd

    def fwd_bwd_phase(self, plan, prepared, phase):
        self.calls.append("p%d" % phase)
        if phase == 1:
            self.grads.zero_()
            self._contrib(plan, 0, self.split if self.two else self.n + 4)
        elif self.two:
            self._contrib(plan, self.split, self.n + 4)


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from dr4sr_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n, split, B, U = 1000, 700, 20, 43                       # global batches of 20 rows over 43 rows: tail batch of 3 -> empty slices at 8 ranks
    res = []
    for i in range(3):
        lo, hi = parallel.shard_bounds(i, B, U, world, rank)
        plan = list(range(lo, hi))
        full = (i + 1) * B <= U
        buckets = [(0, split), (split, n + 4)] if full else None       # the decision is global: flat on the partial tail batch
        # rank 1's own plan is a "latency form" plan (everything final after phase 1) although the global decision is two buckets
        eng = _FakeEngine(n, split, two_phase_plan=(rank != 1))
        if plan:
            parallel.dp_backward(eng, plan, False, buckets)
        else:                                                # (at 8 ranks also on the FULL batches: 20 rows = 3 x 6 + 2 + 0)
            parallel.dp_reduce_empty(eng, buckets)
        flat = _FakeEngine(n, split)
        flat.fwd_bwd(plan)
        parallel.allreduce_flat(flat.grads)
        res.append((eng.grads.clone(), flat.grads.clone(), list(eng.calls), len(plan)))
    if rank == 0:
        q.put([(a.numpy(), b.numpy(), c, d) for a, b, c, d in res])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bucketed_allreduce_equals_flat_gloo(world):
    """parallel.dp_backward: phase 1 -> async all-reduce(table bucket) -> phase 2 -> async all-reduce(encoder bucket + tail) -> join
    == one flat all-reduce == the single-rank gradient, with 2 and 8 gloo ranks, a rank whose own plan has one phase, and a tail batch
    that leaves ranks empty (they contribute zeros to a flat all-reduce — every rank enters the same collectives)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for i, (bucketed, flat, calls, nrows) in enumerate(res):
        np.testing.assert_allclose(bucketed, flat, rtol=1e-6, atol=1e-6)
        one = _FakeEngine(1000, 700)
        one.fwd_bwd(list(range(i * 20, min(43, (i + 1) * 20))))         # the single-rank gradient of the whole global batch
        np.testing.assert_allclose(bucketed, one.grads.numpy(), rtol=1e-5, atol=1e-4)
        assert calls == ((["p1", "p2"] if i < 2 else ["flat"]) if nrows else [])


def test_pmc_traffic_matcher_names_exactly_one_kernel_per_regime():
    """bench.py's roofline.traffic summed k_post_mid + k_wt_post_mid (16.64 + 10.03 MB) because the committed round-5
    PMC digest holds both regimes' kernels; tools/pmc_match.py picks the regime's ONE kernel.  Fed with the committed digests."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_match
    pm = json.load(open(os.path.join(ROOT, "profiles", "round5_pmc_traffic_B256_toys.json")))
    b, name = pmc_match.match(pm, "post_mid", False)
    assert name.startswith("k_post_mid<") and abs(b - 16640816) < 1
    b, name = pmc_match.match(pm, "post_mid", True)
    assert name.startswith("k_wt_post_mid<") and abs(b - 10029921) < 1
    b, name = pmc_match.match(pm, "wgrad_fused", False)
    assert name.startswith("k_wgrad_blk<")                                     # not k_wgrad_det_reduce, not k_wgrad_bf64
    assert pmc_match.match(pm, "wgrad_fused", True)[1].startswith("k_wgrad_bf64<")
    assert pmc_match.match(pm, "qkv_embed_bwd", False) == (None, None)         # the latency regime has no such launch
    big = json.load(open(os.path.join(ROOT, "profiles", "round5_pmc_traffic_B8192_toys.json")))
    assert abs(pmc_match.match(big, "wgrad_fused", True)[0] - 420645591) < 1 and pmc_match.match(big, "post_mid", False) == (None, None)
    assert pmc_match.match_attention(big, True) == 16896775 + 20876035 + 46685611
    assert pmc_match.stem("void tiny::k_attn_tiny_bwd<32>(Args)") == "k_attn_tiny_bwd"
    with pytest.raises(ValueError):
        pmc_match.match({"k_post_mid<16, 64, 128, false>": {"hbm_bytes_per_launch": 1}, "k_post_mid<16, 64, 128, true>": {"hbm_bytes_per_launch": 2}},
                        "post_mid", False)


def test_switch_inventory_is_generated_from_the_source_and_small():
    """VERDICT r5 weak #10 / Next #6: the shipped library reads at most 35 run-time switches (the experiment / tuning ones are compile-time
    constants: csrc/common.h DR4SR_XENV), and SWITCHES.md is exactly what tools/gen_switches.py derives from the source"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_switches.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    n_run = int(re.search(r"(\d+) run-time", out.stdout).group(1))
    assert n_run <= 35, out.stdout
    from dr4sr_amd import _lib
    assert _lib.load().dr4sr_build_flags() in (0, 1)


def _control_plane_worker(rank, world, port, q):
    import torch.distributed as dist
    from dr4sr_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), DR4SR_DP_BACKEND="gloo")
    assert parallel.init_distributed() and dist.get_backend() == "gloo" and not parallel.can_capture()
    ok_all = parallel.all_ok(True)
    ok_one_bad = parallel.all_ok(rank != 1)
    mx = parallel.host_allreduce([float(rank), 10.0 - rank], "max")
    sm = parallel.host_allreduce([1.0], "sum")
    ga = parallel.host_allgather(100.0 + rank)
    t = torch.full((5,), float(rank + 1))
    parallel.allreduce_flat(t)
    h = parallel.allreduce_begin(t)
    parallel.allreduce_end(h)
    g = parallel.all_gather_flat(torch.tensor([rank, rank * 2], dtype=torch.int64))
    b = parallel.broadcast(torch.tensor([float(rank)]), src=1)
    parallel.barrier()
    if rank == 0:
        q.put((ok_all, ok_one_bad, mx, sm, ga, t.tolist(), g.tolist(), b.tolist()))
    parallel.shutdown()
    parallel.shutdown()                                    # idempotent


def test_control_plane_helpers_gloo_world3():
    """round 6: torch.distributed carries only the CONTROL plane (a CPU gloo group): decisions every rank must take alike, timings, barriers,
    and — with DR4SR_DP_BACKEND=gloo — the staged data plane of the functional tests.  World size 3 on CPU."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    W = 3
    procs = [ctx.Process(target=_control_plane_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    ok_all, ok_one_bad, mx, sm, ga, t, g, b = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok_all is True and ok_one_bad is False
    assert mx == [2.0, 10.0] and sm == [3.0] and ga == [100.0, 101.0, 102.0]
    assert t == [18.0] * 5                                  # 1 + 2 + 3 = 6 on every rank, then the begin / end form reduces that again: 18
    assert g == [[0, 0], [1, 2], [2, 4]] and b == [1.0]
