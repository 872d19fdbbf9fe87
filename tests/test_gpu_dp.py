"""Round-5 data-parallel tests (VERDICT r4 "Next round" item 1): the two-bucket step and the rank counts that will really run.

  * dr4sr_sasrec_fwd_bwd_phase (ABI 7): phase 1 + phase 2 leave exactly what dr4sr_sasrec_fwd_bwd leaves — the item-table bucket
    bit for bit (owner-computed rows: no atomics), the encoder bucket to fp32 summation order; one bucket in the latency forms;
  * bucketed == flat == single rank over the PRODUCT's reduction path (parallel.grad_buckets / dp_backward) with 2 and 8 gloo ranks
    sharing the GPU: 32 rows per rank at B = 256 (d = 64 and BASELINE configs[3]'s d = 128), a 212-row tail batch whose slices are uneven
    and EMPTY, and at-scale per-rank batches where the step has two buckets (tools/dp_check.py);
  * the two-bucket step with its two RCCL all-reduces captured inside a k-step graph (one RCCL rank — all a 1-GPU box can host): the
    table bucket's collective is a parallel branch of the graph beside the last weight-gradient launch (tools/dp_graph_check.py);
  * MetaModel's outer step on 8 ranks (12 and 11-or-12 rows per rank) == single rank (tools/dp_meta_check.py);
  * `python bench.py --gpus 8` on a shared GPU (gloo).

The reference has no distributed path (/root/reference/utils/callbacks.py:130 is its TODO); the contract is SURVEY.md section 8(e)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(B, D=64, seed=3, NL=2):
    from test_gpu_parity import _random_params
    from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
    from dr4sr_amd.engine import SasrecEngine
    dev = torch.device("cuda", 0)
    L, N = 50, TOYS_N_ITEMS
    rows = make_rows(n_rows=B, n_items=N, seed=17)
    data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
    negs = torch.randint(1, N, (B, L), generator=torch.Generator().manual_seed(8)).to(dev)
    eng = SasrecEngine(N, L, D, 2, 128, NL, 1e-12, 0.3, B, dev, seed=seed, lr=1e-3)
    eng.load_named(_random_params(N, D, 128, NL, seed=6))
    plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], neg_item=negs, sample_neg=False)
    return eng, plan, data


def _grads_of(eng, run):
    eng.state[3] = 7                                        # same dropout stream for every form (state[RNGSTEP] is bumped by the prep)
    run()
    torch.cuda.synchronize()
    return eng.grads.clone()


@pytest.mark.parametrize("split", [None, "2", "1"])
def test_two_phase_step_equals_one_launch_step_at_scale(monkeypatch, split):
    """B = 4 096 toys-shaped rows (22 k tokens: the at-scale forms): two buckets, bounds = [0, offsets[2]) | [offsets[2], n + tail);
    phase 1 + phase 2 == dr4sr_sasrec_fwd_bwd.  split: DR4SR_DP_SPLIT_LAYER (None = default 1: table jobs + layer 1's weight gradients
    first; 2 = a table-only first launch)"""
    from dr4sr_amd import _lib
    if split is not None:
        monkeypatch.setenv("DR4SR_DP_SPLIT_LAYER", split)
    eng, plan, _ = _engine(4096)
    assert int(eng.lib.dr4sr_sasrec_at_scale(_lib.C.byref(plan))) & 1
    b = eng.grad_buckets(plan)
    n = eng.n_params
    assert b == [(0, eng.offsets[2]), (eng.offsets[2], n + _lib.GRAD_TAIL)], b
    g_one = _grads_of(eng, lambda: eng.fwd_bwd(plan))
    # the caller's view between the phases: the table bucket is FINAL after phase 1 (that is what lets its all-reduce start there)
    mid = {}

    def two():
        eng.fwd_bwd_phase(plan, False, 1)
        torch.cuda.synchronize()
        mid["table"] = eng.grads[:b[0][1]].clone()
        mid["tail"] = eng.grads[n:n + 2].clone()
        eng.fwd_bwd_phase(plan, False, 2)
    g_two = _grads_of(eng, two)
    assert torch.equal(mid["table"], g_two[:b[0][1]])                        # nothing touches the table bucket after phase 1
    assert torch.equal(g_one[:eng.offsets[1]], g_two[:eng.offsets[1]])      # item table: owner-computed, bit-reproducible
    assert float(mid["tail"].abs().max()) == 0.0 and float(g_two[n]) > 0     # {n_valid, loss_sum} arrive with the encoder bucket
    scale = float(g_one.abs().max())
    assert float((g_one - g_two).abs().max()) <= 1e-6 * scale, float((g_one - g_two).abs().max()) / scale
    assert float(g_one[n]) == float(g_two[n]) and abs(float(g_one[n + 1]) - float(g_two[n + 1])) <= 1e-6 * abs(float(g_one[n + 1]))


@pytest.mark.parametrize("NL", [1, 3])
def test_two_phase_step_other_layer_counts(NL):
    """the launch cut follows the layer count (default split layer max(1, n_layer / 2)): one layer — the table jobs alone in phase 1;
    three layers — the table jobs + the two upper layers' weight gradients; both == the one-launch step"""
    from dr4sr_amd import _lib
    eng, plan, _ = _engine(4096, NL=NL)
    assert int(eng.lib.dr4sr_sasrec_at_scale(_lib.C.byref(plan))) & 1
    b, n = eng.grad_buckets(plan), eng.n_params
    assert b == [(0, eng.offsets[2]), (eng.offsets[2], n + _lib.GRAD_TAIL)], b
    g_one = _grads_of(eng, lambda: eng.fwd_bwd(plan))
    mid = {}

    def two():
        eng.fwd_bwd_phase(plan, False, 1)
        torch.cuda.synchronize()
        mid["table"] = eng.grads[:b[0][1]].clone()
        eng.fwd_bwd_phase(plan, False, 2)
    g_two = _grads_of(eng, two)
    assert torch.equal(mid["table"], g_two[:b[0][1]]) and torch.equal(g_one[:eng.offsets[1]], g_two[:eng.offsets[1]])
    scale = float(g_one.abs().max())
    assert float((g_one - g_two).abs().max()) <= 1e-6 * scale, float((g_one - g_two).abs().max()) / scale
    assert float(g_one[n]) == float(g_two[n]) > 0


def test_two_phase_step_is_the_whole_step_in_the_latency_forms():
    """B = 256: one bucket; phase 1 runs the whole step, phase 2 nothing — what a rank with a short slice does under a two-bucket decision
    taken from the full slice size (parallel.grad_buckets: the count must be equal on every rank)"""
    from dr4sr_amd import _lib
    eng, plan, data = _engine(256)
    n = eng.n_params
    assert eng.grad_buckets(plan) == [(0, n + _lib.GRAD_TAIL)]
    g_one = _grads_of(eng, lambda: eng.fwd_bwd(plan))
    g_p1 = _grads_of(eng, lambda: eng.fwd_bwd_phase(plan, False, 1))
    before = g_p1.clone()
    eng.fwd_bwd_phase(plan, False, 2)
    torch.cuda.synchronize()
    assert torch.equal(before, eng.grads)
    scale = float(g_one.abs().max())
    assert float((g_one - g_p1).abs().max()) <= 2e-5 * scale                 # fp32 atomics order (table gradient of the latency forms)
    # the probe the model decides with: a full slice of 4 096 rows of this dataset -> two buckets, whatever this plan's own size
    from dr4sr_amd import parallel
    assert len(parallel.grad_buckets(eng, 4096, data["seqlen"], want=2)) == 2 and len(parallel.grad_buckets(eng, 256, data["seqlen"], want=2)) == 1
    assert len(parallel.grad_buckets(eng, 4096, data["seqlen"])) == 1                # flat unless asked (train.dp_buckets / DR4SR_DP_BUCKETS)
    assert len(parallel.grad_buckets(eng, None, data["seqlen"])) == 1


from _launch import report, torchrun


@pytest.mark.parametrize("W,D,B,U,expect", [
    (8, 64, 256, 980, "[1]"),            # 32 rows per rank; tail batch 212 rows -> slices 32 x 6, 20, EMPTY
    (8, 128, 256, 980, "[1]"),           # BASELINE configs[3]'s width
    (2, 64, 6144, 13288, "[1, 2]"),      # 3 072 rows per rank: at scale -> two buckets on the full batches, flat on the 1 000-row tail
    (8, 64, 24576, 27576, "[1, 2]"),     # 8 ranks x 3 072 rows, two buckets; tail 3 000 rows -> rank 0 only, seven EMPTY slices
])
def test_data_parallel_ranks_equal_single_rank_wide(W, D, B, U, expect):
    out = torchrun(W, "tools/dp_check.py", {"DP_D": D, "DP_B": B, "DP_U": U, "DR4SR_DP_BACKEND": "gloo", "DR4SR_DP_BUCKETS": 2})
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_CHECK")]
    assert out.returncode == 0 and len(lines) == 2, report(out)
    assert "buckets=" + expect in lines[0], lines[0]
    assert "replica checksums equal: True" in lines[1]
    if W == 8:
        assert "empty slices seen: 0" not in lines[1], lines[1]
    print(lines[0])


@pytest.mark.parametrize("B,buckets", [(8192, 2), (1280, 1)])
def test_rccl_buckets_inside_k_step_graph(B, buckets):
    """one RCCL rank: k DP steps per graph, each with its collective(s) captured — at B = 8 192 the table bucket's all-reduce is issued
    asynchronously after phase 1 and joined before the optimizer (a parallel branch of the graph), the optimizer launches prepare the
    next batch in two phases — against the un-captured single-GPU loop; B = 1 280 (7.0 k tokens: latency forms — below round 6's 7.7 k boundary — above the
    old 1 024-row limit of the prepared form): one flat bucket"""
    out = torchrun(1, "tools/dp_graph_check.py", {"DP_GRAPH_B": B, "DP_GRAPH_REPLAYS": 6, "DP_GRAPH_K": 3, "DP_GRAPH_EXPECT_BUCKETS": buckets,
                                                   "DR4SR_DP_BUCKETS": 2})
    assert out.returncode == 0 and "DP_GRAPH_OK" in out.stdout, report(out)
    print([l for l in out.stdout.splitlines() if l.startswith("DP_GRAPH ")][0])


@pytest.mark.parametrize("B", [96, 90])
def test_metamodel_outer_step_eight_ranks_equal_single_rank(B):
    """BASELINE configs[4] is an 8-GPU configuration: the outer step's all-reduces on 8 ranks (12 rows per rank; B = 90: 12 x 7 + 6)"""
    out = torchrun(8, "tools/dp_meta_check.py", {"DR4SR_DP_BACKEND": "gloo", "DP_META_B": B})
    assert out.returncode == 0, report(out)
    line = [l for l in out.stdout.splitlines() if l.startswith("DP_META")]
    assert line and "world=8" in line[0] and "replicas identical: True" in line[0], out.stdout[-2000:]
    print(line[0])


@pytest.mark.parametrize("model_name", ["SASRec-d128", "MetaModel"])
def test_data_parallel_fit_eight_ranks(tmp_path, model_name):
    """quickstart.run under 8 ranks sharing the GPU (gloo): batch 128 -> 16 rows per rank, the 13-row tail batch leaves seven ranks
    EMPTY; replicas bit-identical, one checkpoint stem (tools/dp_fit_check.py)"""
    extra = {}
    if model_name == "SASRec-d128":
        model_name, extra = "SASRec", {"DR4SR_EMBED_DIM": "128"}
    out = torchrun(8, "tools/dp_fit_check.py", dict(DR4SR_DP_BACKEND="gloo", DP_FIT_DIR=str(tmp_path), MODEL=model_name, **extra))
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_FIT ")]
    err = out.stdout[out.stdout.find("DP_FIT_ERROR"):][:3000] if "DP_FIT_ERROR" in out.stdout else report(out)
    assert out.returncode == 0 and len(lines) == 1, err
    assert "world=8" in lines[0] and "replicas identical: True; finite: True" in lines[0] and "one ckpt stem: True" in lines[0], lines[0]


@pytest.mark.parametrize("model_name,loss,W", [("SASRec", "bpr", 2), ("SASRec", "bce", 3), ("GRU4Rec", "bpr", 2), ("FMLP", "bpr", 2)])
def test_model_api_loop_under_data_parallelism_equals_one_rank(model_name, loss, W):
    """round 6: the model-API training loop (loss_fn 'bpr', or DR4SR_NO_FAST_PATH with 'bce' — it raised under W > 1 before) sliced over W ranks
    (BaseModel._api_epoch_dp: local mean loss scaled by n_valid_local / n_valid_global, SUM all-reduce of the flat gradient, a rank with an EMPTY
    slice of the tail batch still in both collectives) moves the parameters exactly as the same loop on one rank does (tools/dp_api_check.py:
    dropout 0, negatives a function of the targets): replicas bit-identical, parameters and per-step losses within fp32 summation order"""
    out = torchrun(W, "tools/dp_api_check.py", dict(DR4SR_DP_BACKEND="gloo", MODEL=model_name, LOSS_FN=loss))
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_API")]
    assert out.returncode == 0 and len(lines) == 2 and lines[1] == "DP_API_OK", report(out)
    assert "world=%d" % W in lines[0] and "replicas identical: True" in lines[0], lines[0]


def test_data_parallel_fit_with_the_model_api_loop(tmp_path):
    """quickstart.run (fit + evaluate) of SASRec with loss_fn 'bpr' under two ranks: the model-API loop end to end (tools/dp_fit_check.py)"""
    out = torchrun(2, "tools/dp_fit_check.py", dict(DR4SR_DP_BACKEND="gloo", DP_FIT_DIR=str(tmp_path), MODEL="SASRec", LOSS_FN="bpr"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_FIT ")]
    err = out.stdout[out.stdout.find("DP_FIT_ERROR"):][:3000] if "DP_FIT_ERROR" in out.stdout else report(out)
    assert out.returncode == 0 and len(lines) == 1, err
    assert "world=2" in lines[0] and "replicas identical: True; finite: True" in lines[0] and "one ckpt stem: True" in lines[0], lines[0]


def test_bench_self_launches_eight_ranks_on_a_shared_gpu():
    """`python bench.py --gpus 8` (the driver's widest command) on ONE GPU over gloo: every rank takes the data-parallel step, the
    strong-scaling entry splits 16 384 rows into 2 048 per rank; one JSON line from rank 0"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--repeats", "2",
                          "--no-throughput-mode", "--strong-global-batch", "16384"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, DR4SR_BENCH_SHARE_GPU="1", DR4SR_DP_BACKEND="gloo"), cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 2048 and j["config"]["parallelism"] == "dp8"
    assert "gloo" in j["config"]["collective"] and j["value"] > 0 and j["final_loss"] == j["final_loss"]
    st = j["strong"][0]
    assert st["global_batch"] == 16384 and st["per_gpu_batch"] == 2048 and st["n_gpus"] == 8 and st["speedup"] > 0


# ------------------------------------------------------------------------------------------------ d = 128 at scale
def test_d128_at_scale_bf16x3_tile_gemms_vs_oracle_and_fp32(monkeypatch):
    """BASELINE configs[3]'s width in the at-scale launch forms: the 256-thread tile kernels' GEMMs run as a 3-term bf16 split from
    fragment-major weight images (k_wsplit, csrc/common.h tile_mma_xwT_bf3).  B = 1 024 toys-shaped rows of the yelp-sized table
    against the ORACLE's autograd (loss, every gradient) and against the fp32 MFMA kernels of the same step (DR4SR_TILE_F32=1);
    reference arithmetic: /root/reference/model/sasrec.py:39-75 + model/basemodel.py:204-214 + model/loss_func.py:9-38"""
    from oracle import sasrec_oracle as O
    from test_gpu_parity import _random_params, _toys_batch, relerr
    from dr4sr_amd import _lib
    from dr4sr_amd.engine import SasrecEngine
    dev = torch.device("cuda", 0)
    B, D, N = 1024, 128, 20034
    b, N = _toys_batch(B, False, seed=11, n_items=N)
    params = _random_params(N, D, 128, 2, seed=4)
    out = {}
    for f32 in (False, True):
        if f32:
            monkeypatch.setenv("DR4SR_TILE_F32", "1")
        eng = SasrecEngine(N, 50, D, 2, 128, 2, 1e-12, 0.0, B, dev)
        eng.load_named(params)
        plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                             neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
        assert int(eng.lib.dr4sr_sasrec_at_scale(_lib.C.byref(plan))) & 1          # 5.6 k tokens x 128 columns: the at-scale forms
        eng.fwd_bwd(plan)
        out[f32] = (eng.loss_and_count(), {k: v.clone() for k, v in eng.normalized_grads().items()})
    (loss, n), grads = out[False]
    loss_o, _, grads_o = O.grads_of(params, b, 2, 2, 1e-12)
    assert n == int((b["item_id"] != 0).sum()) and abs(loss - float(loss_o)) < 2e-5
    worst = 0.0
    for k, gv in grads.items():
        e = relerr(gv, grads_o[k])
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
        assert relerr(gv, out[True][1][k].cpu()) < 5e-5, k                          # split vs fp32 MFMA: 5e-6 per product, two layers deep
    assert abs(loss - out[True][0][0]) < 1e-5
    print("d = 128 at scale, bf16x3 tile GEMMs: worst gradient error vs oracle %.2e" % worst)


def test_d128_latency_forms_fragment_images_fused_and_unfused(monkeypatch):
    """d = 128, B = 256 (BASELINE configs[3]'s per-GPU batch): the forward GEMMs of k_post_fwd / k_post_mid read their weight fragments from the
    fragment-major fp32 image written by the step's first launch — k_embqkv_fwd<16, 128> in the fused step, k_wfrag_write under DR4SR_NO_FUSE
    (csrc/linear.hip wfrag_image_write).  Both against the ORACLE, and after the parameters CHANGED between two steps (a stale image would
    reproduce the first step's activations)."""
    from oracle import sasrec_oracle as O
    from test_gpu_parity import _random_params, _toys_batch, relerr
    from dr4sr_amd.engine import SasrecEngine
    dev = torch.device("cuda", 0)
    B, D, N = 256, 128, 20034
    b, N = _toys_batch(B, False, seed=13, n_items=N)
    for nofuse in (False, True):
        if nofuse:
            monkeypatch.setenv("DR4SR_NO_FUSE", "1")
        params = _random_params(N, D, 128, 2, seed=9)
        eng = SasrecEngine(N, 50, D, 2, 128, 2, 1e-12, 0.0, B, dev, lr=1e-2)
        eng.load_named(params)
        plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                             neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
        eng.train_step(plan)                               # step 1 moves every weight by ~lr
        eng.fwd_bwd(plan)                                  # step 2's gradients on the UPDATED weights
        torch.cuda.synchronize()
        p2 = {k: v.detach().cpu().clone() for k, v in eng.views.items()}
        p2["query_encoder.item_encoder.weight"] = p2["item_embedding.weight"]
        loss_o, _, grads_o = O.grads_of(p2, b, 2, 2, 1e-12)
        loss, n = eng.loss_and_count()
        assert abs(loss - float(loss_o)) < 5e-5, (nofuse, loss, float(loss_o))
        for k, gv in eng.normalized_grads().items():
            assert relerr(gv, grads_o[k]) < 2e-4, (nofuse, k)
