"""Data parallelism of the model-API training loop (BaseModel._api_epoch_dp: loss_fn 'bpr' or DR4SR_NO_FAST_PATH — the loop of the reference's
basemodel.py:192-200, one autograd step per batch) against the SAME loop on one rank: W ranks share cuda:0 over the gloo transport, dropout 0,
negatives a fixed function of the targets (so that both runs see the same negatives whatever the slicing), one epoch with a short tail batch;
then every rank repeats the epoch alone (world_size 1 semantics, same permutation) from the same initial parameters and compares.
  DR4SR_DP_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 tools/dp_api_check.py"""
import os, sys, faulthandler
faulthandler.dump_traceback_later(240, exit=True)
os.environ.setdefault("DR4SR_DP_BACKEND", "gloo")
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
from dr4sr_amd.parallel import init_distributed
from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
MODEL, LOSS = os.environ.get("MODEL", "SASRec"), os.environ.get("LOSS_FN", "bpr")
cfg = load_config({"model": MODEL, "dataset": "synthetic-toys"})
cfg["data"].update({"n_items": 300, "n_rows": 3 * 96 + 2 * world - 1, "n_eval_rows": 64, "seed": 5})     # tail batch: fewer rows than 2 per rank -> one EMPTY slice
if MODEL == "FMLP":
    cfg["data"]["prefix_rows"] = True
cfg["model"].update({"dropout_rate": 0.0, "loss_fn": LOSS})
cfg["train"].update({"batch_size": 96, "epochs": 1, "device": "cuda:0"})
if LOSS == "bce":
    os.environ["DR4SR_NO_FAST_PATH"] = "1"
torch.cuda.set_device(0)
init_distributed("cuda:0")


def build():
    seed_everything(cfg["train"]["seed"])
    ds = prepare_datasets(cfg)
    m = prepare_model(cfg, ds)
    m._init_model(ds[0])
    m.train()
    m.engine.p_drop = 0.0                                   # (FMLP hard-codes 0.5 like the reference, model/fmlp.py:13: dropout masks are indexed by the row's slot in ITS batch)
    N = m.num_items
    m._neg_sampling = lambda batch: ((batch[m.fiid] * 7 + 3) % (N - 1) + 1).unsqueeze(-1)      # same negatives whatever the slicing
    return m


m = build()
assert not m._fast_path_ok() and m.world_size == world
p0 = m.engine.params.clone()
seed_everything(11)
out_dp = m.training_epoch(0)[0]
torch.cuda.synchronize()
p_dp = m.engine.params.clone()
losses_dp = torch.stack([o["loss_0"].float().reshape(()) for o in out_dp]).cpu()

s = build()
s.world_size, s.rank = 1, 0                                  # the same loop alone
assert torch.equal(s.engine.params, p0)
seed_everything(11)
out_1 = s.training_epoch(0)[0]
torch.cuda.synchronize()
p_1 = s.engine.params.clone()
losses_1 = torch.stack([o["loss_0"].float().reshape(()) for o in out_1]).cpu()

chk = torch.tensor([float(p_dp.double().sum()), float(p_dp.double().abs().sum())], dtype=torch.float64)
lst = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(lst, chk)
if rank == 0 and os.environ.get("DP_API_DEBUG"):
    print("losses dp", losses_dp.tolist(), "single", losses_1.tolist(), flush=True)
if rank == 0:
    same = all(bool((x == lst[0]).all()) for x in lst)
    dpar = float((p_dp - p_1).abs().max())
    dl = float((losses_dp - losses_1).abs().max())
    print("DP_API model=%s loss=%s world=%d steps=%d (tail batch %d rows): replicas identical: %s; max|dp - single| params %.3e (moved %.3e), losses %.3e; loss %.4f -> %.4f"
          % (MODEL, LOSS, world, len(out_dp), cfg["data"]["n_rows"] % 96, same, dpar, float((p_dp - p0).abs().max()), dl, float(losses_dp[0]), float(losses_dp[-1])), flush=True)
    assert same and len(out_dp) == len(out_1) == 4 and dpar < 2e-5 and dl < 2e-5 and float((p_dp - p0).abs().max()) > 1e-3
    print("DP_API_OK", flush=True)
dist.destroy_process_group()
