"""CL4SRec under data parallelism on ONE GPU: W ranks (gloo transport of dr4sr_amd/parallel.py, all on cuda:0) each train their slice of
every global batch through CL4SRec._dp_step — fused main pass + the views' encoder passes on the local rows, the contrastive term over
the all-gathered GLOBAL batch (InfoNCE's negatives are the batch), one sum-all-reduce, dense Adam — and rank 0 also trains a
single-process model on the concatenated batches with the same negatives and the same views.  Prints the max parameter difference
(fp32 summation order only).  The last batch is a ragged tail (one rank short or EMPTY: it still takes part in the gather).
  DR4SR_DP_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 tools/dp_cl_check.py"""
import os, sys, faulthandler
faulthandler.dump_traceback_later(150, exit=True)
os.environ.setdefault("DR4SR_DP_BACKEND", "gloo")
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
from dr4sr_amd.parallel import init_distributed, shard_bounds
from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
from dr4sr_amd.module.data_augmentation import Item_Random

TAIL = int(os.environ.get("DP_CL_TAIL", "20"))           # rows of the last global batch (20 with W = 2, B = 64: slices of 20 and 0)
B, STEPS = 64, 4
cfg = load_config({"model": "CL4SRec", "dataset": "synthetic-toys"})
cfg["data"].update({"n_items": 300, "n_rows": (STEPS - 1) * B + TAIL, "n_eval_rows": 64, "seed": 5})
cfg["model"]["dropout_rate"] = 0.0
cfg["train"].update({"batch_size": B, "epochs": 1, "device": "cuda:0", "hip_graph": False})
torch.cuda.set_device(0)
init_distributed("cuda:0")
dev = torch.device("cuda", 0)


def build():
    seed_everything(cfg["train"]["seed"])
    ds = prepare_datasets(cfg)
    m = prepare_model(cfg, ds)
    m._init_model(ds[0])
    m.train()
    return ds, m


ds, model = build()
loader = ds[0].get_loader()
F = loader.fields
n = loader.n
for r_ in (3, 70, n - 2):                                # rows of length 1: dropped from InfoNCE's rows AND columns (data_augmentation.py:613-615)
    F["seqlen"][r_] = 1
    F["in_item_id"][r_, 1:] = 0
    F["item_id"][r_, 1:] = 0
g = torch.Generator().manual_seed(17)
perm = torch.randperm(n, generator=g).to(dev)
negs = torch.randint(1, model.num_items, (n, model.max_seq_len, 1), generator=g).to(dev)
aug = Item_Random(mask_id=model.num_items, seed=99)      # one fixed draw of both views of EVERY row, identical on every rank
(vi, li), (vj, lj) = aug.two_views(F["in_item_id"], F["seqlen"])


def batch_of(rows):
    b = {k: F[k].index_select(0, rows) for k in ("in_item_id", "item_id", "seqlen", "user_id")}
    b["neg_item"] = negs.index_select(0, rows)
    b["_views"] = ((vi.index_select(0, rows), li.index_select(0, rows)), (vj.index_select(0, rows), lj.index_select(0, rows)))
    return b


import traceback


def _report(e):                                          # the launcher's error page hides the child's traceback: say it on stdout
    print("DP_CL_ERROR rank %d: %s\n%s" % (rank, repr(e), "".join(traceback.format_exc().splitlines(True)[-14:])), flush=True)


try:
  losses = []
  for i in range(STEPS):
      bounds = [shard_bounds(i, B, n, world, k) for k in range(world)]
      lo, hi = bounds[rank]
      model._dp_step(batch_of(perm[lo:hi]), [b - a for a, b in bounds])
      e = model.engine
      losses.append(float(e.grads[e.n_params + 1] / e.grads[e.n_params]))
  torch.cuda.synchronize()
  p_dp = model.engine.params.clone()
  if rank == 0:
      os.environ["WORLD_SIZE"] = "1"                      # (BaseModel reads it: a W > 1 model broadcasts its parameters at construction)
      _, one = build()
      os.environ["WORLD_SIZE"] = str(world)
      assert one.world_size == 1 and one.rank == 0
      eng = one.engine
      ref_losses = []
      for i in range(STEPS):
          b = batch_of(perm[i * B:min((i + 1) * B, n)])
          eng.fwd_bwd(eng.make_plan(b["in_item_id"], b["item_id"], b["seqlen"], neg_item=b["neg_item"].contiguous().view(-1), sample_neg=False))
          one._cl_term(b["in_item_id"], b["seqlen"], views=b["_views"], fold_loss=True)
          ref_losses.append(float(eng.grads[eng.n_params + 1] / eng.grads[eng.n_params]))
          eng.adam_step(one._api_plan())
      torch.cuda.synchronize()
      d = float((p_dp - eng.params).abs().max())
      dl = max(abs(a - b) for a, b in zip(losses, ref_losses))
      print("DP_CL_CHECK world=%d tail=%d max|dp - single| = %.3e (max|param| %.3f), max loss diff %.2e, losses %s" %
            (world, TAIL, d, float(eng.params.abs().max()), dl, [round(x, 5) for x in losses]), flush=True)
      assert d < 2e-4 and dl < 2e-5, (d, dl)
except BaseException as e:
    _report(e)
    raise
chk = torch.tensor([float(p_dp.double().sum())], dtype=torch.float64)
lst = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(lst, chk)
if rank == 0:
    print("DP_CL_CHECK replica checksums equal:", all(float(x) == float(lst[0]) for x in lst), flush=True)
dist.destroy_process_group()
