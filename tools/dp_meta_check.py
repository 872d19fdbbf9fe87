"""MetaModel outer step under data parallelism, on ONE GPU: W ranks (gloo transport of dr4sr_amd/parallel.py, all on cuda:0) each take
their contiguous slice of a global meta batch and a global train batch, run MetaModel.hypergrad_step (every gradient evaluation of
the hyper-gradient — d L_val / dW, the six Hessian-vector probes, the two mixed-derivative probes — is all-reduced: the flat
sub-model gradient buffer and the meta module's), and must end with the hyper-gradient and the meta-module parameters of a SINGLE
rank run on the full batches (rank 0 re-runs that in the same process with world_size forced to 1).  Explicit Gumbel noise (keyed
by global position) and dropout 0, so that sharding cannot change what is drawn.

Reference: /root/reference model/metamodel.py:123-166, utils/utils.py:145-252 (no distributed path upstream: SURVEY.md §8e).
  DR4SR_DP_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 tools/dp_meta_check.py"""
import os, sys, logging, faulthandler
faulthandler.dump_traceback_later(300, exit=True)
os.environ.setdefault("DR4SR_DP_BACKEND", "gloo")
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
from dr4sr_amd.parallel import init_distributed, shard_bounds
from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
logging.getLogger("CDR").setLevel(logging.WARNING)
cfg = load_config({"model": "MetaModel", "dataset": "synthetic-toys"})
cfg["data"].update({"n_items": 500, "n_rows": 600, "n_eval_rows": 64, "seed": 5})
cfg["model"]["sub_model"] = "SASRec"
cfg["model"]["sub_overrides"] = {"model": {"dropout_rate": 0.0}}
B = int(os.environ.get("DP_META_B", "96"))                 # 8 ranks: 12 rows per rank; 90 -> 12 x 7 + 6 (uneven slices)
cfg["train"].update({"batch_size": B, "device": "cuda:0", "interval": 4, "warmup_epoch": -1})
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
init_distributed("cuda:0")
seed_everything(cfg["train"]["seed"])
ds = prepare_datasets(cfg)
model = prepare_model(cfg, ds)
model._init_model(ds[0])
model.train()
eng, L = model.engine, model.max_seq_len
# a few plain Adam steps away from initialisation would need DP too; instead perturb the parameters identically on every rank
g = torch.Generator().manual_seed(77)
eng.params.add_(0.03 * torch.randn(eng.params.shape, generator=g).to(dev))
eng.views["item_embedding.weight"][0] = 0
model._phi.params.add_(0.05 * torch.randn(model._phi.params.shape, generator=g).to(dev))
fields = ds[0].get_loader().fields
negs = torch.randint(1, 500, (2 * B, L), generator=g).to(dev)
u = torch.rand(2 * B * L, 2, generator=g).clamp_(1e-6, 1 - 1e-6)
gumbel = (-torch.log(-torch.log(u))).to(dev)                                      # rows B..2B-1 (the train batch) are used


def batch(lo, hi):
    rows = torch.arange(lo, hi, device=dev)
    b = {k: v.index_select(0, rows) for k, v in fields.items()}
    b["index"] = rows
    b["neg_item"] = negs[lo:hi].unsqueeze(-1).contiguous()
    return b


def outer(w, r):
    """one hyper-gradient step on slice r of w of the global batches; returns (hyper-gradient, phi after the meta SGD step)"""
    lo, hi = shard_bounds(0, B, B, w, r)
    bv, bt = batch(lo, hi), batch(B + lo, B + hi)
    model._gumbel = gumbel[(B + lo) * L:(B + hi) * L].contiguous()
    hyper = model.hypergrad_step(bv, bt).clone()
    torch.cuda.synchronize()
    return hyper, model._phi.params.clone()


snap = [t.clone() for t in (eng.params, eng.state, model._phi.params, model.meta_optimizer.momentum_buf, model.meta_optimizer.step_count)]
h_dp, phi_dp = outer(world, rank)
ok_single = True
if rank == 0:
    for dst, src in zip((eng.params, eng.state, model._phi.params, model.meta_optimizer.momentum_buf, model.meta_optimizer.step_count), snap):
        dst.copy_(src)
    model.world_size = model.sub_model.world_size = 1                              # the same process as a single rank on the full batches
    h_one, phi_one = outer(1, 0)
    model.world_size = model.sub_model.world_size = world
    eh = float((h_dp - h_one).abs().max() / h_one.abs().max())
    ep = float((phi_dp - phi_one).abs().max())
    print("DP_META world=%d hyper-gradient rel err vs single rank %.3e (|h| max %.3e), phi max abs diff %.3e" % (world, eh, float(h_one.abs().max()), ep), end="; ")
    ok_single = eh < 2e-3 and ep < 1e-6 and float(h_one.abs().max()) > 0
chk = torch.tensor([float(phi_dp.double().sum()), float(h_dp.double().sum())], dtype=torch.float64)
lst = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(lst, chk)
if rank == 0:
    same = all(bool((x == lst[0]).all()) for x in lst)
    print("replicas identical: %s; single-rank equality: %s" % (same, ok_single))
    assert same and ok_single
dist.destroy_process_group()
