"""End-to-end data-parallel fit() on ONE GPU: W ranks (gloo) share cuda:0, each trains its slice of every global batch through the
model API (BaseModel._fused_epoch, W > 1 branch: shard bounds, tail batches, all-reduce, bit-identical replicas).
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 tools/dp_fit_check.py"""
import os, sys, logging, faulthandler
faulthandler.dump_traceback_later(90, exit=True)
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
os.makedirs("/tmp/dpfit", exist_ok=True)            # one working directory: rank 0 writes the checkpoint, every rank loads it
os.chdir("/tmp/dpfit")
logging.getLogger("CDR").setLevel(logging.WARNING)
dist.init_process_group("gloo")
from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
MODEL = os.environ.get("MODEL", "SASRec")
cfg = load_config({"model": MODEL, "dataset": "synthetic-toys"})
cfg["data"].update({"n_items": 300, "n_rows": 1000 + 37, "n_eval_rows": 256, "seed": 5})
cfg["model"]["dropout_rate"] = 0.2
cfg["train"].update({"batch_size": 128, "epochs": 3, "device": "cuda:0", "hip_graph": True})
if "interval" in cfg["train"]:
    cfg["train"]["interval"] = 4                          # MetaModel: several outer steps per epoch
cfg["eval"]["batch_size"] = 128
seed_everything(cfg["train"]["seed"])
ds = prepare_datasets(cfg)
model = prepare_model(cfg, ds)
import dr4sr_amd.parallel as par
_orig = par.allreduce_flat
def _host_allreduce(grads, group=None):          # gloo: reduce on the host
    g = grads.cpu()
    dist.all_reduce(g)
    grads.copy_(g)
    return grads
par.allreduce_flat = _host_allreduce
import dr4sr_amd.model.basemodel as bm, dr4sr_amd.model.sasrec as sm
bm.allreduce_flat = _host_allreduce
sm.allreduce_flat = _host_allreduce
_bc = dist.broadcast
def _host_broadcast(t, src=0, **kw):
    if t.is_cuda:
        h = t.cpu(); _bc(h, src=src, **kw); t.copy_(h)
    else:
        _bc(t, src=src, **kw)
dist.broadcast = _host_broadcast
model.fit()
test = model.evaluate()
torch.cuda.synchronize()
p = model.engine.params
chk = torch.tensor([float(p.double().sum()), float(p.double().abs().sum())], dtype=torch.float64)
lst = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(lst, chk)
if rank == 0:
    same = all(bool((x == lst[0]).all()) for x in lst)
    print("DP_FIT model=" + MODEL + " test " + str({k: round(float(v), 4) for k, v in list(test.items())[:2]}) + " world=%d steps=%d replicas identical: %s; finite: %s; train loss %.4f" %
          (world, int(model.engine.state[0]), same, bool(torch.isfinite(p).all()), float(model.logged_metrics.get("train_loss_0", float("nan")))))
dist.destroy_process_group()
