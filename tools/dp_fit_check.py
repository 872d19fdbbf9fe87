"""End-to-end data-parallel quickstart.run() on ONE GPU: W ranks share cuda:0 over the gloo transport of dr4sr_amd/parallel.py
(DR4SR_DP_BACKEND=gloo: RCCL refuses two ranks on one device), each trains its slice of every global batch through the model API
(BaseModel._fused_epoch, W > 1 branch: shard bounds, tail batches, all-reduce, bit-identical replicas) — no monkey-patching: the
product's own call sites run.  The run goes through quickstart.run, i.e. the per-job log / checkpoint stem (rank 0's, broadcast),
rank 0 writing the best checkpoint and EVERY rank loading it in evaluate().
  DR4SR_DP_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 tools/dp_fit_check.py"""
import os, sys, logging, faulthandler
faulthandler.dump_traceback_later(120, exit=True)
os.environ.setdefault("DR4SR_DP_BACKEND", "gloo")
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
work = os.environ.get("DP_FIT_DIR", "/tmp/dpfit")
os.makedirs(work, exist_ok=True)            # one working directory: rank 0 writes the checkpoint, every rank loads it
os.chdir(work)
from dr4sr_amd import quickstart
from dr4sr_amd.parallel import init_distributed
from dr4sr_amd.utils import load_config, seed_everything
MODEL = os.environ.get("MODEL", "SASRec")
cfg = load_config({"model": MODEL, "dataset": "synthetic-toys"})
cfg["data"].update({"n_items": 300, "n_rows": 1000 + 37, "n_eval_rows": 256, "seed": 5})
if MODEL == "MetaModel":
    cfg["model"]["sub_model"] = os.environ.get("SUB_MODEL", "SASRec")      # BASELINE configs[4]; configs/metamodel.yaml's default sub-model is FMLP (prefix rows)
if MODEL == "FMLP":
    cfg["data"]["prefix_rows"] = True                     # one query per row: left-padded prefixes with scalar targets (model/fmlp.py:38)
cfg["model"]["dropout_rate"] = 0.2
if os.environ.get("LOSS_FN"):                             # 'bpr': the model-API loop (BaseModel._api_epoch_dp) instead of the fused step
    cfg["model"]["loss_fn"] = os.environ["LOSS_FN"]
cfg["train"].update({"batch_size": 128, "epochs": 3, "device": "cuda:0", "hip_graph": True, "steps_per_graph": 3})   # 8 full batches: groups of 3, 3, then singles
if "interval" in cfg["train"]:
    cfg["train"]["interval"] = 4                          # MetaModel: several outer steps per epoch
cfg["eval"]["batch_size"] = 128
torch.cuda.set_device(0)
init_distributed("cuda:0")
seed_everything(cfg["train"]["seed"])
holder = {}
_prep = quickstart.run.__globals__["prepare_model"]
def _keep_model(config, ds):                              # only to read the trained replica back for the checksum below
    holder["model"] = _prep(config, ds)
    return holder["model"]
quickstart.run.__globals__["prepare_model"] = _keep_model
try:
    test = quickstart.run(cfg)
except BaseException as e:                                  # the launcher's error page hides the child's traceback: say it on stdout
    import traceback
    print("DP_FIT_ERROR rank %d: %s\n%s" % (rank, repr(e), "".join(traceback.format_exc().splitlines(True)[-12:])), flush=True)
    raise
logging.getLogger("CDR").setLevel(logging.WARNING)
model = holder["model"]
torch.cuda.synchronize()
p = model.engine.params
chk = torch.tensor([float(p.double().sum()), float(p.double().abs().sum())], dtype=torch.float64)
lst = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(lst, chk)
stems = [None] * world
dist.all_gather_object(stems, model.ckpt_path)
if rank == 0:
    same = all(bool((x == lst[0]).all()) for x in lst)
    print("DP_FIT model=" + MODEL + " test " + str({k: round(float(v), 4) for k, v in list(test.items())[:2]}) + " world=%d steps=%d replicas identical: %s; finite: %s; train loss %.4f; one ckpt stem: %s" %
          (world, int(model.engine.state[0]), same, bool(torch.isfinite(p).all()), float(model.logged_metrics.get("train_loss_0", float("nan"))), len(set(stems)) == 1))
dist.destroy_process_group()
