"""wall time of one training epoch vs one validation epoch (basemodel.py:133-151: the reference validates after EVERY epoch) on the
toys-shaped synthetic dataset, through the model API (fit()'s own code paths)"""
import os, sys, time, logging, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
logging.getLogger("CDR").setLevel(logging.WARNING)
from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
cfg = load_config({"model": "SASRec", "dataset": "synthetic-toys"})
cfg["train"]["device"] = "cuda:0"
seed_everything(cfg["train"]["seed"])
ds = prepare_datasets(cfg)
model = prepare_model(cfg, ds)
model._init_model(ds[0])
model.train()
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
tr = timed(lambda: model.training_epoch(0))
model.eval()
val_loader = ds[1].get_loader() if hasattr(ds[1], "get_loader") else None
model.set_eval_domain(cfg["data"]["domain_name_list"][0])
va = timed(lambda: model.validation_epoch(0, ds[1].get_loader(shuffle=False)))
print("train rows %d, val rows %d: training epoch %.2f ms, validation epoch %.2f ms" % (len(ds[0]), len(ds[1]), tr, va))
