"""wall time per epoch of fit() (training epoch + validation + early-stopping bookkeeping) on the toys-shaped synthetic dataset"""
import os, sys, time, logging, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
os.chdir("/tmp")
logging.getLogger("CDR").setLevel(logging.WARNING)
from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
cfg = load_config({"model": sys.argv[1] if len(sys.argv) > 1 else "SASRec", "dataset": "synthetic-toys"})
cfg["train"]["device"] = "cuda:0"
E = int(os.environ.get("EPOCHS", "30"))
cfg["train"]["epochs"] = E
cfg["train"]["early_stop_patience"] = 1000
seed_everything(cfg["train"]["seed"])
ds = prepare_datasets(cfg)
model = prepare_model(cfg, ds)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.fit()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("fit(): %d epochs in %.3f s = %.2f ms per epoch (train rows %d, val rows %d); metrics %s"
      % (E, dt, dt / E * 1e3, len(ds[0]), len(ds[1]), {k: round(float(v), 4) for k, v in list(model.logged_metrics.items())[:4]}))
