"""fit() on the Markov-signal synthetic data (data.markov) at a small catalog; prints the test metrics.  MODEL, N_ITEMS, N_ROWS, BATCH, EPOCHS env."""
import os, sys, logging, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
os.makedirs("/tmp/learn_probe", exist_ok=True)
os.chdir("/tmp/learn_probe")
logging.getLogger("CDR").setLevel(logging.WARNING)
from dr4sr_amd import quickstart
from dr4sr_amd.utils import load_config
E = os.environ.get
cfg = load_config({"model": E("MODEL", "MetaModel"), "dataset": "synthetic-toys"})
cfg["data"].update({"n_items": int(E("N_ITEMS", "300")), "n_rows": int(E("N_ROWS", "4000")), "n_eval_rows": 512, "markov": 0.9, "seed": 3})
if E("PREFIX"):
    cfg["data"]["prefix_rows"] = True
cfg["train"].update({"device": "cuda", "epochs": int(E("EPOCHS", "30")), "batch_size": int(E("BATCH", "128"))})
if E("WARM"):
    cfg["train"]["warmup_epoch"] = int(E("WARM"))
if E("INTERVAL"):
    cfg["train"]["interval"] = int(E("INTERVAL"))
if E("SUB"):
    cfg["model"]["sub_model"] = E("SUB")
cfg["eval"]["batch_size"] = 512
if E("OVERRIDES"):                                    # JSON {section: {key: value}}
    import json
    for sec, kv in json.loads(E("OVERRIDES")).items():
        cfg[sec].update(kv)
out = quickstart.run(cfg)
print({k: round(float(v), 4) for k, v in out.items()})
