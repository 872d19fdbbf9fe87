"""Run-to-run determinism of whole fit() calls under train.deterministic (GPU tool; tests/test_gpu_deterministic.py calls it).
Usage: python tools/det_fit_check.py [SASRec|CL4SRec|FMLP|GRU4Rec|MetaModel|MetaModel:GRU4Rec|MetaModel:FMLP]  -> prints "identical: True/False" for the
flat parameter buffer (+ the meta module).  MetaModel:<sub>: the DR4SR+ weighting around that sub-model (GRU4Rec / FMLP train through the dense C-ABI composition)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))


def one_fit(name, workdir):
    name, _, meta_sub = name.partition(":")
    from test_gpu_meta import make_config
    from dr4sr_amd.utils import prepare_datasets, prepare_model, seed_everything
    n_items, n_rows, batch = (int(os.environ.get(k, d)) for k, d in (("DET_N_ITEMS", 150), ("DET_ROWS", 600), ("DET_BATCH", 64)))
    sub = (meta_sub or "SASRec") if name in ("SASRec", "MetaModel") else name
    cfg = make_config(n_items, sub=sub, dropout=0.5 if sub != "GRU4Rec" else 0.2, n_rows=n_rows, batch=batch, epochs=3, warmup=0, interval=2)
    if name == "MetaModel" and sub == "FMLP":
        cfg["data"]["prefix_rows"] = True
    if name == "MetaModel" and sub == "GRU4Rec":
        cfg["model"]["sub_overrides"]["model"]["hidden_size"] = 128
    if name != "MetaModel":
        cfg["model"]["model"] = name
        cfg["model"].pop("sub_model"), cfg["model"].pop("sub_overrides")
        if name == "FMLP":
            cfg["data"]["prefix_rows"] = True
        if name == "CL4SRec":
            cfg["model"].update({"augment_type": "item_random", "temperature": 1.0, "cl_weight": 0.1, "tau": 0.2, "gamma": 0.7, "beta": 0.2})
    cfg["train"]["deterministic"] = not os.environ.get("DET_OFF")
    os.makedirs(workdir, exist_ok=True)
    os.chdir(workdir)
    seed_everything(cfg["train"]["seed"])
    ds = prepare_datasets(cfg)
    model = prepare_model(cfg, ds)
    model.fit()
    torch.cuda.synchronize()
    eng = (model.sub_model if name == "MetaModel" else model).engine
    out = [eng.params.detach().clone().cpu()]
    if name == "MetaModel":
        out += [p.detach().clone().cpu() for p in model.meta_module.parameters()]
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "SASRec"
    import tempfile
    a = one_fit(name, tempfile.mkdtemp())
    b = one_fit(name, tempfile.mkdtemp())
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    diff = max(float((x - y).abs().max()) for x, y in zip(a, b))
    print("%s fit x2, train.deterministic %s: identical: %s (max |diff| %.3g, %d tensors, finite %s)"
          % (name, "OFF" if os.environ.get("DET_OFF") else "on", same, diff, len(a), all(bool(torch.isfinite(x).all()) for x in a)))
    sys.exit(0 if same else 1)
