"""Config / environment helpers with the semantics of the reference's utils/utils.py:13-55,:90-109.

load_config merges three YAML layers exactly like the reference: configs/<dataset>.yaml -> config['data'],
configs/basemodel.yaml -> train/model/eval sections, configs/<model>.yaml overlaid section by section.
The config directory is `configs/` relative to the cwd (as in the reference) or $DR4SR_CONFIG_DIR.
Extensions (absent in the reference): `DR4SR_EMBED_DIM` env override for model.embed_dim (the reference CLI
cannot express BASELINE config 4's d=128), train.world_size / train.hip_graph keys with defaults.
"""
from __future__ import annotations

import copy
import importlib
import os
import random

import numpy as np
import torch
import yaml


def _config_dir() -> str:
    return os.environ.get("DR4SR_CONFIG_DIR", "configs")


def seed_everything(seed: int = 1111) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def set_device(train_cfg: dict) -> None:
    """int device id -> that GPU becomes the only visible one and device='cuda' (utils/utils.py:22-26).
    Under torch.distributed.run (LOCAL_RANK set) the local rank selects the GPU instead."""
    if "LOCAL_RANK" in os.environ:
        train_cfg["device"] = f"cuda:{int(os.environ['LOCAL_RANK'])}"
    elif isinstance(train_cfg.get("device"), int):
        if not torch.cuda.is_initialized():
            os.environ["CUDA_VISIBLE_DEVICES"] = str(train_cfg["device"])
            os.environ["HIP_VISIBLE_DEVICES"] = str(train_cfg["device"])
        train_cfg["device"] = "cuda"


def setup_environment(train_cfg: dict) -> None:
    seed_everything(train_cfg["seed"])
    set_device(train_cfg)


def load_config(config: dict) -> dict:
    cdir = _config_dir()
    dataset = config.pop("dataset")
    model_name = config["model"]
    with open(os.path.join(cdir, dataset.lower() + ".yaml")) as f:
        config["data"] = yaml.safe_load(f)
    config["data"]["dataset"] = dataset
    with open(os.path.join(cdir, "basemodel.yaml")) as f:
        config.update(yaml.safe_load(f))
    with open(os.path.join(cdir, model_name.lower() + ".yaml")) as f:
        for section, values in (yaml.safe_load(f) or {}).items():
            config.setdefault(section, {}).update(values)
    config["model"]["model"] = model_name
    if "DR4SR_EMBED_DIM" in os.environ:
        config["model"]["embed_dim"] = int(os.environ["DR4SR_EMBED_DIM"])
    config["train"].setdefault("hip_graph", True)
    return config


def get_model_class(name: str):
    """model.<name.lower()>.<Name> resolution of the reference (utils/utils.py:32-36), inside dr4sr_amd."""
    module = importlib.import_module("dr4sr_amd.model." + name.lower())
    return getattr(module, name)


def prepare_datasets(config: dict):
    model_class = get_model_class(config["model"]["model"])
    dataset_class = model_class._get_dataset_class(config)
    out = []
    for phase in ("train", "val", "test"):
        ds = dataset_class(config, phase=phase)
        ds.build()
        out.append(ds)
    return tuple(out)


def prepare_model(config: dict, dataset_list):
    return get_model_class(config["model"]["model"])(config, dataset_list)


def clone_config(config: dict) -> dict:
    return copy.deepcopy(config)
