"""EarlyStopping = early stop + best-checkpoint keeper, file format of the reference
(utils/callbacks.py:12-139): torch.save({'config','model','epoch','parameters','metric'}) to
saved/<Model>/<dataset>/<log-stem>.ckpt; 'parameters' is the model state_dict (reference key names), so
checkpoints are interchangeable with the reference in both directions."""
from __future__ import annotations

import copy
import logging
import os

import numpy as np
import torch


class EarlyStopping:
    def __init__(self, model, monitor: str, dataset_name: str, save_dir: str = "saved", filename: str = None,
                 patience: int = 10, delta: float = 0.0, mode: str = "max"):
        if mode not in ("min", "max"):
            raise ValueError(f"`mode` can only be `min` or `max`, but `{mode}` is given.")
        self.monitor, self.patience, self.delta, self.mode = monitor, patience, delta, mode
        self.model_name = model.__class__.__name__
        self.save_dir = save_dir
        self.logger = logging.getLogger("CDR")
        self._counter = 0
        self.best_value = np.inf if mode == "min" else -np.inf
        self.best_ckpt = {"config": model.config, "model": self.model_name, "epoch": 0,
                          "parameters": copy.deepcopy(model.state_dict()), "metric": {monitor: self.best_value}}
        if filename is not None:
            self._best_ckpt_path = filename
        else:
            stem = "model"
            for h in self.logger.handlers:
                if type(h) == logging.FileHandler:
                    stem = os.path.basename(h.baseFilename).split(".")[0]
            self._best_ckpt_path = f"{self.model_name}/{dataset_name}/{stem}.ckpt"
        if self.save_dir is not None:
            os.makedirs(os.path.dirname(os.path.join(self.save_dir, self._best_ckpt_path)), exist_ok=True)

    def __call__(self, model, epoch: int, metrics: dict) -> bool:
        if self.monitor not in metrics:
            raise ValueError(f"monitor {self.monitor} not in given `metrics`.")
        v = float(metrics[self.monitor])
        better = v >= self.best_value + self.delta if self.mode == "max" else v <= self.best_value - self.delta
        if better:
            self._counter = 0
            self.best_value = v
            self.best_ckpt["parameters"] = copy.deepcopy(model.state_dict())
            self.best_ckpt["metric"] = metrics
            self.best_ckpt["epoch"] = epoch
            self.logger.info("{} improved. Best value: {:.4f}".format(self.monitor, v))
            self.save_checkpoint(epoch)
        else:
            self._counter += 1
        if self._counter >= self.patience:
            self.logger.info(f"Early stopped. Since the metric {self.monitor} haven't been improved for {self._counter} epochs.")
            self.logger.info(f"The best score of {self.monitor} is {self.best_value:.4f} on epoch {self.best_ckpt['epoch']}")
            return True
        return False

    def save_checkpoint(self, epoch: int) -> None:
        if self.save_dir is None:
            raise ValueError("fail to save the model, self.save_dir can't be None!")
        if int(os.environ.get("RANK", "0")) != 0:          # data parallel: replicas are identical, rank 0 writes
            return
        self.save_path = os.path.join(self.save_dir, self._best_ckpt_path)
        torch.save(self.best_ckpt, self.save_path)
        self.logger.info(f"Best model checkpoint saved in {self.save_path}.")

    def get_checkpoint_path(self) -> str:
        return self._best_ckpt_path
