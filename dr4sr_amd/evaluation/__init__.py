"""Rank metrics used by the shipped configs (eval.val_metrics/test_metrics = [ndcg, recall]), with the
semantics of the reference's evaluation/__init__.py:9-33 (recall), :107-134 (ndcg), :235-299 (registry)."""
from __future__ import annotations

import sys
from typing import List, Union

import torch


def recall(pred: torch.Tensor, target: torch.Tensor, k: int, mean: bool = True):
    """pred [B,K] bool hit matrix in rank order; target [B,n_target] relevance."""
    count = (target > 0).sum(-1)
    out = pred[:, :k].sum(dim=-1).float() / count
    return out.mean() if mean else out


def _dcg(pred: torch.Tensor, k: int):
    k = min(k, pred.size(1))
    denom = torch.log2(torch.arange(k, device=pred.device).type_as(pred) + 2.0).view(1, -1)
    return (pred[:, :k] / denom).sum(dim=-1)


def ndcg(pred: torch.Tensor, target: torch.Tensor, k: int, mean: bool = True):
    pred_dcg = _dcg(pred.float(), k)
    ideal = _dcg(torch.sort((target > 0).float(), descending=True)[0], k)
    irrelevant = torch.all(target <= sys.float_info.epsilon, dim=-1)
    pred_dcg = torch.where(irrelevant, torch.zeros_like(pred_dcg), pred_dcg / torch.where(irrelevant, torch.ones_like(ideal), ideal))
    return pred_dcg.mean() if mean else pred_dcg


metric_dict = {"ndcg": ndcg, "recall": recall}
_RANK = {"ndcg", "precision", "recall", "map", "mrr", "hit", "f1"}


def get_rank_metrics(metric):
    metric = metric if isinstance(metric, list) else [metric]
    return [(m, metric_dict[m]) for m in metric if m in _RANK and m in metric_dict]


def get_eval_metrics(metric_names: Union[List[str], str], cutoffs, validation: bool = False) -> List[str]:
    metric_names = metric_names if isinstance(metric_names, list) else [metric_names]
    rank = {m for m, _ in get_rank_metrics(metric_names)}
    if cutoffs is None:
        return []
    cutoffs = cutoffs if isinstance(cutoffs, list) else [cutoffs]
    if validation:
        cutoffs = cutoffs[:1]
    return [f"{m}@{c}" if m in rank else m for c in cutoffs for m in metric_names]
