"""Oracle: query x gallery squared-L2 matrix, ranking, Recall@N.  TEST INFRASTRUCTURE ONLY.

Restates ibl/evaluators.py:105-167 on plain arrays (no filename-keyed dict, no process group).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch


def pairwise_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """evaluators.py:122-129: |x_i|^2 + |y_j|^2 - 2 x_i.y_j, float32, no clamp, no sqrt."""
    m, n = x.size(0), y.size(0)
    x = x.view(m, -1)
    y = y.view(n, -1)
    dist_m = torch.pow(x, 2).sum(dim=1, keepdim=True).expand(m, n) + \
        torch.pow(y, 2).sum(dim=1, keepdim=True).expand(n, m).t()
    return torch.addmm(dist_m, x, y.t(), beta=1, alpha=-2)


def pairwise_distance_all(x: torch.Tensor) -> torch.Tensor:
    """evaluators.py:106-114 (query is None and gallery is None): 2|x_i|^2 - 2 x_i.x_j."""
    n = x.size(0)
    x = x.view(n, -1)
    dist_m = torch.pow(x, 2).sum(dim=1, keepdim=True) * 2
    return dist_m.expand(n, n) - 2 * torch.mm(x, x.t())


def ranking(distmat: np.ndarray) -> np.ndarray:
    """Ascending order of every row (evaluators.py:143 uses np.argsort, whose tie order is
    unspecified; the oracle fixes ties as lowest index first = a stable sort)."""
    return np.argsort(distmat, axis=1, kind="stable")


def topk(distmat: np.ndarray, k: int):
    """First k entries of `ranking` with their distances."""
    idx = ranking(distmat)[:, :k]
    return np.take_along_axis(distmat, idx, axis=1), idx


def spatial_nms(pred: Sequence[int], db_ids: Sequence[int], topN: int) -> List[int]:
    """evaluators.py:132-140: among the first topN predictions keep the first of each pid."""
    assert len(pred) == len(db_ids)
    pred_select = list(pred[:topN])
    seen = set()
    keep = []
    for i in pred_select:
        pid = db_ids[i]
        if pid not in seen:
            seen.add(pid)
            keep.append(i)
    return keep


def recalls_from_ranking(sort_idx: np.ndarray, gt: Sequence[Sequence[int]],
                         gallery_pids: Sequence[int] = None,
                         recall_topk: Sequence[int] = (1, 5, 10), nms: bool = False) -> np.ndarray:
    """evaluators.py:149-160: query q counts at n (and every larger n) if pred[:n] hits gt[q].

    NB the reference applies spatial_nms to the FULL argsort row with pid list
    `db_ids = [db[1] for db in gallery]`, where the list index is the gallery position."""
    correct_at_n = np.zeros(len(recall_topk))
    for qIx, pred in enumerate(sort_idx):
        pred = pred.tolist()
        if nms:
            # spatial_nms asserts len(pred) == len(db_ids) on the full row; only the first
            # max(recall_topk)*12 predictions are looked at (evaluators.py:152-153)
            sel = pred[: max(recall_topk) * 12]
            seen, keep = set(), []
            for i in sel:
                pid = gallery_pids[i]
                if pid not in seen:
                    seen.add(pid)
                    keep.append(i)
            pred = keep
        for i, n in enumerate(recall_topk):
            if np.any(np.isin(pred[:n], gt[qIx])):
                correct_at_n[i:] += 1
                break
    return correct_at_n / len(gt)


def evaluate_all(distmat: np.ndarray, gt, gallery_pids=None, recall_topk=(1, 5, 10), nms=False):
    """evaluators.py:142-167 without the prints."""
    return recalls_from_ranking(ranking(np.asarray(distmat)), gt, gallery_pids, recall_topk, nms)
