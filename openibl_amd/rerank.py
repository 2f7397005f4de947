"""k-reciprocal re-ranking of a distance matrix (f4: a host-side caller of the matching path).

Mirror of `ibl.utils.rerank.re_ranking` (ibl/utils/rerank.py:32-100; Zhong et al., CVPR 2017), the
optional post-processing `Evaluator.evaluate(rerank=True)` applies to the query x gallery matrix
(ibl/evaluators.py:190-200).  The reference runs it in numpy on the host over the dense
(Q+G) x (Q+G) matrix; so does this module — the three distance matrices it consumes come from the
HIP distance kernel.  Same parameters, same return value ([Q][G] float32).

The algorithm, in the order the reference applies it:
  1. D = [[qq, qg], [qgT, gg]] ** 2, every column divided by its maximum, then transposed.
  2. R(i) = the k1+1 nearest items of i;  i and j are k-reciprocal iff j in R(i) and i in R(j).
     The reciprocal set of i is expanded by the (k1/2-sized) reciprocal set of each member whose
     overlap with it exceeds 2/3 of its own size.
  3. V[i, j] = exp(-D[i, j]) over that set, normalised to sum 1 (a sparse soft encoding of i).
  4. k2 > 1: every V[i] is replaced by the mean encoding of i's k2 nearest items.
  5. Jaccard distance of the encodings, J[i, j] = 1 - s / (2 - s), s = sum_c min(V[i, c], V[j, c]);
     result = (1 - lambda) * J + lambda * D for query rows / gallery columns.
"""
from __future__ import annotations

import numpy as np

__all__ = ["re_ranking"]


def _nearest(D: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k smallest entries of every row in ascending order (ties: lowest index)."""
    n = D.shape[1]
    k = min(k, n)
    part = np.argpartition(D, k - 1, axis=1)[:, :k] if k < n else np.tile(np.arange(n), (D.shape[0], 1))
    vals = np.take_along_axis(D, part, axis=1)
    order = np.lexsort((part, vals), axis=1)
    return np.take_along_axis(part, order, axis=1).astype(np.int32)


def _reciprocal(rank: np.ndarray, i: int, k: int) -> np.ndarray:
    """Members j of the k+1 nearest of i that have i among their own k+1 nearest."""
    fwd = rank[i, : k + 1]
    back = rank[fwd, : k + 1]
    return fwd[(back == i).any(axis=1)]


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    q_g = np.asarray(q_g_dist, dtype=np.float32)
    q_q = np.asarray(q_q_dist, dtype=np.float32)
    g_g = np.asarray(g_g_dist, dtype=np.float32)
    nq, ng = q_g.shape
    n = nq + ng
    D = np.empty((n, n), dtype=np.float32)
    D[:nq, :nq], D[:nq, nq:], D[nq:, :nq], D[nq:, nq:] = q_q, q_g, q_g.T, g_g
    D = np.power(D, 2).astype(np.float32)
    D = np.ascontiguousarray((D / D.max(axis=0)).T)

    half = int(np.around(k1 / 2.0))
    rank = _nearest(D, max(k1 + 1, half + 1, k2))
    V = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        base = _reciprocal(rank, i, k1)
        members = [base]
        for cand in base:
            sub = _reciprocal(rank, int(cand), half)
            if len(np.intersect1d(sub, base)) > 2.0 / 3.0 * len(sub):
                members.append(sub)
        idx = np.unique(np.concatenate(members))
        w = np.exp(-D[i, idx])
        V[i, idx] = w / w.sum()
    if k2 != 1:
        V = np.stack([V[rank[i, :k2]].mean(axis=0) for i in range(n)]).astype(np.float32)

    # s[i, j] = sum_c min(V[i, c], V[j, c]) for the query rows, walking only the non-zero columns
    cols = [np.nonzero(V[:, c])[0] for c in range(n)]
    jac = np.zeros((nq, n), dtype=np.float32)
    for i in range(nq):
        s = np.zeros(n, dtype=np.float32)
        for c in np.nonzero(V[i])[0]:
            rows = cols[c]
            s[rows] += np.minimum(V[i, c], V[rows, c])
        jac[i] = 1.0 - s / (2.0 - s)
    final = jac * (1.0 - lambda_value) + D[:nq] * lambda_value
    return final[:, nq:]
