"""NetVLAD centroid initialisation: the k-means of the reference's examples/cluster.py:110-115,

    kmeans = KMeans(n_clusters=args.num_clusters, max_iter=niter, random_state=args.seed).fit(dbFeat[...])
    centroids = kmeans.cluster_centers_

with the Lloyd iterations on the GPU.  What scikit-learn's `KMeans.fit` does for a dense float32
matrix (sklearn/cluster/_kmeans.py: `fit`, `_kmeans_single_lloyd`, `_tolerance`) is restated step by
step:

  * tolerance   tol_abs = mean(var(X, axis=0)) * 1e-4
  * centring    X -= X.mean(axis=0)  (the mean is added back to the centres at the end)
  * seeding     k-means++ with `RandomState(seed)`.  The seeding IS scikit-learn's: its candidate draws
                come out of its own RNG stream and decide everything that follows, so the public
                `sklearn.cluster.kmeans_plusplus` is called on the centred matrix — the same function
                `KMeans.fit` reaches through `_init_centroids` — instead of imitating it.  (The
                reference imports scikit-learn for this step anyway.)
  * one run (n_init = 'auto' = 1 for k-means++) of Lloyd: assign every point to its nearest centre
    (`oibl_sqdist_topk`, k = 1, exact-fp32 mode, ties to the lowest index as `argmin` does), move
    every centre to the mean of its points (`oibl_cluster_means`), relocate empty clusters to the
    points farthest from their centres, stop when the labels repeat or the summed squared centre
    shift is <= tol_abs, at most `max_iter` times.

scikit-learn accumulates the means in float32 in thread-dependent chunks; here they are correctly
rounded, so centres agree to ~1e-6 relative, not bit for bit (tests/golden/kmeans.npz, produced by
the reference's own call).  `assign_fn` / `update_fn` are injection points for the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

__all__ = ["kmeans_centroids"]


def _hip_assign(x_dev, centers: np.ndarray):
    import torch
    from . import ops
    c = torch.from_numpy(centers).to(x_dev.device)
    _, idx = ops.sqdist_topk(x_dev, c, 1, precision="fp32")
    return idx.reshape(-1).contiguous()


def _hip_update(x_dev, labels_dev, centers: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    import torch
    from . import ops
    c = torch.from_numpy(centers).to(x_dev.device).contiguous()
    counts = ops.cluster_means(x_dev, labels_dev, c)
    return c.cpu().numpy(), counts.cpu().numpy()


def _relocate_empty(x: np.ndarray, labels: np.ndarray, centers_old: np.ndarray, centers_new: np.ndarray,
                    counts: np.ndarray) -> None:
    """sklearn/cluster/_k_means_common.pyx `_relocate_empty_clusters_dense`, on the means instead of
    the sums: an empty cluster takes the point farthest from its own centre, that point leaves the
    sum of the cluster it was counted in (its label does not change in this iteration)."""
    empty = np.where(counts == 0)[0]
    if not len(empty):
        return
    dist = ((x - centers_old[labels]) ** 2).sum(axis=1)
    far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]
    sums = centers_new.astype(np.float64) * counts[:, None]
    w = counts.astype(np.float64)
    for new_id, i in zip(empty, far):
        old_id = labels[i]
        sums[old_id] -= x[i]
        sums[new_id] = x[i]
        w[new_id] = 1.0
        w[old_id] -= 1.0
    for c in set(empty.tolist()) | set(labels[far].tolist()):
        if w[c] > 0:
            centers_new[c] = (sums[c] / w[c]).astype(np.float32)
    counts[:] = w.astype(counts.dtype)


def kmeans_centroids(descriptors, num_clusters: int = 64, max_iter: int = 100, seed: int = 43, tol: float = 1e-4,
                     device=None, assign_fn: Optional[Callable] = None, update_fn: Optional[Callable] = None,
                     return_n_iter: bool = False):
    """`KMeans(n_clusters=num_clusters, max_iter=max_iter, random_state=seed).fit(X).cluster_centers_`
    for a dense float32 matrix X [n][d] (numpy or torch).  Returns float32 [num_clusters][d]."""
    from sklearn.cluster import kmeans_plusplus     # the reference's own dependency (cluster.py:13)
    import torch
    x = np.array(descriptors.detach().cpu().numpy() if torch.is_tensor(descriptors) else descriptors,
                 dtype=np.float32, order="C", copy=True)
    if x.ndim != 2 or x.shape[0] < num_clusters:
        raise ValueError(f"n_samples={x.shape[0] if x.ndim == 2 else '?'} should be >= n_clusters={num_clusters}.")
    tol_abs = float(np.mean(np.var(x, axis=0)) * tol)
    x_mean = x.mean(axis=0)
    x -= x_mean
    sq = np.einsum("ij,ij->i", x, x)
    centers, _ = kmeans_plusplus(x, num_clusters, x_squared_norms=sq, random_state=np.random.RandomState(seed))
    centers = np.ascontiguousarray(centers, dtype=np.float32)
    if assign_fn is None or update_fn is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        x_work = torch.from_numpy(x).to(dev)
        assign_fn, update_fn = assign_fn or _hip_assign, update_fn or _hip_update
    else:
        x_work = x
    labels_old = None
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        labels_dev = assign_fn(x_work, centers)
        centers_new, counts = update_fn(x_work, labels_dev, centers)
        labels = labels_dev.cpu().numpy() if torch.is_tensor(labels_dev) else np.asarray(labels_dev)
        labels = labels.astype(np.int64, copy=False)
        if (counts == 0).any():
            _relocate_empty(x, labels, centers, centers_new, counts)
        shift = float(((centers_new.astype(np.float64) - centers.astype(np.float64)) ** 2).sum())
        centers = centers_new
        if labels_old is not None and np.array_equal(labels, labels_old):
            break                                   # strict convergence
        if shift <= tol_abs:
            break
        labels_old = labels
    out = (centers + x_mean).astype(np.float32)
    return (out, n_iter) if return_n_iter else out
