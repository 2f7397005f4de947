"""NetVLAD centroid initialisation on the GPU (f4; examples/cluster.py:110-115)."""
import numpy as np
import pytest
import torch

from openibl_amd import cluster, ops, synth
from test_host_logic import check_kmeans_against_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,K", [(1000, 64, 7), (5000, 512, 64), (777, 96, 3), (300, 40, 300)])
def test_cluster_means_equals_the_rounded_fp64_mean(dev, n, d, K):
    """oibl_cluster_means: per-cluster means, bit-identical to the fp64 mean rounded to fp32; clusters
    without members keep their centre and report a zero count; ragged d (not a multiple of 256)."""
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    labels = rng.integers(0, K, size=n).astype(np.int32)
    labels[labels == K - 1] = 0                      # the last cluster stays empty
    c0 = rng.standard_normal((K, d)).astype(np.float32)
    c = torch.from_numpy(c0).to(dev)
    counts = ops.cluster_means(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), c)
    want = c0.copy()
    for k in range(K):
        m = labels == k
        if m.any():
            want[k] = x[m].astype(np.float64).mean(0).astype(np.float32)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(labels, minlength=K))
    assert int(counts[K - 1]) == 0
    assert np.array_equal(c.cpu().numpy(), want)


def test_kmeans_centroids_on_the_gpu_against_the_reference_call(dev):
    """The whole initialisation with the HIP assignment (oibl_sqdist_topk, k = 1, exact fp32) and
    update steps against the output of the reference's own KMeans call (tests/golden/kmeans.npz)."""
    check_kmeans_against_golden(lambda x, K, seed: cluster.kmeans_centroids(x, K, 100, seed, device=dev,
                                                                            return_n_iter=True))


def test_assignment_step_is_argmin_of_the_distance(dev):
    x = synth.kmeans_points(4000, 128, 30, seed=9)
    c = x[:50].copy()
    lab = cluster._hip_assign(torch.from_numpy(x).to(dev), c).cpu().numpy()
    d = ((x[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(2)
    got = d[np.arange(len(x)), lab]
    assert np.all(got <= d.min(1) + 1e-6)            # the nearest centre up to fp32 rounding of a tie
    assert np.mean(lab == d.argmin(1)) > 0.999
