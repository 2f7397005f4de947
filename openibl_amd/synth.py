"""Deterministic synthetic weights and inputs for the descriptor + matching path.

Neither the released checkpoint (hubconf.py:10, a GitHub release URL) nor the Pittsburgh / Tokyo
images are reachable here, so tests, goldens and the benchmark all run on these seeded stand-ins.
Everything is drawn from numpy's PCG64 stream (stable across numpy versions and machines), never
from torch's RNG, so that the build container and the GPU box generate bit-identical tensors.

State-dict keys and shapes are exactly those of the reference's
`vgg16_netvlad()` / `EmbedNetPCA`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

# torchvision vgg16 `features` indices of the 13 convolutions kept by ibl/models/vgg.py:40-42
CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
CONV_CH = ((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
           (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512))
NUM_CLUSTERS = 64
DIM = 512
PCA_DIM = 4096

# input normalisation of the reference test transform (ibl/utils/data/__init__.py:40-41):
# Normalize(mean, std = 1/255) on ToTensor output  ==  pixel_0..255 - 255 * mean
MEAN = (0.48501960784313836, 0.4579568627450961, 0.4076039215686255)
STD = 0.00392156862745098


def _normal(rng: np.random.Generator, shape, std: float) -> torch.Tensor:
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(std))


def backbone_state(seed: int = 0, prefix: str = "base_model.base.", trained_like: bool = False,
                   gain: float = 30.0) -> "OrderedDict[str, torch.Tensor]":
    """conv weights ~ N(0, sqrt(2 / fan_out)) like VGG.reset_params (vgg.py:72-77); biases are
    small non-zero values (the reference zero-fills them, which would leave the bias path of the
    kernels untested).

    trained_like=True: the same draw reshaped towards what a trained MatConvNet VGG16 looks like to the
    arithmetic on 0-255-scale pixels (ibl/utils/data/__init__.py:40-41) — the released checkpoint cannot be
    fetched here.  Per OUTPUT channel a log-normal gain (sigma 0.5: the channel norms of a trained layer
    spread over an order of magnitude), 3 % dead channels (zero filter, negative bias: always off behind the
    ReLU), per INPUT channel a milder log-normal gain (sigma 0.25), every layer re-normalised to its original
    Frobenius norm so that the signal neither dies nor explodes with depth, and conv1_1 scaled by `gain`
    (biases of all layers with it: the network is positively homogeneous) so that the activations behind
    conv1_2 peak in the thousands instead of at ~100.  A separate random stream: the default draw (the
    goldens' weights) is unchanged."""
    rng = np.random.default_rng([seed, 1])
    sd = OrderedDict()
    for idx, (cin, cout) in zip(CONV_IDX, CONV_CH):
        std = math.sqrt(2.0 / (cout * 9))
        sd[f"{prefix}{idx}.weight"] = _normal(rng, (cout, cin, 3, 3), std)
        sd[f"{prefix}{idx}.bias"] = _normal(rng, (cout,), 0.05)
    if not trained_like:
        return sd
    rng2 = np.random.default_rng([seed, 11])
    for li, (idx, (cin, cout)) in enumerate(zip(CONV_IDX, CONV_CH)):
        w, b = sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"]
        frob = float(w.norm())
        go = torch.from_numpy(np.exp(rng2.normal(0.0, 0.5, size=cout)).astype(np.float32))
        gi = torch.from_numpy(np.exp(rng2.normal(0.0, 0.25, size=cin)).astype(np.float32))
        dead = torch.from_numpy(rng2.uniform(size=cout) < (0.03 if li < 12 else 0.0))
        w = w * go[:, None, None, None] * gi[None, :, None, None]
        w[dead] = 0.0
        w = w * (frob / float(w.norm()))
        b = b * go
        b[dead] = -0.1
        if li == 0:
            w = w * gain
        sd[f"{prefix}{idx}.weight"] = w.contiguous()
        sd[f"{prefix}{idx}.bias"] = (b * gain).contiguous()
    return sd


# Per-layer activation maxima (behind ReLU; conv5_3: |pre-ReLU|) that fixed-point studies of the Caffe / MatConvNet
# VGG16 report on mean-subtracted 0-255 pixels: hundreds behind conv1_1, rising to a few 1e4 in conv3 / conv4_1, falling
# back to hundreds at conv5_3 — inside fp16 (65504) with a factor of two to spare at the peak, on a calibration set.
CALIBRATED_PEAKS = (9.0e2, 2.5e3, 5.0e3, 9.0e3, 1.4e4, 2.0e4, 2.8e4, 1.8e4, 9.0e3, 4.0e3, 2.0e3, 8.0e2, 3.0e2)


def backbone_state_calibrated(seed: int = 0, prefix: str = "base_model.base.", peaks=CALIBRATED_PEAKS,
                              calib_images: int = 2, calib_hw=(128, 160)) -> "OrderedDict[str, torch.Tensor]":
    """backbone_state(trained_like=True) with HEAVIER tails (log-normal output gains of sigma 0.7) whose layers are
    rescaled one after the other so that the largest activation of a small calibration batch behind layer l is
    peaks[l] (the f16mx range guard's fallback RATE depends on activations the synthetic
    weights never produced).  A 480x640 test batch has ~20x the pixels of the calibration batch: its maxima lie
    further out in the tails — which is the point.  Deterministic (CPU convolutions, fp32)."""
    import torch.nn.functional as F
    sd = backbone_state(seed, prefix, trained_like=True, gain=1.0)
    rng = np.random.default_rng([seed, 12])
    x = images(calib_images, calib_hw[0], calib_hw[1], seed=4242 + seed)
    for li, (idx, (cin, cout)) in enumerate(zip(CONV_IDX, CONV_CH)):
        w, b = sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"]
        extra = torch.from_numpy(np.exp(rng.normal(0.0, 0.5, size=cout)).astype(np.float32))   # 0.5 (+) 0.5 -> 0.7
        w, b = w * extra[:, None, None, None], b * extra
        y = F.conv2d(x, w, b, padding=1)
        peak = float(y.abs().max() if li == 12 else y.clamp_min(0).max())
        sc = float(peaks[li]) / max(peak, 1e-20)
        w, b = (w * sc).contiguous(), (b * sc).contiguous()
        sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"] = w, b
        x = F.conv2d(x, w, b, padding=1)
        if li != 12:
            x = F.relu(x)
        if li in (1, 3, 6, 9):
            x = F.max_pool2d(x, 2, 2)
    return sd


def netvlad_state(seed: int = 0, prefix: str = "net_vlad.", alpha: float = 30.0
                  ) -> "OrderedDict[str, torch.Tensor]":
    """A trained-looking NetVLAD layer: centroids of norm ~0.08, conv.weight = alpha * unit
    centroid (+ a perturbation, since the two are independent parameters after training) so that
    the soft-assignment is neither uniform nor one-hot."""
    rng = np.random.default_rng([seed, 2])
    cent = _normal(rng, (NUM_CLUSTERS, DIM), 0.08 / math.sqrt(DIM))
    unit = cent / cent.norm(dim=1, keepdim=True)
    conv = alpha * unit + _normal(rng, (NUM_CLUSTERS, DIM), 0.02 * alpha / math.sqrt(DIM))
    sd = OrderedDict()
    sd[f"{prefix}centroids"] = cent.contiguous()
    sd[f"{prefix}conv.weight"] = conv.reshape(NUM_CLUSTERS, DIM, 1, 1).contiguous()
    return sd


def pca_state(seed: int = 0, prefix: str = "pca_layer.", dim: int = PCA_DIM
              ) -> "OrderedDict[str, torch.Tensor]":
    rng = np.random.default_rng([seed, 3])
    sd = OrderedDict()
    sd[f"{prefix}weight"] = _normal(rng, (dim, NUM_CLUSTERS * DIM, 1, 1), 0.01)
    sd[f"{prefix}bias"] = _normal(rng, (dim,), 0.01)
    return sd


def embednetpca_state(seed: int = 0, trained_like: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Full state dict of the reference's EmbedNetPCA (30 tensors, 149 002 048 parameters)."""
    sd = backbone_state(seed, trained_like=trained_like)
    sd.update(netvlad_state(seed))
    sd.update(pca_state(seed))
    return sd


def embednet_state(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    sd = backbone_state(seed)
    sd.update(netvlad_state(seed))
    return sd


def images(n: int, height: int = 480, width: int = 640, seed: int = 1) -> torch.Tensor:
    """[n][3][height][width] float32, normalised exactly like the reference's test transform.

    uint8 pixels = a per-image mixture of low-frequency sinusoids, soft blobs and noise (white
    noise alone makes every image collapse to nearly the same descriptor), then
    (pixel / 255 - mean) / std  ->  values in about [-124, 151]."""
    rng = np.random.default_rng([seed, 4])
    yy = np.linspace(0.0, 1.0, height, dtype=np.float32)[:, None]
    xx = np.linspace(0.0, 1.0, width, dtype=np.float32)[None, :]
    out = np.empty((n, 3, height, width), dtype=np.float32)
    for i in range(n):
        img = np.zeros((3, height, width), dtype=np.float32)
        for c in range(3):
            acc = np.zeros((height, width), dtype=np.float32)
            for _ in range(6):
                fy, fx = rng.uniform(0.5, 14.0, size=2).astype(np.float32)
                ph = np.float32(rng.uniform(0, 2 * math.pi))
                amp = np.float32(rng.uniform(0.3, 1.0))
                acc += amp * np.sin(2 * math.pi * (fy * yy + fx * xx) + ph)
            for _ in range(4):
                cy, cx = rng.uniform(0, 1, size=2).astype(np.float32)
                s = np.float32(rng.uniform(0.03, 0.2))
                amp = np.float32(rng.uniform(-2.0, 2.0))
                acc += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
            img[c] = acc
        img = (img - img.min()) / max(float(img.max() - img.min()), 1e-6) * 255.0
        img += rng.normal(0.0, 6.0, size=img.shape).astype(np.float32)
        u8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        x = u8.astype(np.float32) / np.float32(255.0)
        for c in range(3):
            out[i, c] = (x[c] - np.float32(MEAN[c])) / np.float32(STD)
    return torch.from_numpy(out)


def descriptors(n: int, dim: int = PCA_DIM, seed: int = 2) -> torch.Tensor:
    """n unit-norm float32 descriptors."""
    rng = np.random.default_rng([seed, 5])
    x = rng.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return torch.from_numpy(x)


def retrieval_problem(num_query: int, num_gallery: int, dim: int = PCA_DIM, seed: int = 3,
                      positives_per_query: int = 2, noise: float = 0.6,
                      hard_fraction: float = 0.3, views_per_place: int = 1,
                      hard_noise_mult: float = 25.0):
    """Synthetic matching set with planted positives.

    Returns (q [Q][dim], g [G][dim], gt: list[list[int]], gallery_pids: list[int]).
    For each query a few gallery rows are replaced by normalize(query + sigma * noise); a fraction
    of the queries get a much larger sigma (cosine to the query ~4 sigma of the random-pair
    distribution) so that Recall@1 < Recall@5 < Recall@10 < 1 and the numbers
    are not trivially 0 or 1.  `views_per_place` > 1 gives consecutive gallery rows the same pid
    (Tokyo 24/7 style, exercises spatial_nms)."""
    rng = np.random.default_rng([seed, 6])
    q = rng.standard_normal((num_query, dim), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g = rng.standard_normal((num_gallery, dim), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    gt = []
    slots = rng.permutation(num_gallery)[: num_query * positives_per_query]
    slots = slots.reshape(num_query, positives_per_query)
    hard = rng.uniform(size=num_query) < hard_fraction
    for i in range(num_query):
        amp = noise * (hard_noise_mult if hard[i] else 1.0) / math.sqrt(dim)
        for j in slots[i]:
            v = q[i] + np.float32(amp) * rng.standard_normal(dim, dtype=np.float32)
            g[j] = v / np.linalg.norm(v)
        gt.append(sorted(int(j) for j in slots[i]))
    pids = [int(j // views_per_place) for j in range(num_gallery)]
    return torch.from_numpy(q), torch.from_numpy(g), gt, pids


def tokyo_problem(num_query: int = 315, num_gallery: int = 75984, dim: int = PCA_DIM, seed: int = 5,
                  views: int = 12, distractors: int = 12, place_noise: float = 0.6, noise: float = 0.6,
                  hard_fraction: float = 0.5, far=(11.0, 20.0)):
    """A matching set shaped like Tokyo 24/7 (315 queries x 75 984 gallery images = 12
    views of each place; the reference evaluates it with spatial NMS — examples/test.py:130 — i.e. de-duplicates
    the first 120 ranks by place id, ibl/evaluators.py:132-140, 152-153).

    Returns (q [Q][dim], g [G][dim], gt, gallery_pids) like retrieval_problem.  The views of a place are
    near-duplicates (normalize(centre + place_noise * noise / sqrt(dim))).  Every query has ONE true place — all of
    its views are ground truth, each its own normalize(query + sigma * noise): near (sigma = `noise`) for the easy
    queries, sigma = noise * U(far) for the hard ones — and `distractors` other places whose views sit at
    noise * U(far) from it.  A hard query's first ranks are therefore runs of ~12 near-tied views of one place after
    the other: without NMS the top-10 is one place, with NMS it is ten — Recall@N with and without NMS differ, and
    Recall@1 < Recall@5 < Recall@10 < 1."""
    rng = np.random.default_rng([seed, 24, 7])
    V = int(views)
    P = -(-num_gallery // V)
    if num_query * (1 + distractors) > P:
        raise ValueError("tokyo_problem: not enough places for one true place + distractors per query")
    q = rng.standard_normal((num_query, dim), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    # the background gallery in 16 blocks of places, each from its own child stream ([seed, 24, 7, block]) on a
    # thread of its own: the result does not depend on how many threads ran (311M normal draws are 20 s on one core)
    g = np.empty((num_gallery, dim), dtype=np.float32)
    amp_v = np.float32(place_noise / math.sqrt(dim))
    blocks = 16

    def fill(b):
        p0, p1 = b * P // blocks, (b + 1) * P // blocks
        r0, r1 = p0 * V, min(p1 * V, num_gallery)
        if r1 <= r0:
            return
        brng = np.random.default_rng([seed, 24, 7, b])
        c = brng.standard_normal((p1 - p0, dim), dtype=np.float32)
        c /= np.linalg.norm(c, axis=1, keepdims=True)
        v = np.repeat(c, V, axis=0)[: r1 - r0]
        v += amp_v * brng.standard_normal((r1 - r0, dim), dtype=np.float32)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        g[r0:r1] = v

    from concurrent.futures import ThreadPoolExecutor
    import os
    with ThreadPoolExecutor(max_workers=min(blocks, os.cpu_count() or 4)) as ex:
        list(ex.map(fill, range(blocks)))
    places = rng.permutation(P)[: num_query * (1 + distractors)].reshape(num_query, 1 + distractors)
    hard = rng.uniform(size=num_query) < hard_fraction
    sig = noise * rng.uniform(far[0], far[1], size=(num_query, 1 + distractors))
    sig[~hard, 0] = noise
    gt = []
    for i in range(num_query):
        for c in range(1 + distractors):
            r0 = int(places[i, c]) * V
            r1 = min(r0 + V, num_gallery)
            v = q[i][None, :] + np.float32(sig[i, c] / math.sqrt(dim)) * rng.standard_normal((r1 - r0, dim),
                                                                                             dtype=np.float32)
            g[r0:r1] = v / np.linalg.norm(v, axis=1, keepdims=True)
            if c == 0:
                gt.append(list(range(r0, r1)))
    pids = [int(j // V) for j in range(num_gallery)]
    return torch.from_numpy(q), torch.from_numpy(g), gt, pids


def tie_free_matrix(rows: int, cols: int, seed: int = 41, scale: float = 4.0) -> torch.Tensor:
    """float32 [rows][cols] "distance" matrix whose rows are jittered permutations: every row holds
    distinct values in [0, scale), so its argsort is unique (the mining samplers' torch.argsort and
    np.argsort leave the order of ties unspecified)."""
    rng = np.random.default_rng([seed, 9])
    m = np.stack([(rng.permutation(cols) + rng.uniform(0.0, 0.4, size=cols)) * (scale / cols)
                  for _ in range(rows)]).astype(np.float32)
    return torch.from_numpy(m)


def tuple_lists(num_query: int, num_gallery: int, seed: int = 41, positives: int = 3):
    """Positive / exclusion lists for the mining-sampler tests: per query a few "positives" and a
    larger "non-negative" zone around them (gallery positions), seeded.  `positives` > 3 draws them
    from a wider window (the SFRS sampler ranks a pool of positives)."""
    rng = np.random.default_rng([seed, 10] if positives == 3 else [seed, 10, positives])
    half = 3 if positives == 3 else positives
    pos, neg = [], []
    for _ in range(num_query):
        c = int(rng.integers(30, num_gallery - 30))
        pos.append(sorted(int(v) for v in rng.choice(np.arange(c - half, c + half + 1), size=positives, replace=False)))
        neg.append(list(range(c - 25, c + 26)))
    return pos, neg


def kmeans_points(n: int, d: int, blobs: int, seed: int = 43, spread: float = 0.35) -> np.ndarray:
    """float32 [n][d] unit-norm points around `blobs` random directions (the shape of the data
    examples/cluster.py clusters: L2-normalised local descriptors), seeded."""
    rng = np.random.default_rng([seed, 12])
    dirs = rng.standard_normal((blobs, d))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    which = rng.integers(0, blobs, size=n)
    x = dirs[which] + spread * rng.standard_normal((n, d)) / np.sqrt(d) * np.sqrt(d) * 0.1
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x.astype(np.float32))
