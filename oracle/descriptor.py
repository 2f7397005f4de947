"""Oracle: image -> VGG16 conv5_3 -> NetVLAD -> intra-/L2-norm -> PCA -> 4096-d descriptor.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch-CPU restatement of the reference;
`dtype=torch.float64` gives a tighter yardstick than the reference's own fp32.
All functions take the reference's state-dict tensors (keys as in SURVEY.md §8a1).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

# torchvision vgg16.features[:-2] (ibl/models/vgg.py:40-42): conv at these indices, ReLU after
# every conv except the last (28), 2x2/2 max-pool after conv 2, 7, 14, 21.
CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
POOL_AFTER = (2, 7, 14, 21)


def vgg16_conv5(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "base_model.base.",
                upto: Optional[int] = None) -> torch.Tensor:
    """VGG.forward's `self.base(x)` (vgg.py:61-62): [N][3][H][W] -> [N][512][H/16][W/16], conv5_3
    output WITHOUT its ReLU.  `upto` stops after that many convolutions (activation after the
    layer's ReLU / pool), for per-layer checks."""
    dt = x.dtype
    for li, idx in enumerate(CONV_IDX):
        w = sd[f"{prefix}{idx}.weight"].to(dt)
        b = sd[f"{prefix}{idx}.bias"].to(dt)
        x = F.conv2d(x, w, b, stride=1, padding=1)
        if idx != 28:
            x = F.relu(x)
        if idx in POOL_AFTER:
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        if upto is not None and li + 1 == upto:
            break
    return x


def global_max(feat: torch.Tensor) -> torch.Tensor:
    """AdaptiveMaxPool2d(1) + view (vgg.py:67-68): [N][C][h][w] -> [N][C]."""
    return feat.flatten(2).max(dim=2).values


def netvlad(feat: torch.Tensor, conv_weight: torch.Tensor, centroids: torch.Tensor,
            normalize_input: bool = True) -> torch.Tensor:
    """NetVLAD.forward (netvlad.py:44-61): [N][C][h][w] -> [N][K][C], un-normalised.

    The reference expands a [N][K][C][P] residual and sums over P (netvlad.py:56-59); the same sum
    is  sum_p a[k,p] x[c,p]  -  centroids[k,c] sum_p a[k,p]  (evaluated per image here)."""
    N, C = feat.shape[:2]
    dt = feat.dtype
    K = centroids.shape[0]
    x = F.normalize(feat, p=2, dim=1) if normalize_input else feat       # netvlad.py:46-47
    logits = F.conv2d(x, conv_weight.to(dt).view(K, C, 1, 1)).view(N, K, -1)   # :50
    a = F.softmax(logits, dim=1)                                          # :51
    xf = x.reshape(N, C, -1)                                              # :53
    vlad = torch.bmm(a, xf.transpose(1, 2)) - a.sum(dim=2, keepdim=True) * centroids.to(dt)[None]
    return vlad


def netvlad_residual_form(feat, conv_weight, centroids, normalize_input=True):
    """Literal form of netvlad.py:56-59 (materialises the residual); small inputs only.  Used to
    pin `netvlad` above against the reference's own formulation."""
    N, C = feat.shape[:2]
    K = centroids.shape[0]
    x = F.normalize(feat, p=2, dim=1) if normalize_input else feat
    a = F.softmax(F.conv2d(x, conv_weight.view(K, C, 1, 1)).view(N, K, -1), dim=1)
    xf = x.reshape(N, C, -1)
    residual = xf[:, None, :, :] - centroids[None, :, :, None]
    residual = residual * a[:, :, None, :]
    return residual.sum(dim=-1)


def normalize_vlad(vlad: torch.Tensor) -> torch.Tensor:
    """netvlad.py:100-102 (= :78-80, :202-204): intra-norm over C, flatten k-major, L2."""
    v = F.normalize(vlad, p=2, dim=2)
    v = v.reshape(vlad.shape[0], -1)
    return F.normalize(v, p=2, dim=1)


def pca_project(v: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """pca_layer (1x1 conv) + F.normalize (netvlad.py:105-108; pca.py:117-121): [N][D] -> [N][d]."""
    dt = v.dtype
    d = weight.shape[0]
    y = F.linear(v, weight.to(dt).view(d, -1), bias.to(dt))
    return F.normalize(y, p=2, dim=-1)


def embednet(x, sd, dtype=torch.float32):
    """EmbedNet.forward eval path (netvlad.py:73-82): returns (pool_x [N][512], vlad [N][32768])."""
    feat = vgg16_conv5(x.to(dtype), sd)
    vl = netvlad(feat, sd["net_vlad.conv.weight"], sd["net_vlad.centroids"])
    return global_max(feat), normalize_vlad(vl)


def embednetpca(x, sd, dtype=torch.float32, return_intermediates: bool = False):
    """EmbedNetPCA.forward (netvlad.py:95-110): [N][3][H][W] -> [N][4096] unit-norm rows."""
    feat = vgg16_conv5(x.to(dtype), sd)
    vl = netvlad(feat, sd["net_vlad.conv.weight"], sd["net_vlad.centroids"])
    vn = normalize_vlad(vl)
    out = pca_project(vn, sd["pca_layer.weight"], sd["pca_layer.bias"])
    if return_intermediates:
        return {"feat": feat, "vlad_raw": vl, "vlad_norm": vn, "desc": out}
    return out


def extract_cnn_feature(x, sd, vlad=True, dtype=torch.float32, with_pca=True):
    """extract_cnn_feature (evaluators.py:22-34): forward + one more F.normalize(dim=-1)."""
    if with_pca:
        return F.normalize(embednetpca(x, sd, dtype), p=2, dim=-1)
    pool_x, vl = embednet(x, sd, dtype)
    return F.normalize(vl if vlad else pool_x, p=2, dim=-1)


def scaled_size(H: int, W: int, s: float):
    return max(16, int(round(H * s))), max(16, int(round(W * s)))


def multiscale_descriptor(x, sd, scales=(1.0, 2.0 ** -0.5, 0.5), dtype=torch.float32, with_pca=True):
    """Definition of the multi-scale extension (BASELINE.json configs[4]; NOT in the reference, see
    openibl_amd/multiscale.py): per scale F.interpolate(bilinear, align_corners=False) ->
    extract_cnn_feature, then the L2-normalised sum over scales."""
    H, W = int(x.shape[2]), int(x.shape[3])
    acc = None
    for s in scales:
        size = scaled_size(H, W, float(s))
        xs = x if size == (H, W) else F.interpolate(x, size=size, mode="bilinear", align_corners=False)
        d = extract_cnn_feature(xs, sd, dtype=dtype, with_pca=with_pca)
        acc = d if acc is None else acc + d
    return F.normalize(acc, p=2, dim=-1)
