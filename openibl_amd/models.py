"""nn.Module mirror of the reference's model zoo (ibl/models/{vgg,netvlad,__init__}.py).

Same class names, constructor arguments, return conventions and — because `base` is a real
nn.Sequential of the same 29 modules and NetVLAD / pca_layer hold the same parameters —
bit-for-bit the same `state_dict()` keys and shapes, so the released `vgg16_netvlad.pth` and
DDP-saved (`module.`-prefixed) checkpoints load unchanged.  The modules only *own* the parameters;
`forward` never calls them.  It hands raw device pointers to the HIP kernels via openibl_amd.ops:

    VGG.forward          -> oibl_vgg16_conv5_forward      (vgg.py:61-70)
    NetVLAD.forward      -> oibl_netvlad_forward           (netvlad.py:44-61)
    EmbedNet.forward     -> backbone + NetVLAD + norms      (netvlad.py:73-82)
    EmbedNetPCA.forward  -> ... + oibl_pca_forward           (netvlad.py:95-110)

Inference only (the reference's training paths are out of scope).  Inputs must be
CUDA(HIP) tensors; there is no CPU path.
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops

__all__ = ["VGG", "vgg16", "NetVLAD", "EmbedNet", "EmbedNetPCA", "EmbedRegionNet", "names",
           "create", "default_precision"]

_CFG_D = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


def default_precision() -> str:
    """The arithmetic a model runs in when nobody chose one: 'f16mx' (fp16 main term + MX-fp6 cross terms on the
    gfx950 matrix cores: descriptors within north_star's 1e-4 of the reference's fp32 result, range-guarded, 4x the
    rate of exact fp32) — what `torch.hub.load(..., 'vgg16_netvlad')(x)` and an unmodified examples/test.py get
    since round 6.  `OPENIBL_AMD_PRECISION` or `model.set_precision(...)` select 'fp32' (exact fp32 MFMA:
    reference-grade numerics, 8e-7), 'bf16x3' (split bf16: 4e-6 at 3x the bf16 matrix work) or 'bf16' (3e-3: not a
    parity mode)."""
    return os.environ.get("OPENIBL_AMD_PRECISION", "f16mx").lower()


class _PrecisionMixin:
    """`module.set_precision('bf16' | 'f16mx' | 'bf16x3' | 'fp32')` switches the arithmetic of the contractions."""

    def set_precision(self, precision: str):
        ops.precision_code(precision)
        for m in self.modules():
            if isinstance(m, _PrecisionMixin):
                m._precision = precision.lower()
                m._drop_cache()
        return self

    def _drop_cache(self):
        """Forget the packed / cast parameter copies and advance the cache generation: captured forwards
        kept for the model (extract._graph_store) point into those copies and are keyed on the generation."""
        self._cache = {}
        self._cache_gen = getattr(self, "_cache_gen", 0) + 1

    @property
    def precision(self) -> str:
        return getattr(self, "_precision", None) or default_precision()

    def invalidate(self):
        """Drop every packed / cast copy of the parameters (they are rebuilt by the next forward).
        The caches are keyed on (data_ptr, _version, device) of the parameters, which catches
        `load_state_dict`, `.to()` and in-place autograd-visible updates, but NOT writes through
        `param.data` (`p.data.copy_(...)`, the idiom of the reference's NetVLAD._init_params,
        ibl/models/netvlad.py:34-42): call this after such an update.  A GraphedDescriptor froze
        its copies at capture time and has to be re-created."""
        for m in self.modules():
            if isinstance(m, _PrecisionMixin):
                m._drop_cache()
        return self

    def _hook_state_dict_loads(self):
        # load_state_dict copies through param.data as well on some paths: always start clean
        def _drop(module, incompatible_keys):   # (a post hook must return None)
            module.invalidate()
        self.register_load_state_dict_post_hook(_drop)


def _fingerprint(params: List[torch.Tensor]) -> Tuple:
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in params)


class VGG(_PrecisionMixin, nn.Module):
    """VGG16 cfg-D up to conv5_3 without the last ReLU / pool (ibl/models/vgg.py:15-70)."""

    _fix_layers = {"conv5": 24, "conv4": 17, "conv3": 10, "conv2": 5, "full": 0}

    def __init__(self, depth, pretrained=True, cut_at_pooling=False, train_layers="conv5",
                 matconvnet=None):
        super().__init__()
        if depth != 16:
            raise KeyError("Unsupported depth:", depth)
        self.pretrained = pretrained
        self.depth = depth
        self.cut_at_pooling = cut_at_pooling
        self.train_layers = train_layers
        self.feature_dim = 512
        self.matconvnet = matconvnet
        layers, cin = [], 3
        for v in _CFG_D:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.base = nn.Sequential(*layers[:-2])   # drop the last ReLU and max-pool (vgg.py:41-42)
        self.gap = nn.AdaptiveMaxPool2d(1)
        self._cache: Dict = {}
        self.range_fallbacks = 0           # batches recomputed in bf16x3 because f16mx met a value beyond fp16
        self.precision_runs: Dict = {}     # effective precision -> backbone passes run in it
        self._hook_state_dict_loads()
        self._init_params()
        if not pretrained:
            self.reset_params()
        else:
            if matconvnet is None:
                warnings.warn("openibl_amd: ImageNet weights for vgg16 cannot be downloaded here; "
                              "the backbone is randomly initialised until a checkpoint is loaded")
                self.reset_params()
            for l in list(self.base.children())[: VGG._fix_layers[train_layers]]:
                for p in l.parameters():
                    p.requires_grad = False

    def _init_params(self):
        if self.matconvnet is not None:
            self.base.load_state_dict(torch.load(self.matconvnet))
            self.pretrained = True

    def reset_params(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ---- HIP path ---------------------------------------------------------------------------
    def _convs(self) -> List[nn.Conv2d]:
        return [m for m in self.base if isinstance(m, nn.Conv2d)]

    # Rounds 1-3 ran f16mx batches whose conv4 layers gave fewer than 256 ring tiles in bf16x3 (every Tokyo 24/7
    # query, every ragged last batch).  Since round 4 the f16mx ring kernels split K for layers that would leave
    # the chip idle (csrc/conv.hip, mx_split_plan) and a single 480x640 image runs f16mx in 0.82 ms (bf16x3:
    # 0.95).  Below 8 tiles of 256 conv4 pixels (one image of 320x320, two of 224x224) the split layers' extra
    # launches cost what the cheaper products save and bf16x3 — the other 1e-4 mode, three times as exact — is as
    # fast or faster (profiles/r04_l_small_sizes.md: 0.49 against 0.51 ms at 224x224, 0.44 against 0.49 at
    # 128x160; from 9 tiles on f16mx wins by 7 % and more), so THAT is what such a batch runs in.  In ring tiles
    # (256 pixels x 256 of the 512 channels): 16.  A knob: 0 = f16mx whenever it can run (the kernel tests),
    # 256 = rounds 1-3.
    F16MX_MIN_TILES = 16

    def effective_precision(self, x: torch.Tensor) -> str:
        """The arithmetic the backbone runs this input in: the module's precision, except for f16mx batches of fewer
        than F16MX_MIN_TILES conv4 ring tiles (where it is the slower of the two 1e-4 modes) and single images beyond
        the 32-bit offsets of the f16mx kernels, which run in bf16x3.  (A BATCH beyond those offsets — 95 images of
        480x640 and more — runs f16mx in image groups since round 6: `f16mx_groups`.)  `precision_runs` counts what
        actually ran."""
        p = self.precision
        if ops.precision_code(p) != ops.F16MX:
            return p
        n = int(x.shape[0])
        h, w = (int(x.shape[1]), int(x.shape[2])) if x.dtype == torch.uint8 else (int(x.shape[2]), int(x.shape[3]))
        tiles = -(-(n * (h // 8) * (w // 8)) // 256) * 2
        if self._f16mx_group(h, w) < 1:
            return "bf16x3"
        return p if tiles >= self.F16MX_MIN_TILES else "bf16x3"

    @staticmethod
    def _f16mx_group(h: int, w: int) -> int:
        """Images of h x w one f16mx pass takes: its kernels address their input through 32-bit buffer offsets and have
        no other implementation — the largest activation they read (conv2_2's input, [N][H/2][W/2][128] 4-byte
        elements: 39 MB per 480x640 image) and the stem's fp32 input must stay below 3.5 GB (vgg16_f16mx_fits in
        csrc/conv.hip is the same test; the C entry point refuses what this lets through): 94 images of 480x640."""
        per = max(3 * h * w * 4, (h // 2) * (w // 2) * 128 * 4, 1)
        return (0xE0000000 - 1) // per

    def f16mx_groups(self, x: torch.Tensor):
        """[(first image, images)] of the passes an f16mx batch is run in: one, unless the batch is beyond the 32-bit
        offsets of the kernels — then equal groups of at most `_f16mx_group` images (128 x 480x640: 2 x 64), each a
        pass of its own into its rows of the feature map, instead of the silent bf16x3 run of rounds 1-5."""
        n = int(x.shape[0])
        h, w = (int(x.shape[1]), int(x.shape[2])) if x.dtype == torch.uint8 else (int(x.shape[2]), int(x.shape[3]))
        cap = max(1, self._f16mx_group(h, w))
        parts = -(-n // cap)
        per = -(-n // parts)
        return [(i, min(per, n - i)) for i in range(0, n, per)]

    def _packed(self, device: torch.device, precision: str = None):
        precision = precision or self.precision
        convs = self._convs()
        params = [c.weight for c in convs] + [c.bias for c in convs]
        if params[0].device != device:
            raise RuntimeError(f"VGG: parameters are on {params[0].device}, input on {device}")
        key = _fingerprint(params)
        if self._cache.get("packed_key") != key:      # weights changed: every precision's packing is stale
            self._cache["packed_key"] = key
            self._cache["packed"] = {}
        hit = self._cache["packed"].get(precision)
        if hit is None:
            ws = [convs[0].weight.detach().float().contiguous()]
            ws += [ops.pack_conv3x3(c.weight.detach().float().contiguous(), precision) for c in convs[1:]]
            bs = [c.bias.detach().float().contiguous() for c in convs]
            hit = self._cache["packed"][precision] = (ws, bs)
        return hit

    def features_nhwc(self, x: torch.Tensor, defer_flag: bool = False) -> torch.Tensor:
        """[N][3][H][W] fp32 (normalised) or [N][H][W][3] uint8 (raw image; the loader's
        ToTensor + Normalize run inside the first kernel) -> conv5_3 map [N][h][w][512] in the
        precision's element type.

        f16mx range guard: the fp16 main term of f16mx exists only for |activation| <= 65504 (x 8: the backbone
        stores its activations times 2^-3).  The kernels raise a device flag when a layer's output is beyond that
        (include/openibl_amd.h, OIBL_F16MX) and a flagged batch is recomputed in bf16x3, the other mode inside the
        1e-4 tolerance, which has no range limit (`range_fallbacks` counts them).  WHERE the flag is read:
          * defer_flag=False (a bare call): here, one 4-byte copy — a host synchronisation behind the backbone;
          * defer_flag=True: not here.  The caller enqueues whatever consumes the map first and then calls
            `settle_range_flag(x)` (the models' forwards do: ONE synchronisation at the end of the whole forward,
            where the caller's `.cpu()` would wait anyway — round 6), or ships
            `last_range_flag()` to a pinned word and settles batches later (extract.EagerLanes, GraphedForward);
          * under a hipGraph capture nothing can be read: the capturer takes `last_range_flag()` and checks it
            after every replay (extract.GraphedForward does).
        `_range_flag` is per-module state of the LAST call: one thread and one stream per model object at a time."""
        if x.dtype != torch.uint8 and x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        prec = self.effective_precision(x)
        ws, bs = self._packed(x.device, prec)
        ev = getattr(self, "profile_events", None)
        self.precision_runs[prec] = self.precision_runs.get(prec, 0) + 1
        if ops.precision_code(prec) != ops.F16MX:
            self._range_flag = None
            return ops.vgg16_conv5(x, ws, bs, prec, events=ev)
        groups = self.f16mx_groups(x)
        if len(groups) == 1:
            feat, flag = ops.vgg16_conv5(x, ws, bs, prec, events=ev, return_flag=True)
        else:
            # a batch beyond the kernels' 32-bit offsets: image groups, each pass into its rows of one map; every
            # pass clears the workspace's flag word when it starts, so the groups' flags are OR-ed into one
            h, w = ops.vgg16_feature_hw(*((x.shape[1], x.shape[2]) if x.dtype == torch.uint8 else (x.shape[2], x.shape[3])))
            feat = torch.empty((int(x.shape[0]), h, w, 512), dtype=torch.float32, device=x.device)
            flag = torch.zeros(1, dtype=torch.int32, device=x.device)
            for i0, cnt in groups:
                _, f = ops.vgg16_conv5(x[i0:i0 + cnt], ws, bs, prec, return_flag=True, out=feat[i0:i0 + cnt])
                flag.bitwise_or_(f)
            self.precision_runs["f16mx(groups)"] = self.precision_runs.get("f16mx(groups)", 0) + len(groups)
        self._range_flag = flag
        if defer_flag or torch.cuda.is_current_stream_capturing():
            return feat
        fb = self.settle_range_flag(x)
        return feat if fb is None else fb

    def settle_range_flag(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """Read the range flag of the last `features_nhwc(x, defer_flag=True)` pass (a host synchronisation with the
        stream): None when the f16mx map stands, else the batch's map recomputed in bf16x3 — whatever was computed
        from the flagged map has to be recomputed from this one."""
        flag = getattr(self, "_range_flag", None)
        if flag is None or int(flag.item()) == 0:
            return None
        self.range_fallbacks += 1
        if x.dtype != torch.uint8 and x.dtype != torch.float32:
            x = x.float()
        return self.features_fallback(x.contiguous())

    def last_range_flag(self) -> Optional[torch.Tensor]:
        """The f16mx range flag of the last `features_nhwc` call (int32 [1] view of the first word of that
        stream's backbone workspace, rewritten by every later pass on the stream), None if that call did not
        run in f16mx."""
        return getattr(self, "_range_flag", None)

    def features_fallback(self, x: torch.Tensor) -> torch.Tensor:
        """The conv5_3 map of a batch whose f16mx pass raised the range flag: the same network in bf16x3."""
        ws, bs = self._packed(x.device, "bf16x3")
        self.precision_runs["bf16x3(range)"] = self.precision_runs.get("bf16x3(range)", 0) + 1
        return ops.vgg16_conv5(x.contiguous(), ws, bs, "bf16x3")

    def _outputs(self, feat):
        x_nchw = ops.nhwc_to_nchw_f32(feat)
        if self.cut_at_pooling:
            return x_nchw
        return ops.global_maxpool_nhwc(feat), x_nchw

    @torch.no_grad()
    def forward(self, x):
        return _with_range_guard(self, x, self._outputs)


def _with_range_guard(base: "VGG", x: torch.Tensor, head):
    """head(conv5_3 map) with the f16mx range flag settled BEHIND the head's launches: the device never waits for
    the host inside a forward, and the one synchronisation sits where the caller's first read of the result would
    wait anyway.  A flagged batch (rare: DESIGN §4.1a) has map and head recomputed in bf16x3."""
    out = head(base.features_nhwc(x, defer_flag=True))
    if torch.cuda.is_current_stream_capturing():
        return out
    fb = base.settle_range_flag(x)
    return out if fb is None else head(fb)


def vgg16(**kwargs):
    return VGG(16, **kwargs)


class NetVLAD(_PrecisionMixin, nn.Module):
    """NetVLAD layer (ibl/models/netvlad.py:8-61); forward returns the UN-normalised [N][K][C]."""

    def __init__(self, num_clusters=64, dim=512, alpha=100.0, normalize_input=True):
        super().__init__()
        self.num_clusters = num_clusters
        self.dim = dim
        self.alpha = alpha
        self.normalize_input = normalize_input
        self.conv = nn.Conv2d(dim, num_clusters, kernel_size=(1, 1), bias=False)
        self.centroids = nn.Parameter(torch.rand(num_clusters, dim), requires_grad=True)
        self.clsts = None
        self.traindescs = None
        self._cache: Dict = {}
        self._hook_state_dict_loads()

    def _init_params(self):
        raise NotImplementedError("NetVLAD._init_params (k-means initialisation for training) is "
                                  "outside the inference path this package implements")

    def _params(self):
        w = self.conv.weight.detach().float().reshape(self.num_clusters, self.dim).contiguous()
        return w, self.centroids.detach().float().contiguous()

    def aggregate_nhwc(self, feat: torch.Tensor, want_raw: bool, want_norm: bool):
        w, c = self._params()
        return ops.netvlad(feat, w, c, self.normalize_input, want_raw=want_raw, want_norm=want_norm)

    @torch.no_grad()
    def forward(self, x):
        """x: the backbone's [N][C][h][w] fp32 map (reference layout)."""
        feat = ops.nchw_f32_to_nhwc(x.float().contiguous(), self.precision)
        raw, _ = self.aggregate_nhwc(feat, want_raw=True, want_norm=False)
        return raw


class EmbedNet(_PrecisionMixin, nn.Module):
    """ibl/models/netvlad.py:63-82: returns (pool_x [N][512], vlad [N][K*C] intra+L2 normalised)."""

    def __init__(self, base_model, net_vlad):
        super().__init__()
        self.base_model = base_model
        self.net_vlad = net_vlad

    def _init_params(self):
        self.base_model._init_params()
        self.net_vlad._init_params()

    def _head(self, feat):
        _, vlad = self.net_vlad.aggregate_nhwc(feat, want_raw=False, want_norm=True)
        return ops.global_maxpool_nhwc(feat), vlad

    @torch.no_grad()
    def forward(self, x):
        return _with_range_guard(self.base_model, x, self._head)


class EmbedNetPCA(_PrecisionMixin, nn.Module):
    """ibl/models/netvlad.py:84-110: image batch -> [N][dim] unit-norm descriptors."""

    def __init__(self, base_model, net_vlad, dim=4096):
        super().__init__()
        self.base_model = base_model
        self.net_vlad = net_vlad
        self.pca_layer = nn.Conv2d(net_vlad.num_clusters * net_vlad.dim, dim, 1, stride=1, padding=0)
        self._cache: Dict = {}
        self._hook_state_dict_loads()

    def _init_params(self):
        self.base_model._init_params()
        self.net_vlad._init_params()

    def _pca_params(self):
        w, b = self.pca_layer.weight, self.pca_layer.bias
        key = (self.precision, _fingerprint([w, b]))
        hit = self._cache.get("pca")
        if hit is None or hit[0] != key:
            w2 = w.detach().float().reshape(w.shape[0], -1).contiguous()
            self._cache["pca"] = (key, ops.PcaWeight(ops.cast(w2, self.precision)), b.detach().float().contiguous())
            hit = self._cache["pca"]
        return hit[1], hit[2]

    def head_from_features(self, feat: torch.Tensor) -> torch.Tensor:
        """conv5_3 map (NHWC) -> descriptors: NetVLAD, intra + L2 norm, PCA, L2 (netvlad.py:98-108)."""
        _, vlad = self.net_vlad.aggregate_nhwc(feat, want_raw=False, want_norm=True)
        w, b = self._pca_params()
        return ops.pca(vlad, w, b, l2norm=True)

    @torch.no_grad()
    def forward(self, x):
        return _with_range_guard(self.base_model, x, self.head_from_features)

    def graphed(self, example: torch.Tensor, pipeline: bool = False):
        """hipGraph-replayed forward for a fixed batch shape (see GraphedDescriptor)."""
        return GraphedDescriptor(self, example, pipeline=pipeline)


def GraphedDescriptor(model: "EmbedNetPCA", example: torch.Tensor, pipeline: bool = False):
    """`EmbedNetPCA.forward` captured once into hipGraphs and replayed (openibl_amd.extract
    .GraphedForward): backbone (the matrix-core launches) and head (NetVLAD + PCA, ~10 launches).  A
    forward then costs the host two graph launches instead of ~30 kernel launches — on a shared host
    the eager path can become launch-bound at ~5 ms per batch.  Shapes, precision and parameters are
    frozen at capture time; the returned descriptor tensor is static and overwritten by a later call
    (clone it to keep it).

        fwd = model.graphed(example_batch)        # example_batch: resident [N][3][H][W] fp32
        desc = fwd(batch)                         # same shape; copied into the slot's static input
        desc = fwd()                              # again on the batch already in place

    pipeline=True keeps two slots (input, feature map, output each) and runs the head of batch i on
    a second stream while the backbone of batch i+1 already occupies the matrix cores; a batch passed
    to the call travels on a third (copy) stream.  The tensor returned by call i is then complete
    once `fwd.wait()` (or a device synchronisation) has returned, and is overwritten by call i+2."""
    from .extract import GraphedForward, _module_caches
    return GraphedForward(model.base_model.features_nhwc, model.head_from_features, example,
                          pipeline=pipeline, keep=_module_caches(model))


class EmbedRegionNet(_PrecisionMixin, nn.Module):
    """ibl/models/netvlad.py:112-207.  Only the evaluation branch (:199-205, identical to
    EmbedNet.forward) is implemented; the SFRS region-similarity training branch is out of scope."""

    def __init__(self, base_model, net_vlad, tuple_size=1):
        super().__init__()
        self.base_model = base_model
        self.net_vlad = net_vlad
        self.tuple_size = tuple_size

    def _init_params(self):
        self.base_model._init_params()
        self.net_vlad._init_params()

    @torch.no_grad()
    def forward(self, x):
        if self.training:
            raise NotImplementedError("EmbedRegionNet: the SFRS training branch is not part of the "
                                      "MI355X inference path; call .eval() first")
        return _with_range_guard(self.base_model, x, self._head)

    def _head(self, feat):
        _, vlad = self.net_vlad.aggregate_nhwc(feat, want_raw=False, want_norm=True)
        return ops.global_maxpool_nhwc(feat), vlad


_factory = {
    "vgg16": vgg16,
    "netvlad": NetVLAD,
    "embednet": EmbedNet,
    "embednetpca": EmbedNetPCA,
    "embedregionnet": EmbedRegionNet,
}


def names():
    return sorted(_factory.keys())


def create(name, *args, **kwargs):
    """Same factory contract as ibl/models/__init__.py:20-53."""
    if name not in _factory:
        raise KeyError("Unknown model:", name)
    return _factory[name](*args, **kwargs)
