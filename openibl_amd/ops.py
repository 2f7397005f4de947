"""Torch-facing wrappers over the C ABI: tensors in, tensors out, device pointers across.

PyTorch is used only for device memory, streams and dtype bookkeeping; every computation below is
a call into libopenibl_amd.so.  All tensors must live on a CUDA(HIP) device and be contiguous.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import lib as _lib

BF16 = 0
F32 = 1
BF16X3 = 2   # split bf16: (hi, lo) operand pairs, three MFMAs per product, fp32-class results
F16MX = 3    # fp16 main term + MX-fp6 cross terms: 1e-4-class results at half the matrix time of bf16x3
F16R = 4     # matching only (top-k entry points): fp16 filter pass + exact rescoring from the fp32 rows — fp32-exact
             # lists at the cost of a 2-byte operand stream (csrc/match_f16r.h); not a backbone / matrix arithmetic
# element containers: bf16x3 / f16mx activations and packed weights are opaque 4-byte elements (128-byte
# groups of 32 elements, see x3_split / mx_split) carried in int32 tensors of the logical shape
_DTYPES = {BF16: torch.bfloat16, F32: torch.float32, BF16X3: torch.int32, F16MX: torch.int32}
_NAMES = {"bf16": BF16, "bfloat16": BF16, "fp32": F32, "f32": F32, "float32": F32,
          "bf16x3": BF16X3, "x3": BF16X3, "f16mx": F16MX, "mx": F16MX, "f16r": F16R}


def precision_code(p) -> int:
    if isinstance(p, str):
        try:
            return _NAMES[p.lower()]
        except KeyError:
            raise ValueError(f"unknown precision {p!r} (use 'bf16', 'f16mx', 'bf16x3' or 'fp32')")
    if p in (BF16, F32, BF16X3, F16MX, F16R):
        return int(p)
    raise ValueError(f"unknown precision {p!r}")


F16R_MAX_FUSED_K = 496      # the fused f16r path: a member window of 2k + 32 <= 1024 slots (csrc/match.hip, f16r_plan)


def topk_precision(precision, storage_dtype=torch.float32, k: int = 10) -> int:
    """The arithmetic the fused distance + top-k of a model precision runs in: an f16mx model's descriptors are
    matched in f16r — fp16 filter pass + exact rescoring from the stored rows: fp32-exact lists, and faster than
    contracting every pair in f16mx — whatever their storage type (float32, float16, bfloat16: a 16-bit row is
    widened exactly) and for every ranked prefix the reference reads (Recall@1/5/10, and the 120 ranks of spatial
    NMS, ibl/evaluators.py:152-153); everything else as asked."""
    p = precision_code(precision)
    if p == F16MX and storage_dtype in (torch.float32, torch.float16, torch.bfloat16) and k <= F16R_MAX_FUSED_K:
        return F16R
    return p


def head_precision(p) -> int:
    """Precision of the small contractions behind the backbone (NetVLAD assignment, PCA): bf16x3
    runs them in exact fp32 — they are HBM / latency bound there (the PCA streams its weight once
    per batch), so splitting the operands would buy nothing."""
    p = precision_code(p)
    return F32 if p in (BF16X3, F16MX) else p


def elem_dtype(p) -> torch.dtype:
    return _DTYPES[precision_code(p)]


def _need_cuda(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.OpenIBLAmdError(
                "openibl_amd: tensor is on %s; this path runs only on an AMD GPU through the HIP "
                "extension (there is no CPU fallback)" % t.device)
        if not t.is_contiguous():
            raise ValueError("openibl_amd: tensors must be contiguous")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError("openibl_amd: tensors on different devices")
    return dev


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_WS = {}


def workspace(nbytes: int, dev: torch.device, slot: str = "default") -> torch.Tensor:
    """Grow-only per-(device, stream, slot) scratch buffer (torch allocations are 512-B aligned)."""
    key = (dev.index, _stream(dev), slot)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf


def release_workspaces() -> None:
    _WS.clear()


def workspaces_snapshot():
    """The scratch buffers currently alive (a captured hipGraph keeps those it recorded referenced:
    the grow-only buffers are replaced, not resized, when a later call needs more)."""
    return list(_WS.values())


# ---------------------------------------------------------------------------------------------
def cast(x: torch.Tensor, precision) -> torch.Tensor:
    """fp32 tensor -> tensor of the precision's element type (bf16 round-to-nearest-even)."""
    p = head_precision(precision)
    dev = _need_cuda(x)
    if x.dtype != torch.float32:
        raise ValueError("cast expects float32 input")
    if p == F32:
        return x
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.load().oibl_cast_f32_to_bf16(_ptr(x), _ptr(out), x.numel(), _stream(dev)),
               "cast_f32_to_bf16")
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    dev = _need_cuda(x)
    if x.dtype == torch.float32:
        return x
    if x.dtype != torch.bfloat16:
        raise ValueError("to_f32 expects bf16 or fp32")
    out = torch.empty(x.shape, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_cast_bf16_to_f32(_ptr(x), _ptr(out), x.numel(), _stream(dev)),
               "cast_bf16_to_f32")
    return out


def x3_split(x: torch.Tensor) -> torch.Tensor:
    """float32 [..., C] (C % 32 == 0) -> the bf16x3 operand layout (int32 container, same shape):
    per row, C/32 groups of [32 x hi | 32 x lo] with hi = bf16(v), lo = bf16(v - hi)."""
    dev = _need_cuda(x)
    if x.dtype != torch.float32 or x.dim() < 1 or x.shape[-1] % 32 != 0:
        raise ValueError("x3_split expects a float32 tensor whose last dimension is a multiple of 32")
    out = torch.empty(x.shape, dtype=torch.int32, device=dev)
    C_ = int(x.shape[-1])
    _lib.check(_lib.load().oibl_x3_split_rows(_ptr(x), _ptr(out), x.numel() // C_, C_, _stream(dev)),
               "x3_split_rows")
    return out


def new_range_flag(dev: torch.device) -> torch.Tensor:
    """A cleared f16mx range flag (int32 [1] on the device): producers of f16mx lines set it to 1 when they
    meet a value beyond fp16 (|v| > 65504), see include/openibl_amd.h, OIBL_F16MX."""
    return torch.zeros(1, dtype=torch.int32, device=dev)


def mx_split(x: torch.Tensor, range_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 [..., C] (C % 32 == 0) -> the f16mx operand layout (int32 container, same shape): per 32
    elements one 128-byte line [32 fp16 | e2m3 images of hi and lo + their scale bytes].  range_flag: optional
    int32 [1] device tensor, set to 1 when a value is beyond fp16."""
    dev = _need_cuda(x, range_flag)
    if x.dtype != torch.float32 or x.shape[-1] % 32 != 0:
        raise ValueError("mx_split expects a float32 tensor whose last dimension is a multiple of 32")
    C_ = int(x.shape[-1])
    out = torch.empty(x.shape, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().oibl_mx_split_rows_flagged(_ptr(x), _ptr(out), x.numel() // C_, C_, _ptr(range_flag),
                                                      _stream(dev)), "mx_split_rows")
    return out


def mx_join(x: torch.Tensor, which: int = 0) -> torch.Tensor:
    """f16mx tensor (int32 container) -> float32: which = 0 hi + q6(lo) (the stored value to ~2^-15 of
    the group's largest element), 1 hi (fp16), 2 q6(hi), 3 q6(lo) — what the kernels multiply."""
    dev = _need_cuda(x)
    if x.dtype != torch.int32 or x.shape[-1] % 32 != 0:
        raise ValueError("mx_join expects an int32 f16mx container whose last dimension is a multiple of 32")
    C_ = int(x.shape[-1])
    out = torch.empty(x.shape, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_mx_join_rows(_ptr(x), _ptr(out), x.numel() // C_, C_, int(which), _stream(dev)),
               "mx_join_rows")
    return out


def x3_join(x: torch.Tensor) -> torch.Tensor:
    """bf16x3 tensor (int32 container) -> float32 hi + lo (exact)."""
    dev = _need_cuda(x)
    if x.dtype != torch.int32 or x.dim() < 1 or x.shape[-1] % 32 != 0:
        raise ValueError("x3_join expects an int32 bf16x3 container whose last dimension is a multiple of 32")
    out = torch.empty(x.shape, dtype=torch.float32, device=dev)
    C_ = int(x.shape[-1])
    _lib.check(_lib.load().oibl_x3_join_rows(_ptr(x), _ptr(out), x.numel() // C_, C_, _stream(dev)),
               "x3_join_rows")
    return out


def set_regstage(on: bool) -> None:
    """Test hook: run the GEMM main loops register-staged instead of through global_load_lds."""
    _lib.debug_hooks().oibl_debug_set_regstage(1 if on else 0)


def set_conv_tile(mode: int) -> None:
    """Test hook: 0 = auto tile choice, 1 = 128-row tiles, 2 = 256x{128,64}, 3 = 256x256."""
    _lib.debug_hooks().oibl_debug_set_conv_tile(int(mode))


def set_ring_raster(mode: int) -> None:
    """Test hook: tile rasterisation of the ring convolutions (0 = contiguous ids per XCD, 1 = one
    N-tile per XCD, see xcd_tile in csrc/common.h).  Results do not depend on it."""
    _lib.debug_hooks().oibl_debug_set_ring_raster(int(mode))


def set_conv_korder(mode: int) -> None:
    """Test hook: K order of the implicit-GEMM convolutions, 0 = (tap, channel chunk), 1 = (channel
    chunk, tap): 3-8x fewer fetched bytes, slower on all layers but bf16x3 conv2_2; -1 = the per-layer
    default (order 1 for that layer, 0 elsewhere; see csrc/conv.hip)."""
    _lib.debug_hooks().oibl_debug_set_conv_korder(int(mode))


def set_conv_splitk(on: bool) -> None:
    """Test hook: allow / forbid split-K for the backbone layers whose tiling leaves the chip idle
    (small batches: conv4 / conv5 of a single image)."""
    _lib.debug_hooks().oibl_debug_set_conv_splitk(1 if on else 0)


def set_conv_c64(on) -> None:
    """Test hook: resident-weights kernel for Cin = 64 layers (bf16): False/0 = never, True/1 = auto
    (Cout = 64 only; wider layers go to the ring kernel), 2 = every Cin = 64 layer."""
    _lib.debug_hooks().oibl_debug_set_conv_c64(int(on))


def set_conv11_valu(on: bool) -> None:
    """Test hook: run conv1_1 on the vector ALU (exact fp32) also in bf16 mode."""
    _lib.debug_hooks().oibl_debug_set_conv11_valu(1 if on else 0)


# ---- backbone -------------------------------------------------------------------------------
VGG16_CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)  # vgg.py:40-42 conv positions
VGG16_CFG = ((3, 64, 1, 0), (64, 64, 1, 1), (64, 128, 1, 0), (128, 128, 1, 1), (128, 256, 1, 0),
             (256, 256, 1, 0), (256, 256, 1, 1), (256, 512, 1, 0), (512, 512, 1, 0),
             (512, 512, 1, 1), (512, 512, 1, 0), (512, 512, 1, 0), (512, 512, 0, 0))


def pack_conv3x3(w: torch.Tensor, precision) -> torch.Tensor:
    """[Cout][Cin][3][3] fp32 -> packed [9][Cout][Cin] in the precision's element type."""
    p = precision_code(precision)
    dev = _need_cuda(w)
    if w.dtype != torch.float32 or w.dim() != 4 or w.shape[2:] != (3, 3):
        raise ValueError("pack_conv3x3 expects a float32 [Cout][Cin][3][3] tensor")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    out = torch.empty((9, cout, cin), dtype=_DTYPES[p], device=dev)
    _lib.check(_lib.load().oibl_pack_conv3x3_weights(_ptr(w), cout, cin, p, _ptr(out),
                                                     _stream(dev)), "pack_conv3x3_weights")
    return out


def conv3x3_nhwc(x: torch.Tensor, packed_w: torch.Tensor, bias: torch.Tensor, relu: bool,
                 pool: bool, precision, range_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N][H][W][Cin] T -> [N][Ho][Wo][Cout] T (conv3x3 pad 1 + bias (+ReLU) (+2x2 max-pool)).
    range_flag (f16mx): optional int32 [1] device tensor, set to 1 when an output is beyond fp16."""
    p = precision_code(precision)
    dev = _need_cuda(x, packed_w, bias, range_flag)
    if x.dtype != _DTYPES[p] or packed_w.dtype != _DTYPES[p] or bias.dtype != torch.float32:
        raise ValueError("conv3x3_nhwc: dtype mismatch with precision")
    N, H, W, cin = map(int, x.shape)
    cout = int(packed_w.shape[1])
    if int(packed_w.shape[2]) != cin:
        raise ValueError("conv3x3_nhwc: packed weight Cin mismatch")
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty((N, Ho, Wo, cout), dtype=_DTYPES[p], device=dev)
    lib = _lib.load()
    # scratch for split-K partials: f16mx only here (its small-problem path); the bf16 / bf16x3 / fp32 layers
    # keep running one-pass when called alone — the backbone entry point gives them their scratch — so that
    # the tile-variant tests compare like with like
    ws_bytes = lib.oibl_conv3x3_workspace_bytes(N, H, W, cin, cout, int(pool), p) if p == F16MX else 0
    ws = workspace(ws_bytes, dev, "conv") if ws_bytes else None
    _lib.check(lib.oibl_conv3x3_nhwc_ws(_ptr(x), N, H, W, cin, _ptr(packed_w), _ptr(bias), cout, int(relu),
                                        int(pool), p, _ptr(out), _ptr(ws), ws_bytes, _ptr(range_flag),
                                        _stream(dev)), "conv3x3_nhwc")
    return out


def conv1_1_nchw(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, precision) -> torch.Tensor:
    """x [N][3][H][W] fp32 -> [N][H][W][64] T (conv1_1 + bias + ReLU)."""
    p = precision_code(precision)
    dev = _need_cuda(x, w, bias)
    if x.dtype != torch.float32 or w.dtype != torch.float32 or tuple(w.shape) != (64, 3, 3, 3):
        raise ValueError("conv1_1_nchw: expects fp32 x and a [64][3][3][3] fp32 weight")
    N, _, H, W = map(int, x.shape)
    out = torch.empty((N, H, W, 64), dtype=_DTYPES[p], device=dev)
    _lib.check(_lib.load().oibl_conv1_1_nchw(_ptr(x), N, H, W, _ptr(w), _ptr(bias), p, _ptr(out),
                                             _stream(dev)), "conv1_1_nchw")
    return out


def set_stem_fused(on: bool) -> None:
    """Test hook: use / bypass the fused conv1_1+conv1_2+pool stem kernel inside vgg16_conv5 (bf16)."""
    _lib.debug_hooks().oibl_debug_set_stem_fused(1 if on else 0)


def vgg16_stem(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, packed_w2: torch.Tensor,
               b2: torch.Tensor) -> torch.Tensor:
    """Fused bf16 stem: x [N][3][H][W] fp32 -> [N][H//2][W//2][64] bf16 (conv1_1, conv1_2, pool)."""
    dev = _need_cuda(x, w1, b1, packed_w2, b2)
    if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("vgg16_stem expects a float32 [N][3][H][W] tensor")
    if tuple(w1.shape) != (64, 3, 3, 3) or tuple(packed_w2.shape) != (9, 64, 64) \
            or packed_w2.dtype != torch.bfloat16:
        raise ValueError("vgg16_stem: conv1_1 must be [64][3][3][3] fp32, conv1_2 packed bf16 [9][64][64]")
    N, _, H, W = map(int, x.shape)
    out = torch.empty((N, H // 2, W // 2, 64), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.load().oibl_vgg16_stem_bf16(_ptr(x), N, H, W, _ptr(w1), _ptr(b1), _ptr(packed_w2),
                                                _ptr(b2), _ptr(out), _stream(dev)), "vgg16_stem_bf16")
    return out


def vgg16_stem_x3(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, packed_w2: torch.Tensor,
                  b2: torch.Tensor) -> torch.Tensor:
    """Fused bf16x3 stem: x [N][3][H][W] fp32 -> [N][H//2][W//2][64] split elements (int32 container)."""
    dev = _need_cuda(x, w1, b1, packed_w2, b2)
    if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("vgg16_stem_x3 expects a float32 [N][3][H][W] tensor")
    if tuple(w1.shape) != (64, 3, 3, 3) or tuple(packed_w2.shape) != (9, 64, 64) \
            or packed_w2.dtype != torch.int32:
        raise ValueError("vgg16_stem_x3: conv1_1 must be [64][3][3][3] fp32, conv1_2 packed bf16x3 [9][64][64]")
    N, _, H, W = map(int, x.shape)
    out = torch.empty((N, H // 2, W // 2, 64), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().oibl_vgg16_stem_x3(_ptr(x), N, H, W, _ptr(w1), _ptr(b1), _ptr(packed_w2),
                                              _ptr(b2), _ptr(out), _stream(dev)), "vgg16_stem_x3")
    return out


def vgg16_stem_mx(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, packed_w2: torch.Tensor,
                  b2: torch.Tensor) -> torch.Tensor:
    """Fused f16mx stem: x [N][3][H][W] fp32 -> [N][H//2][W//2][64] f16mx lines (int32 container);
    packed_w2 = pack_conv3x3(conv1_2, "f16mx")."""
    dev = _need_cuda(x, w1, b1, packed_w2, b2)
    if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("vgg16_stem_mx expects a float32 [N][3][H][W] tensor")
    if tuple(w1.shape) != (64, 3, 3, 3) or tuple(packed_w2.shape) != (9, 64, 64) \
            or packed_w2.dtype != torch.int32:
        raise ValueError("vgg16_stem_mx: conv1_1 must be [64][3][3][3] fp32, conv1_2 packed f16mx [9][64][64]")
    N, _, H, W = map(int, x.shape)
    out = torch.empty((N, H // 2, W // 2, 64), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().oibl_vgg16_stem_mx(_ptr(x), N, H, W, _ptr(w1), _ptr(b1), _ptr(packed_w2),
                                              _ptr(b2), _ptr(out), _stream(dev)), "vgg16_stem_mx")
    return out


def vgg16_feature_hw(H: int, W: int) -> Tuple[int, int]:
    for _ in range(4):
        H, W = H // 2, W // 2
    return H, W


# Normalize constants of the reference's loader (ibl/utils/data/__init__.py:40-41): std = 1/255,
# i.e. mean-subtracted 0..255 pixels
REF_MEAN = (0.48501960784313836, 0.4579568627450961, 0.4076039215686255)
REF_STD = (0.00392156862745098, 0.00392156862745098, 0.00392156862745098)


def vgg16_conv5(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                precision, events=None, mean=REF_MEAN, std=REF_STD, return_flag: bool = False,
                out: Optional[torch.Tensor] = None):
    """conv5_3 feature map [N][h][w][512] T (NHWC), h = H//16, w = W//16, of
      x [N][3][H][W] float32, already normalised (what the reference's loader hands over), or
      x [N][H][W][3] uint8, the raw decoded image: ToTensor + Normalize(mean, std) are folded into
        the first kernel (a quarter of the bytes over PCIe; bit-identical results in bf16 / fp32, within
        ~1e-6 / ~1e-5 of the fp32-input result in bf16x3 / f16mx: include/openibl_amd.h).

    weights[0] is the plain conv1_1 tensor, weights[1:] come from pack_conv3x3.
    `events`: optional pair of already-recorded torch.cuda.Event(enable_timing=True); they are
    re-recorded right before / after the matrix-core convolutions (bench.py's roofline).
    return_flag: also return the pass's f16mx range flag — an int32 [1] VIEW of the first word of this
    stream's backbone workspace (None in the other precisions): non-zero once the pass has run = an
    activation was beyond fp16 and `feat` must be recomputed in bf16x3 (models.VGG does).
    out: optional preallocated [N][h][w][512] map (a contiguous slice of a larger batch's map: models.VGG runs an
    f16mx batch beyond the kernels' 32-bit offsets in image groups).  The entry point clears the flag when a pass
    starts; a caller running several groups reads it between them (or accumulates it: models.VGG)."""
    p = precision_code(precision)
    dev = _need_cuda(x, *weights, *biases)
    u8 = x.dtype == torch.uint8
    if u8:
        if x.dim() != 4 or x.shape[3] != 3:
            raise ValueError("vgg16_conv5: uint8 input must be [N][H][W][3]")
        N, H, W, _ = map(int, x.shape)
    else:
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("vgg16_conv5 expects a float32 [N][3][H][W] or uint8 [N][H][W][3] tensor")
        N, _, H, W = map(int, x.shape)
    if len(weights) != 13 or len(biases) != 13:
        raise ValueError("vgg16_conv5 needs 13 weights and 13 biases")
    x = x.contiguous()
    h, w = vgg16_feature_hw(H, W)
    lib = _lib.load()
    ws_bytes = (lib.oibl_vgg16_u8_workspace_bytes if u8 else lib.oibl_vgg16_workspace_bytes)(N, H, W, p)
    if ws_bytes == 0:
        raise ValueError(f"vgg16_conv5: unsupported input shape {tuple(x.shape)}")
    ws = workspace(ws_bytes, dev, "vgg")
    # bf16x3: the last layer writes a plain fp32 map for the (fp32) head
    if out is not None:
        if tuple(out.shape) != (N, h, w, 512) or out.dtype != _DTYPES[head_precision(p)] or not out.is_contiguous() \
                or out.device != dev:
            raise ValueError("vgg16_conv5: `out` must be a contiguous [N][h][w][512] map of the head's element type")
        feat = out
    else:
        feat = torch.empty((N, h, w, 512), dtype=_DTYPES[head_precision(p)], device=dev)
    wp = (C.c_void_p * 13)(*[t.data_ptr() for t in weights])
    bp = (C.c_void_p * 13)(*[t.data_ptr() for t in biases])
    ev0 = ev1 = None
    if events is not None:
        ev0, ev1 = int(events[0].cuda_event), int(events[1].cuda_event)
    if u8:
        m3 = (C.c_float * 3)(*[float(v) for v in mean])
        s3 = (C.c_float * 3)(*[float(v) for v in std])
        _lib.check(lib.oibl_vgg16_conv5_forward_u8(_ptr(x), N, H, W, m3, s3, wp, bp, p, _ptr(feat),
                                                   _ptr(ws), ws.numel(), _stream(dev), ev0, ev1),
                   "vgg16_conv5_forward_u8")
    else:
        _lib.check(lib.oibl_vgg16_conv5_forward_ev(_ptr(x), N, H, W, wp, bp, p, _ptr(feat), _ptr(ws),
                                                   ws.numel(), _stream(dev), ev0, ev1),
                   "vgg16_conv5_forward")
    if return_flag:
        return feat, (ws[:4].view(torch.int32) if p == F16MX else None)
    return feat


def global_maxpool_nhwc(feat: torch.Tensor) -> torch.Tensor:
    """[N][h][w][C] T -> [N][C] fp32 (AdaptiveMaxPool2d(1))."""
    dev = _need_cuda(feat)
    p = BF16 if feat.dtype == torch.bfloat16 else F32
    N, C_ = int(feat.shape[0]), int(feat.shape[-1])
    P = feat.numel() // (N * C_)
    out = torch.empty((N, C_), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_global_maxpool_nhwc(_ptr(feat), N, P, C_, p, _ptr(out),
                                                    _stream(dev)), "global_maxpool_nhwc")
    return out


def nhwc_to_nchw_f32(feat: torch.Tensor) -> torch.Tensor:
    """[N][h][w][C] T -> [N][C][h][w] fp32."""
    dev = _need_cuda(feat)
    p = BF16 if feat.dtype == torch.bfloat16 else F32
    N, h, w, C_ = map(int, feat.shape)
    out = torch.empty((N, C_, h, w), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_nhwc_to_nchw_f32(_ptr(feat), N, h * w, C_, p, _ptr(out),
                                                 _stream(dev)), "nhwc_to_nchw_f32")
    return out


def nchw_f32_to_nhwc(x: torch.Tensor, precision) -> torch.Tensor:
    """[N][C][h][w] fp32 -> [N][h][w][C] T (bf16x3: fp32, what its head consumes)."""
    p = head_precision(precision)
    dev = _need_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("nchw_f32_to_nhwc expects a float32 [N][C][h][w] tensor")
    N, C_, h, w = map(int, x.shape)
    out = torch.empty((N, h, w, C_), dtype=_DTYPES[p], device=dev)
    _lib.check(_lib.load().oibl_nchw_f32_to_nhwc(_ptr(x), N, C_, h * w, p, _ptr(out),
                                                 _stream(dev)), "nchw_f32_to_nhwc")
    return out


# ---- NetVLAD ----------------------------------------------------------------------------------
def netvlad(feat: torch.Tensor, assign_w: torch.Tensor, centroids: torch.Tensor,
            normalize_input: bool = True, want_raw: bool = False, want_norm: bool = True):
    """feat [N][h][w][C] (or [N][P][C]) T -> (vlad_raw [N][K][C] | None, vlad_norm [N][K*C] | None)."""
    dev = _need_cuda(feat, assign_w, centroids)
    p = BF16 if feat.dtype == torch.bfloat16 else F32
    if feat.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("netvlad: feature map must be bf16 or fp32")
    N, C_ = int(feat.shape[0]), int(feat.shape[-1])
    P = feat.numel() // (N * C_)
    K = int(centroids.shape[0])
    aw = assign_w.reshape(K, C_)
    if aw.dtype != torch.float32 or centroids.dtype != torch.float32 or not aw.is_contiguous():
        raise ValueError("netvlad: assign_w / centroids must be contiguous float32")
    lib = _lib.load()
    ws_bytes = lib.oibl_netvlad_workspace_bytes(N, P, K, C_)
    ws = workspace(ws_bytes, dev, "netvlad")
    raw = torch.empty((N, K, C_), dtype=torch.float32, device=dev) if want_raw else None
    nrm = torch.empty((N, K * C_), dtype=torch.float32, device=dev) if want_norm else None
    _lib.check(lib.oibl_netvlad_forward(_ptr(feat), N, P, K, C_, p, _ptr(aw), _ptr(centroids),
                                        int(normalize_input), _ptr(raw), _ptr(nrm), _ptr(ws),
                                        ws.numel(), _stream(dev)), "netvlad_forward")
    return raw, nrm


# ---- PCA --------------------------------------------------------------------------------------
class PcaWeight:
    """A PCA weight [d][D] resident on the device, with — fp32 only — the re-packed copy the streaming kernel
    reads (oibl_pca_pack_weight; made on first use by a batch of 3 .. 32 rows, a second d * D * 4 bytes)."""

    def __init__(self, rows: torch.Tensor):
        if rows.dim() != 2 or rows.dtype not in (torch.bfloat16, torch.float32) or not rows.is_contiguous():
            raise ValueError("PcaWeight: a contiguous bf16 or fp32 [d][D] tensor")
        self.rows = rows
        self._packed = None

    def packed(self) -> torch.Tensor:
        if self._packed is None:
            w = self.rows
            if w.dtype != torch.float32:
                raise ValueError("PcaWeight.packed: fp32 weights only")
            dev = _need_cuda(w)
            d, D = map(int, w.shape)
            out = torch.empty_like(w)
            _lib.check(_lib.load().oibl_pca_pack_weight(_ptr(w), D, d, _ptr(out), _stream(dev)), "pca_pack_weight")
            self._packed = out
        return self._packed


PCA_STREAM_MIN_ROWS = 3      # one or two rows: the row-major streaming kernel (pca_small_kernel) is as fast


def pca(v: torch.Tensor, w, b: torch.Tensor, l2norm: bool = True) -> torch.Tensor:
    """normalize(W v + b): v [N][D] fp32, w [d][D] bf16|fp32 (selects the precision) or a PcaWeight, b [d] fp32."""
    holder = w if isinstance(w, PcaWeight) else None
    if holder is not None:
        w = holder.rows
    dev = _need_cuda(v, w, b)
    if w.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("pca: weight must be bf16 or fp32")
    p = BF16 if w.dtype == torch.bfloat16 else F32
    if v.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError("pca: v and b must be float32")
    N, D = map(int, v.shape)
    d = int(w.shape[0])
    if w.numel() != d * D:
        raise ValueError("pca: weight shape mismatch")
    lib = _lib.load()
    ws_bytes = lib.oibl_pca_workspace_bytes(N, D, d, p)
    ws = workspace(ws_bytes, dev, "pca")
    out = torch.empty((N, d), dtype=torch.float32, device=dev)
    if (holder is not None and p == F32 and N >= PCA_STREAM_MIN_ROWS and lib.oibl_pca_packed_supported(N, D, d)):
        _lib.check(lib.oibl_pca_forward_packed(_ptr(v), N, D, _ptr(holder.packed()), _ptr(b), d, int(l2norm),
                                               _ptr(out), _ptr(ws), ws.numel(), _stream(dev)), "pca_forward_packed")
        return out
    _lib.check(lib.oibl_pca_forward(_ptr(v), N, D, _ptr(w), _ptr(b), d, p, int(l2norm), _ptr(out),
                                    _ptr(ws), ws.numel(), _stream(dev)), "pca_forward")
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """Row-wise x / max(||x||, 1e-12) for a float32 [N][D] tensor."""
    dev = _need_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 2:
        raise ValueError("l2_normalize expects a float32 [N][D] tensor")
    out = torch.empty_like(x)
    if x.shape[0] == 0:
        return out
    _lib.check(_lib.load().oibl_l2_normalize_rows(_ptr(x), int(x.shape[0]), int(x.shape[1]),
                                                  _ptr(out), _stream(dev)), "l2_normalize_rows")
    return out


# ---- 16-bit descriptor storage (BASELINE.json configs[4]; no counterpart in the reference) ----------
ST_F32, ST_F16, ST_BF16 = 0, 1, 2
_ST_OF = {torch.float32: ST_F32, torch.float16: ST_F16, torch.bfloat16: ST_BF16}


def storage_code(t: torch.Tensor) -> int:
    try:
        return _ST_OF[t.dtype]
    except KeyError:
        raise ValueError("descriptors must be float32, float16 or bfloat16, not %s" % t.dtype)


def store_descriptors(x: torch.Tensor, dtype: Optional[torch.dtype]) -> torch.Tensor:
    """float32 descriptors -> their storage type (round-to-nearest-even), through the HIP casts."""
    if dtype is None or dtype == torch.float32:
        return x
    dev = _need_cuda(x)
    if x.dtype != torch.float32:
        raise ValueError("store_descriptors expects float32 input")
    if dtype == torch.bfloat16:
        return cast(x, BF16)
    if dtype != torch.float16:
        raise ValueError("descriptor storage must be float32, float16 or bfloat16")
    out = torch.empty(x.shape, dtype=torch.float16, device=dev)
    _lib.check(_lib.load().oibl_cast_f32_to_f16(_ptr(x), _ptr(out), x.numel(), _stream(dev)),
               "cast_f32_to_f16")
    return out


def load_descriptors(x: torch.Tensor) -> torch.Tensor:
    """Stored descriptors widened to float32 (exact)."""
    if x.dtype != torch.float16:
        return to_f32(x)
    dev = _need_cuda(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_cast_f16_to_f32(_ptr(x), _ptr(out), x.numel(), _stream(dev)),
               "cast_f16_to_f32")
    return out


def sum_l2_normalize(xs: torch.Tensor) -> torch.Tensor:
    """float32 [S][N][D] -> [N][D]: rows of xs[0] + xs[1] + ... L2-normalised."""
    dev = _need_cuda(xs)
    if xs.dtype != torch.float32 or xs.dim() != 3:
        raise ValueError("sum_l2_normalize expects a float32 [S][N][D] tensor")
    S, N, D = map(int, xs.shape)
    out = torch.empty((N, D), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_sum_l2_normalize(_ptr(xs), S, N, D, _ptr(out), _stream(dev)),
               "sum_l2_normalize")
    return out


def resize_bilinear(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """float32 [N][C][H][W] -> [N][C][H2][W2] with the arithmetic of
    F.interpolate(x, size=size, mode="bilinear", align_corners=False)."""
    dev = _need_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("resize_bilinear expects a float32 [N][C][H][W] tensor")
    N, Cc, H, W = map(int, x.shape)
    H2, W2 = int(size[0]), int(size[1])
    out = torch.empty((N, Cc, H2, W2), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_resize_bilinear_nchw(_ptr(x), N, Cc, H, W, _ptr(out), H2, W2,
                                                     _stream(dev)), "resize_bilinear")
    return out


# ---- matching ---------------------------------------------------------------------------------
def _pad_dim(x: torch.Tensor) -> torch.Tensor:
    """Descriptor rows zero-padded to the next multiple of 64 dimensions (the K-tile of the distance
    kernels): zeros change neither norms nor dot products, so e.g. PCA(pca_n_components=16)
    descriptors match like the reference's."""
    d = int(x.shape[1])
    if d % 64 == 0:
        return x
    out = torch.zeros((x.shape[0], d + (-d) % 64), dtype=x.dtype, device=x.device)
    out[:, :d] = x
    return out


def pairwise_sqdist(x: torch.Tensor, y: torch.Tensor, precision=F32,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dist[i][j] = |x_i|^2 + |y_j|^2 - 2 x_i.y_j for x [m][d], y [n][d] stored as float32, float16
    or bfloat16 (16-bit rows are widened exactly; see oibl_pairwise_sqdist_st)."""
    p = precision_code(precision)
    if p == F16R:
        raise ValueError("pairwise_sqdist: 'f16r' is a top-k arithmetic (filter + rescoring); a full matrix is "
                         "computed in 'f16mx', 'bf16x3' or 'fp32'")
    dev = _need_cuda(x, y)
    if out is not None and (not out.is_cuda or out.dtype != torch.float32 or out.dim() != 2
                            or out.stride(1) != 1 or out.device != dev):
        raise ValueError("pairwise_sqdist: `out` must be a float32 CUDA matrix with unit column stride")
    if x.dim() != 2 or y.dim() != 2:
        raise ValueError("pairwise_sqdist expects [m][d] and [n][d]")
    xs, ys = storage_code(x), storage_code(y)
    if int(y.shape[1]) != int(x.shape[1]):
        raise ValueError("pairwise_sqdist: dimension mismatch")
    x, y = _pad_dim(x), _pad_dim(y)
    m, d = map(int, x.shape)
    n = int(y.shape[0])
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=dev)
    if m == 0 or n == 0:
        return out
    lib = _lib.load()
    ws_bytes = lib.oibl_pairwise_st_workspace_bytes(m, n, d, p, xs, ys)
    ws = workspace(ws_bytes, dev, "pairwise")
    _lib.check(lib.oibl_pairwise_sqdist_st(_ptr(x), xs, m, _ptr(y), ys, n, d, p, _ptr(out),
                                           int(out.stride(0)), _ptr(ws), ws.numel(), _stream(dev)),
               "pairwise_sqdist")
    return out


def row_topk(vals: torch.Tensor, k: int, index_base: int = 0,
             idx_in: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """k smallest per row, ascending, lowest index first on ties -> (values [m][k], idx [m][k] int32)."""
    dev = _need_cuda(vals, idx_in)
    if vals.dtype != torch.float32 or vals.dim() != 2:
        raise ValueError("row_topk expects a float32 [m][n] tensor")
    if idx_in is not None and (idx_in.dtype != torch.int32 or idx_in.shape != vals.shape):
        raise ValueError("row_topk: idx_in must be int32 with the shape of vals")
    m, n = map(int, vals.shape)
    ov = torch.empty((m, k), dtype=torch.float32, device=dev)
    oi = torch.empty((m, k), dtype=torch.int32, device=dev)
    if m == 0:
        return ov, oi
    _lib.check(_lib.load().oibl_row_topk(_ptr(vals), _ptr(idx_in), m, n, int(vals.stride(0)), k,
                                         int(index_base), _ptr(ov), _ptr(oi), _stream(dev)),
               "row_topk")
    return ov, oi


def row_argsort(vals: torch.Tensor, want_values: bool = False, max_ws_bytes: int = 1 << 31):
    """Stable ascending argsort of every row of a float32 [m][n] device matrix -> int32 [m][n]
    (and the sorted values): torch.argsort(vals, dim=1, stable=True) on the GPU (oibl_row_argsort).
    Rows are sorted in groups so that the ping-pong workspace (16 bytes per element) stays below
    `max_ws_bytes`."""
    if not vals.is_cuda:
        raise _lib.OpenIBLAmdError("openibl_amd: row_argsort runs only on an AMD GPU (there is no CPU fallback)")
    dev = vals.device
    if vals.dtype != torch.float32 or vals.dim() != 2 or (vals.shape[1] > 1 and vals.stride(1) != 1):
        raise ValueError("row_argsort expects a float32 [m][n] tensor with unit column stride")
    m, n = map(int, vals.shape)
    idx = torch.empty((m, n), dtype=torch.int32, device=dev)
    sv = torch.empty((m, n), dtype=torch.float32, device=dev) if want_values else None
    if m == 0 or n == 0:
        return (idx, sv) if want_values else idx
    lib = _lib.load()
    rows = max(1, min(m, int(max_ws_bytes // (16 * n + 1024))))
    ws = workspace(lib.oibl_row_argsort_workspace_bytes(rows, n), dev, "argsort")
    for r0 in range(0, m, rows):
        r = min(rows, m - r0)
        _lib.check(lib.oibl_row_argsort(vals[r0:].data_ptr(), r, n, int(vals.stride(0)),
                                        idx[r0:].data_ptr(), None if sv is None else sv[r0:].data_ptr(),
                                        _ptr(ws), ws.numel(), _stream(dev)), "row_argsort")
    return (idx, sv) if want_values else idx


def first_hit_rank(topk_idx: torch.Tensor, gt_offsets: torch.Tensor, gt_values: torch.Tensor,
                   gallery_pids: Optional[torch.Tensor] = None, nms_window: int = 120) -> torch.Tensor:
    """Per query: rank of the first prediction that is a ground-truth neighbour (-1: none), with
    the reference's spatial NMS when gallery_pids is given.  All tensors int32 on the device."""
    dev = _need_cuda(topk_idx, gt_offsets, gt_values, gallery_pids)
    for t in (topk_idx, gt_offsets, gt_values, gallery_pids):
        if t is not None and t.dtype != torch.int32:
            raise ValueError("first_hit_rank expects int32 tensors")
    m, k = map(int, topk_idx.shape)
    if int(gt_offsets.numel()) != m + 1:
        raise ValueError("first_hit_rank: gt_offsets must have m + 1 entries")
    out = torch.empty((m,), dtype=torch.int32, device=dev)
    if m == 0:
        return out
    _lib.check(_lib.load().oibl_first_hit_rank(_ptr(topk_idx.contiguous()), m, k, _ptr(gt_offsets),
                                               _ptr(gt_values), _ptr(gallery_pids), int(nms_window),
                                               _ptr(out), _stream(dev)), "first_hit_rank")
    return out


def flag_to_host(flag_dev: torch.Tensor, ring: torch.Tensor, index: int) -> None:
    """The int32 device word `flag_dev` -> word `index` of the PINNED host tensor `ring`, by a kernel on the
    current stream (oibl_copy_words: no DMA-engine copy to queue behind a large input transfer).  The host
    reads ring[index] once an event recorded behind this call has fired."""
    dev = _need_cuda(flag_dev)
    if not ring.is_pinned() or ring.dtype != torch.int32 or flag_dev.dtype != torch.int32:
        raise ValueError("flag_to_host: ring must be a pinned int32 host tensor, the flag an int32 device tensor")
    _lib.check(_lib.load().oibl_copy_words(_ptr(flag_dev), ring.data_ptr() + 4 * int(index), 1, _stream(dev)),
               "copy_words")


def event_elapsed_ms(start: "torch.cuda.Event", stop: "torch.cuda.Event") -> float:
    """hipEventElapsedTime on the raw handles — works for events recorded by the event nodes of a
    replayed hipGraph, which torch's own bookkeeping does not see."""
    ms = C.c_float(0.0)
    _lib.check(_lib.load().oibl_event_elapsed_ms(int(start.cuda_event), int(stop.cuda_event),
                                                       C.byref(ms)), "event_elapsed_ms")
    return float(ms.value)


def set_match_ring(mode: int) -> None:
    """Test hook: ring-schedule distance kernel 0 = never, 1 = auto, 2 = whenever legal."""
    _lib.debug_hooks().oibl_debug_set_match_ring(int(mode))


def set_match_splitk(on: bool) -> None:
    """Test hook: allow / forbid the 2-way split-K contraction of the fused path's threshold sample."""
    _lib.debug_hooks().oibl_debug_set_match_splitk(1 if on else 0)


def sqdist_topk(x: torch.Tensor, y: torch.Tensor, k: int, index_base: int = 0, precision=F32,
                exact: bool = False, defer_check: bool = False):
    """k nearest rows of y (squared L2) for every row of x, without materialising the matrix:
    (values [m][k] ascending, indices [m][k] int32 = index_base + row of y).  Equals
    row_topk(pairwise_sqdist(x, y), k).  The fused bf16 path reports candidate-list overflow through
    a device flag; it is read here (one 4-byte copy) and the call repeated on the exact path.
    defer_check=True skips that host synchronisation and returns (values, indices, flag) with the
    int32 device flag: the caller reads it when it synchronises anyway and repeats with exact=True
    if it is set (sharded.sharded_topk ships it with the per-shard lists)."""
    p = precision_code(precision)
    dev = _need_cuda(x, y)
    if x.dim() != 2 or y.dim() != 2:
        raise ValueError("sqdist_topk expects [m][d] and [n][d]")
    if p == F16R:
        if x.shape[0] == 0 or y.shape[0] == 0:
            p = F32                                           # (nothing to prepare: the empty result below)
        else:
            return sqdist_topk_prepared(PreparedRows(x, F16R), PreparedRows(y, F16R), k, index_base=index_base,
                                        exact=exact, defer_check=defer_check)
    xs, ys = storage_code(x), storage_code(y)
    if int(y.shape[1]) != int(x.shape[1]):
        raise ValueError("sqdist_topk: dimension mismatch")
    x, y = _pad_dim(x), _pad_dim(y)
    m, d = map(int, x.shape)
    n = int(y.shape[0])
    ov = torch.full((m, k), float("inf"), dtype=torch.float32, device=dev)
    oi = torch.full((m, k), -1, dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    if m == 0 or n == 0:
        return (ov, oi, flag) if defer_check else (ov, oi)
    lib = _lib.load()
    x, y = x.contiguous(), y.contiguous()
    ws_bytes = lib.oibl_sqdist_topk_st_workspace_bytes(m, n, d, k, p, xs, ys)
    ws = workspace(ws_bytes, dev, "sqdist_topk")

    def run(ex: int) -> None:
        _lib.check(lib.oibl_sqdist_topk_st(_ptr(x), xs, m, _ptr(y), ys, n, d, k, int(index_base), p, ex,
                                           _ptr(ov), _ptr(oi), _ptr(flag), _ptr(ws), ws.numel(),
                                           _stream(dev)), "sqdist_topk")

    if defer_check:
        run(1 if exact else 0)
        return ov, oi, flag
    for ex in ([1] if exact else [0, 1]):
        run(ex)
        if ex == 1 or int(flag.item()) == 0:
            break
    return ov, oi


def cluster_means(x: torch.Tensor, labels: torch.Tensor, centers: torch.Tensor) -> torch.Tensor:
    """Lloyd update step (oibl_cluster_means): centers[c] <- mean of the rows x[i] with labels[i] == c,
    in place (fp64 accumulation in point order, correctly rounded); a centre without points keeps its
    value.  Returns the int32 member counts [num_clusters]."""
    dev = _need_cuda(x, labels, centers)
    if x.dtype != torch.float32 or centers.dtype != torch.float32 or x.dim() != 2 or centers.dim() != 2 \
            or x.shape[1] != centers.shape[1] or not x.is_contiguous() or not centers.is_contiguous():
        raise ValueError("cluster_means expects contiguous float32 x [n][d] and centers [K][d]")
    lab = labels.reshape(-1).to(torch.int32).contiguous()
    if lab.numel() != x.shape[0]:
        raise ValueError("cluster_means: one label per row")
    counts = torch.zeros((centers.shape[0],), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().oibl_cluster_means(_ptr(x), _ptr(lab), int(x.shape[0]), int(x.shape[1]),
                                              int(centers.shape[0]), _ptr(centers), _ptr(counts), _stream(dev)),
               "cluster_means")
    return counts


class PreparedRows:
    """A descriptor matrix ready for matching in one precision: the rows the contraction reads and
    their fp32 squared norms (oibl_match_prepare).  Build it once for a matrix that is matched many
    times — the resident gallery shard — and hand it to sqdist_topk / sharded_topk in place of the
    tensor: the norm / operand pass (8 % of a 8192 x 81920 matching step) leaves the call."""

    def __init__(self, x: torch.Tensor, precision):
        p = precision_code(precision)
        dev = _need_cuda(x)
        if x.dim() != 2:
            raise ValueError("PreparedRows expects [rows][d]")
        x = _pad_dim(x.contiguous())
        st = storage_code(x)
        rows, d = map(int, x.shape)
        self.precision, self.shape, self.device = p, (rows, d), dev
        self.norms = torch.empty((rows,), dtype=torch.float32, device=dev)
        self.operand = x
        self.aux = None
        if p == F16R:
            # four parts: scaled fp16 rows (the filter pass), {2^-e, |x|, |residual|, 0} per row, the fp32 squared
            # norms, and the STORED rows themselves (fp32, fp16 or bf16: read again — widened exactly — by the rescoring)
            self._source = x
            self.operand = torch.empty((rows, d), dtype=torch.float16, device=dev)
            self.aux = torch.empty((rows, 4), dtype=torch.float32, device=dev)
            if rows:
                _lib.check(_lib.load().oibl_match_prepare_f16r_st(_ptr(x), st, rows, d, _ptr(self.norms), _ptr(self.aux),
                                                                  _ptr(self.operand), _stream(dev)), "match_prepare_f16r")
            return
        if rows == 0:
            return
        lib = _lib.load()
        nbytes = lib.oibl_match_operand_bytes(rows, d, p, st)
        buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
        _lib.check(lib.oibl_match_prepare(_ptr(x), st, rows, d, p, _ptr(self.norms), _ptr(buf),
                                          _stream(dev)), "match_prepare")
        self._source = x          # read in place when there is no operand copy
        if buf is not None:
            self.operand = buf

    @property
    def storage(self) -> int:
        """OIBL_ST_* of the rows an f16r rescoring reads (the stored descriptors)."""
        return storage_code(self._source) if self._source is not None else ST_F32

    def operand_rows(self) -> torch.Tensor:
        """The operand as a [rows][bytes per row] uint8 matrix (a view): what travels when prepared
        queries are exchanged between ranks instead of fp32 rows."""
        rows = self.shape[0]
        if self.precision == F16R:
            # one row per query: [d fp16 | 4 fp32 aux | d stored elements] — what a rank that extracted the query ships
            d = self.shape[1]
            es = self._source.element_size()
            return torch.cat([self.operand.view(torch.uint8).reshape(rows, 2 * d),
                              self.aux.view(torch.uint8).reshape(rows, 16),
                              self._source.view(torch.uint8).reshape(rows, es * d)], dim=1)
        per = self.shape[1] * (2 if self.precision == BF16 else 4)
        flat = self.operand.contiguous().view(torch.uint8).reshape(-1)
        return flat[: rows * per].view(rows, per)

    def rows(self, lo: int, hi: int) -> "PreparedRows":
        """Rows [lo, hi) as prepared rows of their own — views of every part, no copies (a query block of
        sharded_topk; ADVICE r05: operand_rows() + from_parts concatenated and re-split the whole f16r set)."""
        n = self.shape[0]
        lo, hi = max(0, min(int(lo), n)), max(0, min(int(hi), n))
        out = type(self).__new__(type(self))
        out.precision, out.device = self.precision, self.device
        out.shape = (hi - lo, self.shape[1])
        out.norms = self.norms[lo:hi]
        out.aux = None if self.aux is None else self.aux[lo:hi]
        out._source = None if self._source is None else self._source[lo:hi]
        if self.precision == F16R or self.operand.dim() == 2:
            out.operand = self.operand[lo:hi]
        else:                                     # a flat byte buffer of whole rows (oibl_match_prepare's operand)
            per = self.shape[1] * (2 if self.precision == BF16 else 4)
            out.operand = self.operand.view(torch.uint8).reshape(-1)[lo * per: hi * per]
        return out

    @classmethod
    def from_parts(cls, operand_rows: torch.Tensor, norms: torch.Tensor, d: int, precision,
                   source_dtype: Optional[torch.dtype] = None) -> "PreparedRows":
        """Re-assemble prepared rows from their exchanged parts (operand_rows() and .norms of one or
        several PreparedRows of the same precision, concatenated along dim 0).  f16r: `source_dtype` = the
        storage type of the rows behind the fp16 image (default float32; a 2-byte tail without it: float16)."""
        self = cls.__new__(cls)
        self.precision = precision_code(precision)
        self.device = operand_rows.device
        self.shape = (int(operand_rows.shape[0]), int(d) + (-int(d)) % 64)
        self.norms = norms.contiguous()
        self._source = None
        self.aux = None
        if self.precision == F16R:
            rows, dd = self.shape
            b = operand_rows
            tail = (int(b.shape[1]) - 2 * dd - 16) // dd
            if source_dtype is None:
                source_dtype = torch.float32 if tail == 4 else torch.float16
            if tail != (4 if source_dtype == torch.float32 else 2):
                raise ValueError("PreparedRows.from_parts: row width does not match the f16r layout of %s rows" % source_dtype)
            self.operand = b[:, : 2 * dd].contiguous().view(torch.float16).reshape(rows, dd)
            self.aux = b[:, 2 * dd: 2 * dd + 16].contiguous().view(torch.float32).reshape(rows, 4)
            self._source = b[:, 2 * dd + 16:].contiguous().view(source_dtype).reshape(rows, dd)
            return self
        self.operand = operand_rows.contiguous()
        return self


def sqdist_topk_prepared(x: "PreparedRows", y: "PreparedRows", k: int, index_base: int = 0,
                         exact: bool = False, defer_check: bool = False):
    """sqdist_topk on prepared operands (same results, bit for bit)."""
    if x.precision != y.precision or x.shape[1] != y.shape[1]:
        raise ValueError("sqdist_topk_prepared: operands prepared for different precisions / dimensions")
    dev, p = x.device, x.precision
    (m, d), n = x.shape, y.shape[0]
    ov = torch.full((m, k), float("inf"), dtype=torch.float32, device=dev)
    oi = torch.full((m, k), -1, dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    if m == 0 or n == 0:
        return (ov, oi, flag) if defer_check else (ov, oi)
    lib = _lib.load()
    if p == F16R:
        xs, ys = x.storage, y.storage
        ws = workspace(lib.oibl_sqdist_topk_f16r_st_workspace_bytes(m, n, d, k, xs, ys), dev, "sqdist_topk")
        xsrc, ysrc = x._source.contiguous(), y._source.contiguous()
        xop, xaux, xnorm = x.operand.contiguous(), x.aux.contiguous(), x.norms.contiguous()

        def run(ex: int) -> None:
            _lib.check(lib.oibl_sqdist_topk_f16r_st(_ptr(xop), _ptr(xaux), _ptr(xnorm), _ptr(xsrc), xs, m,
                                                    _ptr(y.operand), _ptr(y.aux), _ptr(y.norms), _ptr(ysrc), ys, n,
                                                    d, k, int(index_base), ex, _ptr(ov), _ptr(oi), _ptr(flag),
                                                    _ptr(ws), ws.numel(), _stream(dev)), "sqdist_topk_f16r")
    else:
        ws = workspace(lib.oibl_sqdist_topk_prepared_workspace_bytes(m, n, d, k, p), dev, "sqdist_topk")

    def run_generic(ex: int) -> None:
        _lib.check(lib.oibl_sqdist_topk_prepared(_ptr(x.operand), _ptr(x.norms), m, _ptr(y.operand),
                                                 _ptr(y.norms), n, d, k, int(index_base), p, ex, _ptr(ov),
                                                 _ptr(oi), _ptr(flag), _ptr(ws), ws.numel(), _stream(dev)),
                   "sqdist_topk_prepared")

    if p != F16R:
        run = run_generic
    if defer_check:
        run(1 if exact else 0)
        return ov, oi, flag
    for ex in ([1] if exact else [0, 1]):
        run(ex)
        if ex == 1 or int(flag.item()) == 0:
            break
    return ov, oi


# ---- the two stages of the f16r top-k, for gallery-sharded matching (sharded.py) ----------------------------
def f16r_members(k: int) -> int:
    """Member slots per query of an f16r selection (K2)."""
    return int(_lib.load().oibl_f16r_members(int(k)))


def f16r_fused(m: int, n: int, d: int, k: int) -> bool:
    return bool(_lib.load().oibl_f16r_fused(int(m), int(n), int(d), int(k)))


def f16r_filter_select(x: "PreparedRows", y: "PreparedRows", k: int, index_base: int = 0):
    """Stage 1 (sample thresholds, fp16 filter pass, selection): (lval [m][K2] filter distances, lidx [m][K2] int32
    global indices (-1 paddings), ymax [2] = the gallery's largest row norm and fp16 residual norm, flag [1])."""
    if x.precision != F16R or y.precision != F16R:
        raise ValueError("f16r_filter_select expects operands prepared for 'f16r'")
    dev = x.device
    (m, d), n = x.shape, y.shape[0]
    K2 = f16r_members(k)
    lval = torch.empty((m, K2), dtype=torch.float32, device=dev)
    lidx = torch.empty((m, K2), dtype=torch.int32, device=dev)
    ymax = torch.zeros(2, dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws = workspace(lib.oibl_sqdist_topk_f16r_workspace_bytes(m, n, d, k), dev, "sqdist_topk")
    _lib.check(lib.oibl_f16r_filter_select(_ptr(x.operand), _ptr(x.aux), _ptr(x.norms), m, _ptr(y.operand), _ptr(y.aux),
                                           _ptr(y.norms), n, d, k, int(index_base), _ptr(lval), _ptr(lidx), _ptr(ymax),
                                           _ptr(flag), _ptr(ws), ws.numel(), _stream(dev)), "f16r_filter_select")
    return lval, lidx, ymax, flag


def f16r_keep_members(lval: torch.Tensor, lidx: torch.Tensor, k: int, thr: torch.Tensor, x: "PreparedRows",
                      ymax_all: torch.Tensor) -> None:
    """In place: entries of lidx whose filter distance exceeds thr[row] + 2 eps (eps: x's rows against the maxima
    over all shards, ymax_all [W][2]) become -1."""
    dev = _need_cuda(lval, lidx, thr, ymax_all)
    m = int(lval.shape[0])
    _lib.check(_lib.load().oibl_f16r_keep_members(_ptr(lval), _ptr(lidx), m, int(k), _ptr(thr.contiguous()), _ptr(x.norms),
                                                  _ptr(x.aux), _ptr(ymax_all.contiguous()), int(ymax_all.shape[0]),
                                                  int(x.shape[1]), _stream(dev)), "f16r_keep_members")


def f16r_rescore(x: "PreparedRows", y: "PreparedRows", lidx: torch.Tensor, k: int, index_base: int = 0):
    """Stage 2: exact distances of the listed members, the k smallest (distance, index) per query."""
    dev = x.device
    m, d = x.shape
    ov = torch.full((m, k), float("inf"), dtype=torch.float32, device=dev)
    oi = torch.full((m, k), -1, dtype=torch.int32, device=dev)
    if m == 0 or y.shape[0] == 0:
        return ov, oi
    lidx = lidx.contiguous()
    _lib.check(_lib.load().oibl_f16r_rescore_st(_ptr(x._source.contiguous()), x.storage, _ptr(x.norms.contiguous()), m,
                                                _ptr(y._source), y.storage, _ptr(y.norms), d, int(k),
                                                int(lidx.shape[1]), int(index_base), _ptr(lidx), _ptr(ov), _ptr(oi),
                                                _stream(dev)), "f16r_rescore")
    return ov, oi


def gemm_nt(a: torch.Tensor, b: torch.Tensor, regstage: bool = False) -> torch.Tensor:
    """Diagnostic: C = A . B^T on the MFMA core; A [M][K], B [N][K] both bf16 or both fp32."""
    dev = _need_cuda(a, b)
    if a.dtype != b.dtype or a.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("gemm_nt: operands must both be bf16 or both fp32")
    p = BF16 if a.dtype == torch.bfloat16 else F32
    M, K = map(int, a.shape)
    N = int(b.shape[0])
    c = torch.empty((M, N), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().oibl_gemm_nt(_ptr(a), M, _ptr(b), N, K, p | (0x100 if regstage else 0),
                                        _ptr(c), N, _stream(dev)), "gemm_nt")
    return c
