"""VERDICT r05 item 4 ("settle 0.40 with a measurement"): a measured LOWER bound for Winograd F(2x2, 3x3) in f16mx on
conv4_2 (512 -> 512 channels, 60x80 maps, batch 32) with the kernels this repository has, against the direct layer.

Winograd F(2x2, 3x3) replaces the 9-tap contraction by 16 independent products, one per position of the 4x4 transformed
tile:  M[p] = U[p] (tiles x Cin) . V[p]^T (Cout x Cin),  p = 0..15, tiles = N (H/2) (W/2) = 38 400 — 16 x 38400 x 512 x 512
MACs = 2.25x fewer than the direct 9 x 153600 x 512 x 512.  Whatever kernel computes them, the 16 products of a tile
meet again only in the inverse transform: with 128 accumulator registers per wave a 256 x 256 tile cannot hold 16
positions, so the products are MATERIALISED (16 x 38400 x 512 fp32 = 1.26 GB written, read once more).
This script times exactly that stage on the f16mx ring GEMM of the distance kernels (the same schedule as the
convolutions; K = 512 = 16 K-tiles, fp32 matrix out): 16 launches of [38400 x 512] x [512 x 512].  The input / weight
transforms, the f16mx re-split of the transformed lines and the inverse transform pass are NOT included — they only add.
If 16 products alone are not 1.2x faster than the direct layer, Winograd is settled for this design.
    python tests/gpu_winograd_bound.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
N, H, W, C = 32, 60, 80, 512


def timed(fn, iters=5, rounds=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return sorted(ts)[len(ts) // 2]


xf = torch.relu(torch.randn((N, H, W, C), generator=g, device=dev)) * 3.0
w = torch.randn((C, C, 3, 3), generator=g, device=dev) * (2.0 / (9 * C)) ** 0.5
b = torch.randn((C,), generator=g, device=dev) * 0.1
x = ops.mx_split(xf)
wp = ops.pack_conv3x3(w, "f16mx")
t_direct = timed(lambda: ops.conv3x3_nhwc(x, wp, b, True, False, "f16mx"))
flop_direct = 2.0 * 9 * N * H * W * C * C
print(f"direct conv4_2, f16mx, batch 32:              {t_direct:.3f} ms  ({flop_direct / t_direct / 1e9:.0f} TFLOP/s algorithmic)")

tiles = N * (H // 2) * (W // 2)
U = ops.PreparedRows(torch.randn((tiles, C), generator=g, device=dev), "f16mx")       # one position's transformed tiles
V = ops.PreparedRows(torch.randn((C, C), generator=g, device=dev) * 0.05, "f16mx")    # its transformed weights
M = torch.empty((tiles, C), device=dev)
lib = ops._lib.load()
ws = ops.workspace(lib.oibl_pairwise_st_workspace_bytes(tiles, C, C, ops.F16MX, 0, 0), dev, "wino")


def products():
    for _ in range(16):
        ops.pairwise_sqdist(U._source, V._source, "f16mx", out=M)      # (prepares + contracts: the preparation is the
                                                                        #  stand-in for the f16mx re-split of a transformed line)


def products_prepared_only():
    # the contraction alone on prepared operands: the fused top-k entry point with k = 1 is the cheapest epilogue the
    # repository has (no matrix written at all) — a bound from BELOW on any product stage
    for _ in range(16):
        ops.sqdist_topk_prepared(U, V, 1, exact=False, defer_check=True)


def prepare_only():
    for _ in range(16):
        ops.PreparedRows(U._source, "f16mx")
        ops.PreparedRows(V._source, "f16mx")


t_mat = timed(products)
t_prep = timed(prepare_only)
print(f"16 x [38400 x 512] x [512 x 512] f16mx, fp32 matrix out (+ operand preparation):  {t_mat:.3f} ms "
      f"({2.0 * 16 * tiles * C * C / t_mat / 1e9:.0f} TFLOP/s); 1.26 GB of products written")
print(f"  of which the operand preparation alone (16 x the f16mx split of a [38400 x 512] fp32 matrix):              {t_prep:.3f} ms")
print(f"  direct / (products + preparation) = {t_direct / t_mat:.2f}x;  direct / products alone = "
      f"{t_direct / (t_mat - t_prep):.2f}x  (the go / no-go bar of VERDICT r05 item 4: 1.2x — BEFORE the input / weight "
      f"transforms, the re-split of the transformed lines and the inverse-transform pass over 1.26 GB)")
