"""Timing of the matching path at the benchmark size (diagnostic, not a pytest).

    python tests/gpu_matchbench.py [--q 8192] [--g 81920] [--d 4096] [--k 10] [--iters 3]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--q", type=int, default=8192)
    ap.add_argument("--g", type=int, default=81920)
    ap.add_argument("--d", type=int, default=4096)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--groups", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    q = torch.nn.functional.normalize(torch.randn((a.q, a.d), generator=gen, device=dev), dim=1)
    g = torch.nn.functional.normalize(torch.randn((a.g, a.d), generator=gen, device=dev), dim=1)
    fl = 2.0 * a.q * a.g * a.d
    out = torch.empty((a.q, a.g), device=dev)
    rows = []
    if a.only in ("", "topk"):
        t = timed(lambda: ops.sqdist_topk(q, g, a.k, precision="bf16"), a.iters)
        rows.append(("sqdist_topk bf16 (fused)", t))
    if a.only in ("", "prepared") or a.only.startswith("prepared:"):
        # resident gallery + prepared queries (bench.py's matching step without the collectives), per arithmetic
        for prec in (a.only.split(":")[1].split(",") if ":" in a.only else ["bf16", "f16mx", "f16r"]):
            gp, qp = ops.PreparedRows(g, prec), ops.PreparedRows(q, prec)
            t = timed(lambda: ops.sqdist_topk_prepared(qp, gp, a.k, defer_check=True), a.iters)
            rows.append((f"sqdist_topk_prepared {prec}", t))
            t = timed(lambda: ops.PreparedRows(q, prec), a.iters)
            rows.append((f"  PreparedRows(queries) {prec}", t))
            _, _, flag = ops.sqdist_topk_prepared(qp, gp, a.k, defer_check=True)
            print(f"  [{prec}] overflow flag {int(flag.item())}")
            del gp, qp
    if a.only in ("", "matrix"):
        t = timed(lambda: ops.pairwise_sqdist(q, g, "bf16", out=out), a.iters)
        rows.append(("pairwise_sqdist bf16 (ring, matrix written)", t))
        t = timed(lambda: ops.row_topk(out, a.k), a.iters)
        rows.append(("row_topk over the matrix", t))
    if a.only in ("", "generic"):
        ops.set_match_ring(0)
        t = timed(lambda: ops.pairwise_sqdist(q, g, "bf16", out=out), a.iters)
        ops.set_match_ring(1)
        rows.append(("pairwise_sqdist bf16 (generic 128x128 kernel)", t))
    if a.groups:
        from openibl_amd import lib as _l
        for gm in [int(v) for v in a.groups.split(",")]:
            _l.debug_hooks().oibl_debug_set_match_group(gm)
            t = timed(lambda: ops.pairwise_sqdist(q, g, "bf16", out=out), a.iters)
            rows.append((f"pairwise ring, group_m={gm}", t))
        _l.debug_hooks().oibl_debug_set_match_group(8)
    for n, t in rows:
        print(f"  {n:48s} {t:8.3f} ms  {fl / t / 1e9:8.1f} TFLOP/s-equivalent  {a.q * a.g / t / 1e6:9.1f} Gpairs/s")


if __name__ == "__main__":
    main()
