"""Product library against the debug library of this build when the latter was compiled with an experiment
macro (openibl_amd/build.py, DBG_EXPERIMENT_FLAGS): per-layer timings of the ring / halo kernels at batch 32 and
bit-identity of the results (diagnostic, not a pytest).     python tests/gpu_dbgvariant_ab.py [f16mx bf16 ...]

Experiments run this way: OIBL_MX_TAIL_B128 (profiles/r04_f_tail_ab.txt: not adopted); the counted fragment waits
(profiles/r04_h_lgkm_ab.txt: adopted — the product library has them now, -DOIBL_RING_LGKM0 in the debug library
gives the old lgkmcnt(0) schedule back, i.e. "variant" is then the OLD code)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops  # noqa: E402

LAYERS = [(64, 128, 240, 320, 1, 0), (128, 128, 240, 320, 1, 1), (128, 256, 120, 160, 1, 0), (256, 256, 120, 160, 1, 0),
          (256, 256, 120, 160, 1, 1), (256, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 0), (512, 512, 60, 80, 1, 1),
          (512, 512, 30, 40, 1, 0)]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)


def timed(fn, iters=4, rounds=5):
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


def _diff(a, b, prec):
    """rel-L2 between the two results when an experiment changes the summation order"""
    if torch.equal(a, b):
        return ""
    fa, fb = (ops.mx_join(a), ops.mx_join(b)) if prec == "f16mx" else (a.float(), b.float())
    return f" (rel-L2 {float((fa.double() - fb.double()).norm() / fb.double().norm()):.2e})"


PRECS = sys.argv[1:] or ["f16mx"]
for prec in PRECS:
  tot = [0.0, 0.0]
  for cin, cout, H, W, relu, pool in LAYERS:
      xf = torch.relu(torch.randn((32, H, W, cin), generator=g, device=dev)) * 3.0
      w = torch.randn((cout, cin, 3, 3), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
      b = torch.randn((cout,), generator=g, device=dev) * 0.1
      x = ops.mx_split(xf) if prec == "f16mx" else xf.to(torch.bfloat16) if prec == "bf16" else xf
      wp = ops.pack_conv3x3(w, prec)
      run = lambda: ops.conv3x3_nhwc(x, wp, b, bool(relu), bool(pool), prec)   # noqa: E731
      t, out = [], []
      for which in (0, 1, 0, 1):
          if which:
              lib.debug_hooks()
          else:
              lib.use_product_library()
          out.append(run())
          t.append(timed(run))
      lib.use_product_library()
      t0, t1 = min(t[0], t[2]), min(t[1], t[3])
      tot[0] += t0
      tot[1] += t1
      print(f"{cin:4d}->{cout:4d} {H:3d}x{W:3d}{' pool' if pool else '     '}: {prec} product {t0:6.3f} ms | variant {t1:6.3f} ms "
            f"({t0 / t1:4.2f}x) | same bits: {torch.equal(out[0], out[1])}{_diff(out[0], out[1], prec)}", flush=True)
  print(f"{prec}: layers behind the stem (conv5 once): {tot[0]:.3f} -> {tot[1]:.3f} ms")

# the fused distance + top-k kernels (match.hip shares ring_core.h)
Q, G, D, K = 8192, 81920, 4096, 10
q = torch.nn.functional.normalize(torch.randn((Q, D), generator=g, device=dev), dim=1)
gal = torch.nn.functional.normalize(torch.randn((G, D), generator=g, device=dev), dim=1)
for prec in PRECS:
    if prec not in ("f16mx", "bf16"):
        continue
    qp, gp = ops.PreparedRows(q, prec), ops.PreparedRows(gal, prec)
    t, res = [], []
    for which in (0, 1, 0, 1):
        if which:
            lib.debug_hooks()
        else:
            lib.use_product_library()
        res.append(ops.sqdist_topk_prepared(qp, gp, K))
        t.append(timed(lambda: ops.sqdist_topk_prepared(qp, gp, K, defer_check=True), iters=3))
    lib.use_product_library()
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    print(f"{prec} 8192 x 81920 x 4096 + top-10: product {min(t[0], t[2]):.3f} ms | variant {min(t[1], t[3]):.3f} ms | "
          f"same lists: {same}", flush=True)
