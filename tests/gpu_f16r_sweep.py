"""Seed sweep of the f16r top-k against fp64 (diagnostic; the claim under test: the lists are those of a correctly
rounded fp32 matrix — every index equal to fp64's except where fp64 itself calls a near-tie, every value within fp32
rounding): random sizes, k, dimensions; unit rows, planted near-duplicates, rows of wildly different magnitude,
duplicated gallery rows (exact ties: the lower index wins).
    python tests/gpu_f16r_sweep.py [cases=60] [first seed=0] [wide]      (wide: k in 33..496, fp32 / fp16 / bf16 storage)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
fused = 0
t0 = time.time()
for seed in range(first, first + cases):
    g = torch.Generator().manual_seed(10_000 + seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))      # noqa: E731
    m, n = r(1, 700), r(8192, 36000)
    d = [256, 512, 1024, 2048, 4096][r(0, 4)]
    k = [1, 5, 10, 20, 32][r(0, 4)]
    store = torch.float32
    if len(sys.argv) > 3 and sys.argv[3] == "wide":   # round 6: k beyond the register rounds (bisection selection) and
        k = [33, 64, 120, 200, 496][r(0, 4)]         # descriptors STORED in 16 bits (rescored from the stored rows)
        store = [torch.float32, torch.float16, torch.bfloat16][r(0, 2)]
        n = max(n, 16 * 2048 if k > 120 else n)
    kind = ["unit", "hard", "scales", "dups"][seed % 4]
    if kind == "hard":
        q, gal, _, _ = synth.retrieval_problem(m, n, dim=d, seed=seed, hard_fraction=0.7)
    else:
        q = torch.nn.functional.normalize(torch.randn((m, d), generator=g), dim=1)
        gal = torch.nn.functional.normalize(torch.randn((n, d), generator=g), dim=1)
        if kind == "scales":                     # rows over eight orders of magnitude (per-row power-of-two scales)
            q = q * (10.0 ** (torch.rand((m, 1), generator=g) * 4 - 2))
            gal = gal * (10.0 ** (torch.rand((n, 1), generator=g) * 4 - 2))
        if kind == "dups":                       # every query's neighbourhood holds exact copies of gallery rows
            src = torch.randint(0, n, (n // 8,), generator=g)
            dst = torch.randint(0, n, (n // 8,), generator=g)
            gal[dst] = gal[src]
            q[: m // 2] = gal[torch.randint(0, n, (m // 2,), generator=g)] + 1e-3 * torch.randn((m // 2, d), generator=g)
    if store != torch.float32:                   # the problem IS the stored values (widened exactly)
        if kind == "scales" and store == torch.float16:
            q, gal = q.clamp(-6e4, 6e4), gal.clamp(-6e4, 6e4)
        qs, gs = q.to(store), gal.to(store)
        q, gal = qs.float(), gs.float()
    else:
        qs, gs = q, gal
    v, i, flag = ops.sqdist_topk(qs.to(dev), gs.to(dev), k, precision="f16r", defer_check=True)
    is_fused = bool(ops.f16r_fused(m, n, d, k))
    fused += int(is_fused)
    if int(flag.item()):                         # (overflow of a candidate list: the exact path answers)
        v, i = ops.sqdist_topk(qs.to(dev), gs.to(dev), k, precision="f16r")
        is_fused = False
    q64, g64 = q.double().to(dev), gal.double().to(dev)
    d64 = (q64 ** 2).sum(1)[:, None] + (g64 ** 2).sum(1)[None] - 2.0 * q64 @ g64.t()
    wv, wi = torch.sort(d64, dim=1, stable=True)
    wv, wi = wv[:, :k], wi[:, :k]
    got = torch.gather(d64, 1, i.long())
    scale = torch.maximum(wv.abs().amax(1, keepdim=True), (q64 ** 2).sum(1, keepdim=True)).clamp_min(1e-30)
    verr = float(((v.double() - got).abs() / scale).max())
    diff = i.long() != wi
    tie = float((((got - wv).abs() / scale)[diff]).max()) if bool(diff.any()) else 0.0
    # an index may differ from fp64's only where the two distances agree to fp32 rounding of the terms; values: the
    # rescoring's are correctly rounded up to the fp32 norms, the exact fp32 path (problems below 64 tiles, overflows)
    # carries the fp32 MFMA accumulation error (d 2^-24 of the terms)
    ok = verr <= (2e-6 if is_fused else 1e-5) and tie <= 2e-6 and bool((i >= 0).all())
    if kind == "dups" and ok:                    # exact ties: lists sorted by (distance, index)
        same = v[:, 1:] == v[:, :-1]
        ok = bool((i[:, 1:][same] > i[:, :-1][same]).all())
    bad += int(not ok)
    sname = str(store).replace("torch.", "")
    print(f"seed {seed:3d} {kind:6s} {m:3d} x {n:5d} x {d:4d} k={k:2d} {sname} {'fused' if is_fused else 'exact'} flag {int(flag.item())}: value err {verr:.1e}, "
          f"{int(diff.sum())} indices differ (fp64 gap {tie:.1e}) {'ok' if ok else 'FAILED'}", flush=True)
print(f"{cases} cases, {fused} on the fused path, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
