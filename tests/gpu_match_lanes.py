"""Experiment: the matching step (prepared operands, 8192 x 81920 x 4096, top-10) with the query rows
split over L streams (each with its own workspace) against one stream."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import ops, lib as _lib  # noqa: E402

dev = torch.device("cuda", 0)
Q, G, d, k = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 81920, 4096, 10
g = torch.nn.functional.normalize(torch.randn(G, d, device=dev), dim=1)
q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=1)
lib = _lib.load()


def run(prec, lanes, reps=20):
    gp, qp = ops.PreparedRows(g, prec), ops.PreparedRows(q, prec)
    p = qp.precision
    per = qp.shape[1] * (2 if prec == "bf16" else 4)
    qrows = qp.operand.contiguous().view(torch.uint8).reshape(-1)
    ov = torch.full((Q, k), float("inf"), dtype=torch.float32, device=dev)
    oi = torch.full((Q, k), -1, dtype=torch.int32, device=dev)
    flags = torch.zeros(lanes, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
    bounds = [(Q * i // lanes, Q * (i + 1) // lanes) for i in range(lanes)]
    wss = []
    for s, (a, b) in zip(streams, bounds):
        with torch.cuda.stream(s):
            wss.append(ops.workspace(lib.oibl_sqdist_topk_prepared_workspace_bytes(b - a, G, d, k, p), dev, "sqdist_topk"))
    torch.cuda.synchronize()

    def once():
        for i, (s, (a, b)) in enumerate(zip(streams, bounds)):
            with torch.cuda.stream(s):
                _lib.check(lib.oibl_sqdist_topk_prepared(qrows.data_ptr() + a * per, qp.norms.data_ptr() + 4 * a, b - a,
                                                         gp.operand.data_ptr(), gp.norms.data_ptr(), G, d, k, 0, p, 0,
                                                         ov.data_ptr() + a * k * 4, oi.data_ptr() + a * k * 4,
                                                         flags.data_ptr() + 4 * i, wss[i].data_ptr(), wss[i].numel(),
                                                         s.cuda_stream), "x")
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ref_v, ref_i = ops.sqdist_topk_prepared(qp, gp, k)
    ok = torch.equal(oi, ref_i) and torch.equal(ov, ref_v) and int(flags.sum()) == 0
    print(f"{prec}: {lanes} lane(s): {ms:.3f} ms  {Q * G / ms * 1e3:.3e} pairs/s  equal: {ok}", flush=True)


for prec in ("bf16", "bf16x3"):
    for lanes in (1, 2, 4, 1, 2):
        run(prec, lanes)
