"""What a CU does between two tiles of a ring-kernel layer: every workgroup stamps its start and the end of its
epilogue (stores issued) with the shader clock and notes the CU it ran on (test hook; csrc/conv_ring.h); per CU the
gaps end(n) -> start(n + 1) are the drain of the tile's stores (s_endpgm waits for them) plus the launch of the next
workgroup (diagnostic, not a pytest).     python tests/gpu_wg_turnover.py"""
import statistics
import sys
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
L = lib.debug_hooks()
g = torch.Generator(device=dev).manual_seed(5)
LAYERS = [(64, 128, 240, 320, 0, 4800), (128, 128, 240, 320, 1, 4800), (512, 512, 60, 80, 0, 1200)]
for cin, cout, H, W, pool, tiles in LAYERS:
    x = ops.mx_split(torch.relu(torch.randn((32, H, W, cin), generator=g, device=dev)) * 3.0)
    w = ops.pack_conv3x3(torch.randn((cout, cin, 3, 3), generator=g, device=dev) * 0.02, "f16mx")
    b = torch.zeros(cout, device=dev)
    for _ in range(3):
        ops.conv3x3_nhwc(x, w, b, True, bool(pool), "f16mx")
    buf = torch.zeros(64 + 4 * tiles, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(buf.data_ptr())
    ops.conv3x3_nhwc(x, w, b, True, bool(pool), "f16mx")
    torch.cuda.synchronize()
    L.oibl_debug_set_prof_buffer(None)
    sec = buf.cpu()[:8].tolist()      # block 0 (first round: every CU in phase): the kernel's own section stamps
    print(f"  block 0: prologue {sec[0]}, main loop {sec[1]}, accumulators -> LDS {sec[2]}, rest of the epilogue {sec[3]} "
          f"(pass 0: pack {sec[4]}, copy-out {sec[5]}; pass 1: pack {sec[6]}, copy-out {sec[7]})")
    t = buf.cpu()[64:].view(tiles, 4)
    per_cu = defaultdict(list)
    main_, epi_ = [], []
    for s, hw, e, m in t.tolist():
        if s == 0:
            continue
        xcc, hwid = hw >> 32, hw & 0xffffffff
        cu = (xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15)      # xcc, se, sh, cu
        per_cu[cu].append((s, e))
        main_.append(m - s)
        epi_.append(e - m)
    dur, gaps = [], []
    for cu, v in per_cu.items():
        v.sort()
        dur += [e - s for s, e in v]
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    q = lambda a, f: sorted(a)[int(f * (len(a) - 1))]   # noqa: E731
    print(f"{cin}->{cout} {H}x{W}{' pool' if pool else ''}: {len(per_cu)} CUs, {sum(len(v) for v in per_cu.values())} workgroups, "
          f"{statistics.mean(len(v) for v in per_cu.values()):.2f} per CU | inside a workgroup (start -> stores issued) "
          f"median {statistics.median(dur):.0f} cycles (p10 {q(dur, .1)}, p90 {q(dur, .9)}) | gap to the next workgroup on the CU "
          f"median {statistics.median(gaps):.0f} (p10 {q(gaps, .1)}, p90 {q(gaps, .9)})\n  per workgroup: prologue + main loop median "
          f"{statistics.median(main_):.0f} (p10 {q(main_, .1)}, p90 {q(main_, .9)}), epilogue {statistics.median(epi_):.0f} "
          f"(p10 {q(epi_, .1)}, p90 {q(epi_, .9)})", flush=True)
lib.use_product_library()
