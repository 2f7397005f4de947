"""Are the two workgroups a CU holds of the 4-wave halo kernel (conv_halo4.h) in phase?  Every workgroup stamps its
start, the end of its K loop and its last store with the shader clock, and the CU / workgroup slot it ran on (test
hook, debug library).  Per CU: how much of a workgroup's epilogue lies inside the K loop of the CU's other workgroup
(the overlap the two-workgroup design is for) (diagnostic, not a pytest).  Round 5: 0.95-1.00 without any start-up
stagger — the two workgroups drift apart within the first tile.
    python tests/gpu_halo4_phase.py"""
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
for cin, cout, H, W, pool in [(64, 128, 240, 320, 0), (128, 128, 240, 320, 1)]:
    x = ops.mx_split(torch.relu(torch.randn((32, H, W, cin), generator=g, device=dev)) * 3.0)
    w = ops.pack_conv3x3(torch.randn((cout, cin, 3, 3), generator=g, device=dev) * 0.02, "f16mx")
    b = torch.zeros(cout, device=dev)
    tiles = 32 * 300
    for variant, name in ((0, "default dispatch"),):
        L.oibl_debug_set_mx_variant(variant)
        for _ in range(3):
            ops.conv3x3_nhwc(x, w, b, True, bool(pool), "f16mx")
        buf = torch.zeros(64 + 4 * tiles, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        L.oibl_debug_set_prof_buffer(buf.data_ptr())
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        ops.conv3x3_nhwc(x, w, b, True, bool(pool), "f16mx")
        e_.record()
        torch.cuda.synchronize()
        L.oibl_debug_set_prof_buffer(None)
        t = buf.cpu()[64:].view(tiles, 4).tolist()
        per_cu = defaultdict(list)
        tg = defaultdict(int)
        for s, hw, e, m in t:
            if s == 0:
                continue
            xcc, hwid = hw >> 32, hw & 0xffffffff
            cu = (xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15)
            per_cu[cu].append((s, m, e, (hwid >> 16) & 15, hwid & 15))
            tg[((hwid >> 16) & 15, hwid & 15)] += 1
        loop, epi, ov, conc = [], [], [], []
        for cu, v in per_cu.items():
            v.sort()
            for i, (s, m, e, _, _) in enumerate(v):
                loop.append(m - s)
                epi.append(e - m)
                # share of this epilogue [m, e) inside another workgroup's K loop [s2, m2) on the same CU
                inside = 0
                for j, (s2, m2, e2, _, _) in enumerate(v):
                    if j != i:
                        inside += max(0, min(e, m2) - max(m, s2))
                ov.append(inside / max(e - m, 1))
            span = v[-1][2] - v[0][0]
            conc.append(sum(e - s for s, m, e, _, _ in v) / max(span, 1))
        print(f"{cin}->{cout}{' pool' if pool else ''} [{name}]: {s_.elapsed_time(e_) * 1e3:.0f} us; {len(per_cu)} CUs, "
              f"workgroups resident per CU (time-average) {statistics.mean(conc):.2f}; K loop (with prologue) median "
              f"{statistics.median(loop):.0f} cycles, epilogue {statistics.median(epi):.0f}; share of an epilogue that lies "
              f"inside the other workgroup's K loop: median {statistics.median(ov):.2f}, mean {statistics.mean(ov):.2f}; "
              f"(TG_ID, WAVE_ID) seen: {dict(sorted(tg.items()))}", flush=True)
        if variant == 0:
            cu0 = sorted(per_cu)[0]
            t0 = per_cu[cu0][0][0]
            print("   one CU's first workgroups (start, loop end, end; relative cycles; TG_ID, WAVE_ID): " +
                  " ".join(f"[{s - t0} {m - t0} {e - t0} tg{tgid} w{wid}]" for s, m, e, tgid, wid in per_cu[cu0][:8]))
L.oibl_debug_set_mx_variant(0)
lib.use_product_library()
