"""What the matrix pipe sustains with NOTHING else to do: 256 workgroups x 8 waves of back-to-back
v_mfma_f32_32x32x16_bf16 on register operands (oibl_debug_mfma_peak), run for several seconds with the
GPU clock / power from rocm-smi.  This is the power-capped ceiling the MFMA-bound kernels are measured
against.      python tests/gpu_mfma_peak.py [seconds] [out.md]"""
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openibl_amd import lib  # noqa: E402

dev = torch.device("cuda", 0)
L = lib.debug_hooks()
scratch = torch.zeros(64, dtype=torch.float32, device=dev)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
rows, stop = [], [False]


def smi():
    while not stop[0]:
        t = time.perf_counter()
        try:
            o = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True,
                               text=True, timeout=5).stdout
            d = next(iter(json.loads(o).values()))
            rows.append((t, {k: v for k, v in d.items() if any(s in k.lower() for s in ("sclk", "power"))}))
        except Exception as e:   # noqa: BLE001
            rows.append((t, {"error": repr(e)}))
        time.sleep(max(0.0, 1.0 - (time.perf_counter() - t)))


out, summary = [], []
for blocks, label in ((256, "256 workgroups x 8 waves (2 waves per SIMD)"), (512, "512 workgroups x 8 waves (4 waves per SIMD)")):
    iters = 20000
    flop = blocks * 8 * iters * 16 * 32768.0
    for _ in range(3):
        lib.check(L.oibl_debug_mfma_peak(iters, blocks, scratch.data_ptr(), torch.cuda.current_stream().cuda_stream), "peak")
    torch.cuda.synchronize()
    rows.clear()
    stop[0] = False
    th = threading.Thread(target=smi, daemon=True)
    th.start()
    t_start = time.perf_counter()
    rates = []
    while time.perf_counter() - t_start < seconds:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            lib.check(L.oibl_debug_mfma_peak(iters, blocks, scratch.data_ptr(), torch.cuda.current_stream().cuda_stream), "peak")
        b.record()
        torch.cuda.synchronize()
        rates.append((round(time.perf_counter() - t_start, 2), flop * 10 / (a.elapsed_time(b) * 1e-3) / 1e12))
    stop[0] = True
    th.join(timeout=3)
    line = (f"{label}: {min(r for _, r in rates):.0f}-{max(r for _, r in rates):.0f} TFLOP/s over {seconds:.0f} s "
            f"(last second: {rates[-1][1]:.0f})")
    print(line, flush=True)
    out.append("## " + line)
    out += ["", "| t (s) | TFLOP/s |", "|---|---|"] + [f"| {t} | {r:.0f} |" for t, r in rates[:: max(1, len(rates) // 12)]]
    out += ["", "| t (s) | rocm-smi |", "|---|---|"] + [f"| {t - t_start:.1f} | {d} |" for t, d in rows] + [""]
    for t, d in rows:
        print(f"   t={t - t_start:5.1f}s {d}", flush=True)
    summary.append({"config": label, "tflops_min": round(min(r for _, r in rates), 1),
                    "tflops_max": round(max(r for _, r in rates), 1),
                    "tflops_median": round(sorted(r for _, r in rates)[len(rates) // 2], 1),
                    "rocm_smi_mid_run": rows[len(rows) // 2][1] if rows else None})
if len(sys.argv) > 2:
    Path(sys.argv[2]).with_suffix(".json").write_text(json.dumps(
        {"what": "dense bf16 v_mfma_f32_32x32x16_bf16 on register operands (pseudo-random values), nothing else running; "
                 "sustained over %.0f s" % seconds, "runs": summary,
         "tflops": max(x["tflops_median"] for x in summary)}, indent=1) + "\n")
    Path(sys.argv[2]).write_text("# dense bf16 MFMA throughput with nothing else to do (tests/gpu_mfma_peak.py)\n\n" + "\n".join(out) + "\n")
