#!/usr/bin/env python
"""Per-kernel MFMA utilisation, effective clock and L2 hit rate from the PMC passes of
tests/run_gpu_pmc_bench.sh  ->  profiles/<tag>_pmc.md

  MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles =
               GRBM_GUI_ACTIVE / 8 XCDs (MI355X_MICROARCH.md: the counter is summed over XCDs;
               SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16)
  clock      = kernel cycles / kernel duration (the chip clocks to its power budget)
  L2 hit     = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import argparse
import csv
import re
from collections import defaultdict
from pathlib import Path


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("oibl::", "")
    return name if len(name) <= 90 else name[:87] + "..."


def load(path):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    dur = defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            n[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc, n, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tag", default=None)
    a = ap.parse_args()
    d = Path(a.dir)
    tag = a.tag or d.name
    sq, n1, _ = load(d / "p1" / "pmc_counter_collection.csv")
    tc, n2, dur2 = load(d / "p2" / "pmc_counter_collection.csv")
    rows = []
    for k in tc:
        if "at::native" in k or "rocclr" in k or n2[k] == 0 or k not in sq:
            continue
        cyc = tc[k]["GRBM_GUI_ACTIVE"] / 8.0 / n2[k]
        us = dur2[k] / n2[k] / 1e3
        mfma = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / n1[k] / 1024.0
        hit, miss = tc[k]["TCC_HIT_sum"], tc[k]["TCC_MISS_sum"]
        rows.append((dur2[k], k, n2[k], us, cyc / us / 1e3 if us else 0.0, 100.0 * mfma / cyc if cyc else 0.0,
                     100.0 * hit / (hit + miss) if hit + miss else 0.0,
                     sq[k]["SQ_LDS_BANK_CONFLICT"] / n1[k]))
    rows.sort(reverse=True)
    lines = [f"# {tag}: MFMA utilisation, clock and L2 hit rate per kernel (rocprofv3 --pmc, eager bench step + matching)",
             "", __doc__.split("\n\n", 1)[1].rstrip(), "",
             "| kernel | launches | avg us | clock GHz | MFMA busy % | L2 hit % | LDS bank-conflict cycles |",
             "|---|---|---|---|---|---|---|"]
    for _, k, n, us, ghz, mf, hit, bc in rows[:22]:
        lines.append(f"| `{short(k)}` | {n} | {us:.1f} | {ghz:.2f} | {mf:.1f} | {hit:.1f} | {bc:.0f} |")
    out = Path(__file__).resolve().parent.parent / "profiles" / f"{tag}_pmc.md"
    out.write_text("\n".join(lines) + "\n")
    print(out.read_text())


if __name__ == "__main__":
    main()
