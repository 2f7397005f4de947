#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x) rocpd SQLite database into the per-kernel table that
`--stats` prints: calls, total / average / min / max duration, share of GPU time.

    python tools/rocpd_kernel_stats.py gpurun_out/prof/x_results.db [--md profiles/NAME.md]
"""
import argparse
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--md", default=None)
    ap.add_argument("--title", default="rocprofv3 --kernel-trace --stats")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = con.execute(
        f"select {namecol}, count(*), sum(end - start), avg(end - start), min(end - start), "
        f"max(end - start) from kernels group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"# {a.title}", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, av, mn, mx in rows:
        lines.append(f"| `{short(n)}` | {c} | {s / 1e6:.3f} | {av / 1e3:.2f} | {mn / 1e3:.2f} | "
                     f"{mx / 1e3:.2f} | {100.0 * s / total:.2f} |")
    text = "\n".join(lines) + "\n"
    if a.md:
        with open(a.md, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
