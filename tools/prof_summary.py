#!/usr/bin/env python
"""Condense one tests/run_gpu_round2.sh (round 1: run_gpu_round.sh) output directory (gpurun_out/<tag>/) into the tracked
summaries under profiles/:  <tag>_bench.json, <tag>_kernel_stats.md, <tag>_hbm_traffic.md,
<tag>_timing.txt.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in
KiB and collected in separate --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced read, so it is doubled ("corrected" column).

    python tools/prof_summary.py gpurun_out/r01_b [--tag r01_b]
"""
import argparse
import csv
import re
import shutil
from collections import defaultdict
from pathlib import Path


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("oibl::", "")
    return name if len(name) <= 110 else name[:107] + "..."


def kernel_stats(path):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:24]:
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | "
                   f"{float(r['AverageNs']) / 1e3:.1f} | {int(r['MinNs']) / 1e3:.1f} | "
                   f"{int(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return out


def counter_avg(path, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tag", default=None)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    d = Path(a.dir)
    tag = a.tag or d.name
    prof = Path(__file__).resolve().parent.parent / "profiles"
    prof.mkdir(exist_ok=True)
    if (d / "bench.json").exists():
        shutil.copy(d / "bench.json", prof / f"{tag}_bench.json")
    with open(prof / f"{tag}_timing.txt", "w") as f:
        if (d / "bench_torchrun.json").exists():
            f.write("==== python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 "
                    "--steps 10 --warmup 2 (RCCL group with one rank)\n"
                    + "\n".join(l for l in (d / "bench_torchrun.json").read_text().splitlines()
                                if l.startswith("{")) + "\n\n")
        if (d / "bench_2ranks_shared.json").exists():
            f.write("==== OIBL_BENCH_SHARED_GPU=1 python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 "
                    "--steps 5 --warmup 2: FLOW CHECK of the N = 2 launch line, two ranks time-sharing the one GPU "
                    "over gloo (not a performance number)\n"
                    + "\n".join(l[:600] for l in (d / "bench_2ranks_shared.json").read_text().splitlines()
                                if l.startswith("{")) + "\n\n")
        for n in ("gpu.txt", "precbench.log", "matchbench.log", "halo4_phase.log", "bar1_ab.log", "stem_mx.log", "mx_stamps.log", "mx_probe.log", "timing_bf16x3.log", "timing_bf16.log", "timing_bf16x3_raster1.log",
                  "timing_bf16_raster1.log", "timing_fp32.log", "pcie.log", "convbench.log", "shardbench.log", "head_bench.log", "pca_bench.log", "scale_probe.log"):
            if (d / n).exists():
                f.write(f"==== {n}\n" + (d / n).read_text() + "\n")
        if (d / "pytest_gpu.log").exists():
            f.write("==== pytest -m gpu (tail)\n"
                    + "\n".join((d / "pytest_gpu.log").read_text().splitlines()[-4:]) + "\n")
        if (d / "smoke.log").exists():
            f.write("==== smoke\n" + "\n".join((d / "smoke.log").read_text().splitlines()[-2:]) + "\n")
    if (d / "latency.md").exists():
        (prof / f"{tag}_latency.md").write_text(
            f"# {tag}: single-image latency (tests/gpu_latency.py; medians / minima over 40 device-synchronised calls)\n\n"
            + (d / "latency.md").read_text())
    if (d / "mfma_peak.md").exists():
        shutil.copy(d / "mfma_peak.md", prof / f"{tag}_mfma_peak.md")
        if (d / "mfma_peak.json").exists():
            shutil.copy(d / "mfma_peak.json", prof / "mfma_peak_latest.json")
    sustain_md(d, prof, tag)
    skip = "--no-pipeline --skip-matching --skip-cpu-baseline --skip-api --skip-fast-mode"
    for sub, name, title in (
            ("prof_stats", f"{tag}_kernel_stats.md", "python bench.py --steps 40 --warmup 5 --skip-matching --skip-cpu-baseline"),
            ("prof_stats_f16mx", f"{tag}_kernel_stats_f16mx.md", f"python bench.py --precision f16mx --steps 40 --warmup 5 {skip}"),
            ("prof_stats_bf16x3", f"{tag}_kernel_stats_bf16x3.md", f"python bench.py --precision bf16x3 --steps 40 --warmup 5 {skip}"),
            ("prof_stats_bf16", f"{tag}_kernel_stats_bf16.md", f"python bench.py --precision bf16 --steps 40 --warmup 5 {skip}"),
            ("prof_match", f"{tag}_kernel_stats_matching.md", "python tests/gpu_matchbench.py --only prepared:f16r,bf16 --iters 3 (8192 x 81920 x 4096-d + top-10, resident prepared operands)")):
        p = d / sub / "bench_kernel_stats.csv"
        if p.exists():
            lines = [f"# {tag}: rocprofv3 --kernel-trace --stats -- {title}", "", a.note, ""] + kernel_stats(p)
            (prof / name).write_text("\n".join(lines) + "\n")
    digest = {}
    for prec in ("", "f16mx", "bf16x3", "bf16"):
        sfx = f"_{prec}" if prec else ""
        fp = d / f"prof_fetch{sfx}" / "bench_counter_collection.csv"
        wp = d / f"prof_write{sfx}" / "bench_counter_collection.csv"
        if fp.exists() and wp.exists():
            ent = traffic_table(fp, wp, prof, tag, prec)
            if ent:
                digest[prec or "bf16"] = ent
    # the matching step (tests/gpu_matchbench.py --only prepared:f16r,bf16): bytes per launch of the filter kernels
    fp, wp = d / "prof_fetch_match" / "bench_counter_collection.csv", d / "prof_write_match" / "bench_counter_collection.csv"
    if fp.exists() and wp.exists():
        fe, wr = counter_avg(fp, "FETCH_SIZE"), counter_avg(wp, "WRITE_SIZE")
        lines = [f"# {tag}: HBM traffic per launch of the matching step's kernels, 8192 x 81920 x 4096-d (rocprofv3 --pmc "
                 "FETCH_SIZE / --pmc WRITE_SIZE, separate passes over tests/gpu_matchbench.py --only prepared:f16r,bf16)", "",
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md; MB = 1e6 bytes.  Algorithmic operand bytes of one step: "
                 "(8192 + 81920) rows x 4096 x 2 B = 738 MB read once (+ 8192 x ~14 x 16 KB = 1.9 GB of fp32 rows gathered by "
                 "the f16r rescoring).", "",
                 "| kernel | launches | fetch MB (raw) | fetch MB (corrected) | write MB | avg us | corrected GB/s |",
                 "|---|---|---|---|---|---|---|"]
        for k, (n, v, t) in sorted(fe.items(), key=lambda kv: -kv[1][2]):
            if n == 0 or "at::native" in k or "rocclr" in k:
                continue
            f_raw = v / n * 1024 / 1e6
            w = wr.get(k, [1, 0.0, 1])
            w_mb = w[1] / max(w[0], 1) * 1024 / 1e6
            us = t / n / 1e3
            lines.append(f"| `{short(k)}` | {n} | {f_raw:.1f} | {2 * f_raw:.1f} | {w_mb:.1f} | {us:.1f} | "
                         f"{(2 * f_raw + w_mb) / us * 1e3:.0f} |")
            for key, pat in (("matching_f16r", "pairwise_f16r_kernel<true"), ("matching_bf16", "pairwise_ring_kernel<true, 0")):
                if pat in k:
                    digest[key] = {"source": f"profiles/{tag}_hbm_traffic_matching.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                             "separate passes; FETCH doubled per MI355X_MICROARCH.md; the filter kernel)",
                                   "bytes_per_launch": (2 * f_raw + w_mb) * 1e6}
        (prof / f"{tag}_hbm_traffic_matching.md").write_text("\n".join(lines) + "\n")
    if digest:
        import json
        latest = prof / "hbm_traffic_latest.json"
        try:
            old = json.loads(latest.read_text())
        except Exception:
            old = {}
        old.update(digest)        # precisions not measured in this set keep their last digest
        latest.write_text(json.dumps(old, indent=1) + "\n")
    print("wrote", sorted(p.name for p in prof.glob(f"{tag}_*")))


def sustain_md(d, prof, tag):
    import json
    out = []
    for prec in ("f16mx", "bf16x3", "bf16"):
        p = d / f"sustain_{prec}.json"
        if not p.exists():
            continue
        try:
            s = json.loads([l for l in p.read_text().splitlines() if l.startswith("{")][-1])
        except Exception:
            continue
        out += [f"## {prec}: `python bench.py --sustain 12 --precision {prec}` — {s['seconds']} s of back-to-back "
                f"pipelined replays of the timed step (batch 32, 480x640)", "",
                f"images/s per ~1 s chunk: min {s['min']}, mean {s['mean']}, max {s['max']}", "",
                "| t (s) | images/s |", "|---|---|"]
        out += [f"| {t} | {r} |" for t, r in s["images_per_s_per_chunk"]]
        out += ["", "| t (s) | sclk | socket power (W) | junction temp (C) |", "|---|---|---|---|"]
        for r in s["rocm_smi"]:
            out.append(f"| {r.get('t')} | {r.get('sclk clock speed:', '')} | "
                       f"{r.get('Current Socket Graphics Package Power (W)', '')} | "
                       f"{r.get('Temperature (Sensor junction) (C)', '')} |")
        out.append("")
    if out:
        (prof / f"{tag}_sustain.md").write_text(
            f"# {tag}: sustained throughput with GPU clock / power (rocm-smi sampled once per second)\n\n"
            + "\n".join(out) + "\n")


def traffic_table(fp, wp, prof, tag, prec):
    if True:
        fe, wr = counter_avg(fp, "FETCH_SIZE"), counter_avg(wp, "WRITE_SIZE")
        lines = [f"# {tag} {prec}: HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
                 "",
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced read); "
                 "MB = 1e6 bytes; GB/s over the kernel's own duration in the FETCH pass.", "",
                 "| kernel | launches | fetch MB (raw) | fetch MB (corrected) | write MB | avg us | "
                 "corrected (fetch+write) GB/s |",
                 "|---|---|---|---|---|---|---|"]
        for k, (n, v, t) in sorted(fe.items(), key=lambda kv: -kv[1][2]):
            if n == 0 or "at::native" in k or "rocclr" in k:
                continue
            f_raw = v / n * 1024 / 1e6
            w = wr.get(k, [1, 0.0, 1])
            w_mb = w[1] / max(w[0], 1) * 1024 / 1e6
            us = t / n / 1e3
            lines.append(f"| `{short(k)}` | {n} | {f_raw:.1f} | {2 * f_raw:.1f} | {w_mb:.1f} | {us:.1f} | "
                         f"{(2 * f_raw + w_mb) / us * 1e3:.0f} |")
        (prof / (f"{tag}_hbm_traffic_{prec}.md" if prec else f"{tag}_hbm_traffic.md")).write_text("\n".join(lines) + "\n")
        # machine-readable digest for bench.py's roofline.traffic: corrected HBM bytes of the
        # matrix-core convolution launches, averaged per launch
        tot_b, tot_n, forwards = 0.0, 0, 0
        for k, (n, v, t) in fe.items():
            if not any(tag_ in k for tag_ in ("conv3x3_ring_kernel", "conv3x3_halo_kernel", "conv3x3_halo4_kernel", "mx_pack_rows_kernel", "vgg_stem_kernel",
                                              "vgg_stem_x3_kernel", "conv3x3_igemm_kernel", "conv3x3_c64_kernel",
                                              "conv_mx_splitk_reduce_kernel", "conv_mx_splitk_reduce8_kernel",
                                              "conv_splitk_reduce_kernel")):
                continue
            w = wr.get(k, [1, 0.0, 1])
            tot_b += 2 * v * 1024 + w[1] * 1024 * (n / max(w[0], 1))
            tot_n += n
            if "vgg_stem" in k:
                forwards += n          # one fused stem launch per forward
        if tot_n:
            name = f"{tag}_hbm_traffic_{prec}.md" if prec else f"{tag}_hbm_traffic.md"
            ent = {"source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                             "FETCH doubled per MI355X_MICROARCH.md; counts Infinity-Cache hits)",
                   "conv_launches": tot_n, "bytes_per_launch": tot_b / tot_n}
            if forwards:
                ent["forwards"] = forwards
                ent["bytes_per_forward"] = tot_b / forwards
                ent["launches_per_forward"] = tot_n / forwards
            return ent
    return None


if __name__ == "__main__":
    main()
