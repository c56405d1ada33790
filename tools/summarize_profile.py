#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into small text summaries for profiles/.

  python tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01
"""
import csv
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:90]


def kernel_stats(path, out):
    with open(path) as f:
        rows = list(csv.DictReader(f))
    out.write(f"{'kernel':92s} {'calls':>6s} {'avg_ms':>10s} {'total_ms':>10s} {'pct':>7s}\n")
    for r in rows[:12]:
        out.write(f"{short(r['Name']):92s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e6:10.3f} "
                  f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):7.2f}\n")


def counters(path, out, only=("render_", "residual_", "reduce_item", "sample_", "field_eval", "ray_directions", "reduce_partials", "project_")):
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = short(r["Kernel_Name"])
            if not any(k in name for k in only):
                continue
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[name] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])
    for name, cs in acc.items():
        grid, wg, lds, vgpr, agpr, sgpr = meta[name]
        out.write(f"{name}\n  grid={grid} workgroup={wg} lds_bytes={lds} vgpr={vgpr} agpr={agpr} sgpr={sgpr}\n")
        for counter, values in sorted(cs.items()):
            out.write(f"  {counter:28s} dispatches={len(values):3d} mean={sum(values) / len(values):.6g} last={values[-1]:.6g}\n")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "kernel_stats.txt"), "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats (top kernels by total time)\n")
        kernel_stats(os.path.join(src, "trace", "bench_kernel_stats.csv"), out)
        log = os.path.join(src, "trace_stdout.log")
        if os.path.exists(log):
            for line in open(log):
                if line.startswith("{\"metric\""):
                    out.write("\n# bench.py line of the same run\n" + line)
    with open(os.path.join(dst, "pmc_counters.txt"), "w") as out:
        out.write("# rocprofv3 --pmc passes (one counter group per pass); values are per dispatch, summed over the chip.\n"
                  "# FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived metrics); MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half\n"
                  "# the bytes of a wide coalesced read stream (uncalibrated for narrow accesses) -- treat as a lower bound.\n")
        for sub in ("pmc_sq", "pmc_lds", "pmc_mix", "pmc_fetch", "pmc_write"):
            path = os.path.join(src, sub, "bench_counter_collection.csv")
            if os.path.exists(path):
                out.write(f"\n## {sub}\n")
                counters(path, out)
    traffic(src, dst)
    counters_json(src, dst)


def counters_json(src, dst):
    """profiles/<tag>/counters.json: the per-launch counters bench.py quotes (`roofline.traffic`, `roofline_valu.executed`), keyed by
    the workload they were measured on (config.workload_key of the bench line of the same command): last dispatch of each kernel."""
    import json
    key = line = None
    log = os.path.join(src, "trace_stdout.log")
    if os.path.exists(log):
        for text in open(log):
            if text.startswith("{\"metric\""):
                line = json.loads(text)
                key = line.get("config", {}).get("workload_key")
    kernels = defaultdict(dict)
    for sub in ("pmc_sq", "pmc_lds", "pmc_mix", "pmc_fetch", "pmc_write"):
        path = os.path.join(src, sub, "bench_counter_collection.csv")
        if not os.path.exists(path):
            continue
        # steps the profiled command ran (warm-up + timed: every one of them launches the step's kernels once)
        steps = 2
        pass_log = os.path.join(src, sub + "_stdout.log")
        if os.path.exists(pass_log):
            for text in open(pass_log):
                if text.startswith("{\"metric\""):
                    record = json.loads(text)
                    steps = int(record.get("steps", 1)) + int(record.get("warmup", 1))
        sums = defaultdict(lambda: defaultdict(float))
        with open(path) as f:
            for r in csv.DictReader(f):
                if "vsrd::" not in r["Kernel_Name"] and "vsrd_split::" not in r["Kernel_Name"]:
                    continue
                name, counter, value = short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"])
                if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                    counter, value = counter + "_bytes", value * 1024.0
                kernels[name][counter] = value                   # rows are in dispatch order: the last one (a timed step) stays
                sums[name][counter] += value
                kernels[name]["vgpr"], kernels[name]["agpr"] = int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"])
        for name, per in sums.items():                           # a step may launch a kernel many times (residual step: once per chunk of rays)
            for counter, total in per.items():
                kernels[name][counter + "_per_step"] = total / steps
    stats = os.path.join(src, "trace", "bench_kernel_stats.csv")
    if os.path.exists(stats):
        with open(stats) as f:
            for r in csv.DictReader(f):
                name = short(r["Name"])
                if name in kernels:
                    kernels[name]["rocprof_avg_ms"] = float(r["AverageNs"]) / 1e6
    with open(os.path.join(dst, "counters.json"), "w") as f:
        json.dump({"workload_key": key, "note": "rocprofv3 --pmc, one counter group per pass (tools/profile_bench.sh), last dispatch of each "
                   "kernel; *_bytes = KiB x 1024, uncorrected (see traffic.json); counts are wave instructions summed over the chip",
                   "bench_line": line, "kernels": kernels}, f, indent=1)


def traffic(src, dst):
    """profiles/<tag>/traffic.json: per-kernel HBM traffic of one launch from the FETCH_SIZE / WRITE_SIZE passes (bytes)."""
    import json
    out = defaultdict(dict)
    for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        path = os.path.join(src, sub, "bench_counter_collection.csv")
        if not os.path.exists(path):
            continue
        per = defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == key and ("vsrd::" in r["Kernel_Name"] or "vsrd_split::" in r["Kernel_Name"]):
                    per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for name, values in per.items():
            out[name][key + "_bytes"] = values[-1] * 1024.0          # rocprofv3 reports KiB; last dispatch = a timed step
    with open(os.path.join(dst, "traffic.json"), "w") as f:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), last dispatch of each kernel, KiB*1024, uncorrected. "
                           "Calibration: WRITE_SIZE of ray_directions_kernel = algorithmic 12 B x pixels exactly; the x2 gfx950 FETCH correction of "
                           "MI355X_MICROARCH.md is for 16 B/lane streams and is NOT applied to these 4 B/lane reads.",
                   "kernels": out}, f, indent=1)


if __name__ == "__main__":
    main()
