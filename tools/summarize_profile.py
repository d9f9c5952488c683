#!/usr/bin/env python3
"""Turn a tools/profile.sh output directory (gpurun_out/prof_<tag>) into the small text files that
are committed under profiles/: per-kernel stats of the --kernel-trace --stats run, mean PMC counters of
the dominant kernel per pass, and the HBM traffic per launch with the gfx950 FETCH_SIZE correction
(MI355X_MICROARCH.md section HBM: FETCH_SIZE under-reports a wide coalesced read stream by exactly 2x;
WRITE_SIZE uncalibrated, taken as is; both are in KiB)."""
import collections
import csv
import glob
import json
import os
import sys


TRAFFIC_NOTE = ("HBM-side bytes per launch of each workload's dominant kernel: rocprofv3 --pmc FETCH_SIZE x 1024 x 2 (gfx950 "
                "correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1024, separate passes (tools/profile.sh, tools/pmc_shape.sh); "
                "bench.py quotes an entry only when source_stamp is the build's")


def merge_traffic(path, stamp, workload, entry):
    """profiles/pmc_traffic.json = {source_stamp, workloads: {name: entry}}; entries taken on other kernel sources go."""
    try:
        cur = json.load(open(path))
    except (OSError, ValueError):
        cur = {}
    if cur.get("source_stamp") != stamp:
        cur = {"source_stamp": stamp, "note": TRAFFIC_NOTE, "workloads": {}}
    cur.setdefault("workloads", {})[workload] = entry
    json.dump(cur, open(path, "w"), indent=1)


def merge_pmc_shape(src, workload, dst):
    """A tools/pmc_shape.sh output directory (gpurun_out/pmc_<tag>) -> profiles/<tag>_rocprof_summary.txt + its entry
    (the launch's dominant kernel, and under "kernels" every other kernel the run was asked to report - kernel A, the
    training kernels - with its own counter traffic and rocprofv3 average)."""
    import re
    tag = os.path.basename(os.path.normpath(src))[len("pmc_"):]
    txt = open(os.path.join(src, "pmc_means.txt")).read()
    stamp = re.search(r"stamp \(bench.kernel_source_stamp\): (\w+)", txt).group(1)
    sections = []   # (name, average us, {counter: mean})
    for line in txt.split("\n"):
        m = re.match(r"# (?:dominant )?kernel: (.*?)\s+\(kernel-trace pass: (\d+) calls, average ([\d.]+) us\)", line)
        if m:
            sections.append((m.group(1).strip(), float(m.group(3)), {}))
            continue
        m = re.match(r"^\s*([\d.]+)\s+n=\s*\d+\s+(\w+)$", line)
        if m and sections:
            sections[-1][2][m.group(2)] = float(m.group(1))

    def entry(name, avg_us, vals):
        f, w = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
        e = {"kernel": name, "rocprof_avg_us": avg_us, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
             "traffic_bytes_per_launch": f + w, "hbm_gbs_over_rocprof_avg": (f + w) / avg_us / 1e3}
        if "TCC_HIT_sum" in vals:
            e["l2_hit_rate"] = vals["TCC_HIT_sum"] / max(1.0, vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            e["mfma_busy_cycles_per_launch"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"]
        if "GRBM_GUI_ACTIVE" in vals:
            e["grbm_gui_active_sum_over_8_xcds"] = vals["GRBM_GUI_ACTIVE"]
        return e
    dom = entry(*sections[0])
    dom["tag"] = tag
    dom["kernels"] = {}
    for name, avg_us, vals in sections[1:]:
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            short = re.sub(r"^(void )?gnf::", "", name).split("(")[0]
            dom["kernels"][short] = entry(name, avg_us, vals)
    merge_traffic(os.path.join(dst, "pmc_traffic.json"), stamp, workload, dom)
    lines = [f"# tools/pmc_shape.sh {tag} {workload}: rocprofv3 --kernel-trace --stats of bench.py --workload {workload} --steps 20 "
             "--warmup 5 --no-cpu-baseline --no-secondary --latency-steps 0 --prewarm-ms 0, then one --pmc pass per counter group",
             open(os.path.join(src, "kernel_stats.txt")).read(), txt,
             f"# bench.py --workload {workload} --steps 50 --warmup 10 on the same box, un-profiled:",
             open(os.path.join(src, "bench.json")).read().strip().split("\n")[-1]]
    open(os.path.join(dst, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")


def main(src, tag, dst):
    os.makedirs(dst, exist_ok=True)
    lines = []
    ks = os.path.join(src, "trace", "kt_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks)))
    lines.append(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --latency-steps 0   ({tag})")
    lines.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for r in rows:
        lines.append(f"{int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e3:12.1f} {float(r['AverageNs'])/1e3:10.3f} "
                     f"{float(r['MinNs'])/1e3:9.3f} {float(r['MaxNs'])/1e3:9.3f} {float(r['Percentage']):6.2f}  {r['Name'][:110]}")
    dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))["Name"]
    lines.append("")
    lines.append(f"# PMC passes (separate runs, --pmc only with --kernel-trace): mean per dispatch of the dominant kernel")
    lines.append(f"# dominant kernel: {dom}")
    pmc = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == dom:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc[k] = sum(v) / len(v)
            lines.append(f"{pmc[k]:18.1f}  n={len(v):4d}  {k}")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    stamp_file = os.path.join(src, "source_stamp.txt")   # written on the GPU box by tools/profile.sh
    stamp = open(stamp_file).read().strip() if os.path.exists(stamp_file) else bench.kernel_source_stamp()
    out = {"tag": tag, "kernel": dom, "source_stamp": stamp}
    lines.append(f"# kernel sources stamp (bench.kernel_source_stamp): {out['source_stamp']}")
    if "FETCH_SIZE" in pmc:
        out["fetch_bytes_per_launch"] = pmc["FETCH_SIZE"] * 1024 * 2      # gfx950 correction x2
        out["write_bytes_per_launch"] = pmc.get("WRITE_SIZE", 0.0) * 1024
        out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
        lines.append("")
        lines.append(f"# HBM-side traffic per launch = FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024 = "
                     f"{out['traffic_bytes_per_launch']/1e6:.2f} MB")
    if "TCC_HIT_sum" in pmc:
        out["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
        lines.append(f"# L2 hit rate = {out['l2_hit_rate']:.4f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pmc and "SQ_WAVE_CYCLES" in pmc:
        lines.append(f"# SQ_WAIT_ANY/SQ_WAVE_CYCLES = {pmc['SQ_WAIT_ANY']/pmc['SQ_WAVE_CYCLES']:.3f}, "
                     f"SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES = {pmc['SQ_WAIT_INST_ANY']/pmc['SQ_WAVE_CYCLES']:.3f}, "
                     f"SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES = {pmc['SQ_ACTIVE_INST_ANY']/pmc['SQ_WAVE_CYCLES']:.3f}")
    open(os.path.join(dst, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    merge_traffic(os.path.join(dst, "pmc_traffic.json"), out.pop("source_stamp"), "config2", out)
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj):
        txt = [l for l in open(bj) if l.startswith("{")]
        if txt:
            open(os.path.join(dst, f"{tag}_bench.json"), "w").write(txt[-1])
    print("\n".join(lines))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--pmc-shape":
    # summarize_profile.py --pmc-shape gpurun_out/pmc_<tag> <workload> [profiles]
    merge_pmc_shape(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "profiles")
elif __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "profiles")
