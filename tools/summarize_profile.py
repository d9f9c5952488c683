#!/usr/bin/env python3
"""Turn a tools/pmc_shape.sh output directory (gpurun_out/pmc_<tag>) into the small text file that is committed
under profiles/ (<tag>_rocprof_summary.txt: per-kernel stats of the --kernel-trace --stats run, mean PMC counters per
dispatch of the dominant kernel and of every other kernel the run reported) and into the workload's entry of
profiles/pmc_traffic.json: HBM traffic per launch with the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md section
HBM: FETCH_SIZE under-reports a wide coalesced read stream by exactly 2x; WRITE_SIZE uncalibrated, taken as is; both
are in KiB).
    python tools/summarize_profile.py --pmc-shape gpurun_out/pmc_<tag> <workload> [profiles]"""
import collections
import csv
import glob
import json
import os
import sys


TRAFFIC_NOTE = ("HBM-side bytes per launch of each workload's dominant kernel: rocprofv3 --pmc FETCH_SIZE x 1024 x 2 (gfx950 "
                "correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1024, separate passes (tools/pmc_shape.sh); every entry names the "
                "kernel sources it was taken on (sources) and their sha256 (source_stamp = bench.kernel_source_stamp(workload)); "
                "bench.py quotes an entry only while that stamp is the build's")


def merge_traffic(path, stamp, sources, workload, entry):
    """profiles/pmc_traffic.json = {note, workloads: {name: entry}}; an entry carries the stamp and the list of the kernel
    sources of ITS workload (bench.WORKLOAD_SOURCES), so evidence for one workload survives edits to another's kernels."""
    try:
        cur = json.load(open(path))
    except (OSError, ValueError):
        cur = {}
    cur.pop("source_stamp", None)
    cur["note"] = TRAFFIC_NOTE
    entry["source_stamp"], entry["sources"] = stamp, sources
    cur.setdefault("workloads", {})[workload] = entry
    json.dump(cur, open(path, "w"), indent=1)


def merge_pmc_shape(src, workload, dst):
    """A tools/pmc_shape.sh output directory (gpurun_out/pmc_<tag>) -> profiles/<tag>_rocprof_summary.txt + its entry
    (the launch's dominant kernel, and under "kernels" every other kernel the run was asked to report - kernel A, the
    training kernels - with its own counter traffic and rocprofv3 average)."""
    import re
    tag = os.path.basename(os.path.normpath(src))[len("pmc_"):]
    txt = open(os.path.join(src, "pmc_means.txt")).read()
    m = re.search(r"stamp \(bench.kernel_source_stamp\): (\w+)\s+sources: (\S+)", txt)
    stamp, sources = m.group(1), m.group(2).split(",")
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
    merge_traffic(os.path.join(dst, "pmc_traffic.json"), stamp, sources, workload, dom)
    lines = [f"# tools/pmc_shape.sh {tag} {workload}: rocprofv3 --kernel-trace --stats of bench.py --workload {workload} --steps 20 "
             "--warmup 5 --no-cpu-baseline --no-secondary --latency-steps 0 --prewarm-ms 0, then one --pmc pass per counter group",
             open(os.path.join(src, "kernel_stats.txt")).read(), txt,
             f"# bench.py --workload {workload} --steps 50 --warmup 10 on the same box, un-profiled:",
             open(os.path.join(src, "bench.json")).read().strip().split("\n")[-1]]
    open(os.path.join(dst, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    # summarize_profile.py --pmc-shape gpurun_out/pmc_<tag> <workload> [profiles]
    if len(sys.argv) < 4 or sys.argv[1] != "--pmc-shape":
        raise SystemExit(__doc__)
    merge_pmc_shape(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "profiles")
