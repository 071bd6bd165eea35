#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<workload>/ directory (rocprofv3 CSVs) into one text summary:
per-kernel call count / average duration from the kernel trace, and per-kernel averages of
each PMC counter.  FETCH_SIZE is shown raw and x2 (gfx950 counts 64 B per 128-B request for
wide reads, MI355X_MICROARCH.md "HBM")."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    for k in ("k_raster", "k_frame", "k_setup", "k_bin_wave", "k_count_wave", "k_scan", "k_fill", "k_project", "k_clear", "k_keys_init",
              "k_cover_atomic", "k_resolve_global", "k_guro"):
        if k in name:
            # (32-pixel plans have two raster kernels, csrc/raster.hip kPath*: <TS, CLEAR, 1> is the pixel owners')
            if k in ("k_raster", "k_frame") and re.search(k + r"<\d+, *(true|false), *1>", name):
                return k + "_owners"
            return k
    return name[:40]


def split_by_overlap(trace_csv, quiet=False):
    """Per kernel: average duration of the dispatches that overlap no other dispatch of the same
    kernel (launches one after another on one stream: the duration is the kernel's own) and of those
    that do (frames in flight together on the swap chain's streams)."""
    rows = []
    for row in csv.DictReader(open(trace_csv)):
        rows.append((short(row["Kernel_Name"]), int(row["Start_Timestamp"]), int(row["End_Timestamp"]),
                     row.get("Scratch_Size", row.get("Private_Segment_Size", "?"))))
    out = {}
    by = defaultdict(list)
    for k, a, b, sc in rows:
        by[k].append((a, b, sc))
    if not quiet:
        print("## kernel trace split by overlap with other dispatches of the same kernel")
    for k, lst in by.items():
        lst.sort()
        alone, over = [], []
        for i, (a, b, sc) in enumerate(lst):
            hit = (i > 0 and max(e for _, e, _ in lst[max(0, i - 8):i]) > a) or (i + 1 < len(lst) and lst[i + 1][0] < b)
            (over if hit else alone).append(b - a)
        out[k] = (alone, over, lst[0][2])
        if not quiet:
            fmt = lambda v: f"n={len(v):5d} avg_ns={sum(v) / len(v):12.1f}" if v else "n=    0"
            print(f"{k:18s} alone: {fmt(alone)}   overlapped: {fmt(over)}   scratch={lst[0][2]}")
    return out


def main(d):
    print(f"# {d}")
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("## kernel stats (rocprofv3 --kernel-trace --stats)")
        for row in csv.DictReader(open(f)):
            print(f"{short(row['Name']):18s} calls={row['Calls']:>6s} avg_ns={float(row['AverageNs']):>12.1f} "
                  f"total_ns={row['TotalDurationNs']:>12s} pct={row['Percentage']}")
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        split_by_overlap(f)
        regs = {}
        for row in csv.DictReader(open(f)):
            regs[short(row["Kernel_Name"])] = (row.get("VGPR_Count"), row.get("SGPR_Count"),
                                               row.get("LDS_Block_Size"), row.get("Grid_Size"),
                                               row.get("Workgroup_Size"))
        print("## launch shape (vgpr, sgpr, lds, grid, wg)")
        for k, v in regs.items():
            print(f"{k:18s} {v}")
    for pd in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(pd):
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(pd, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if acc:
            print(f"## {os.path.basename(pd)} (average per launch)")
        for k, cs in acc.items():
            for c, vals in cs.items():
                avg = sum(vals) / len(vals)
                extra = ""
                if c == "FETCH_SIZE":
                    extra = f"  = {avg * 1024 / 1e6:.2f} MB raw, {avg * 2048 / 1e6:.2f} MB x2-corrected"
                if c == "WRITE_SIZE":
                    extra = f"  = {avg * 1024 / 1e6:.2f} MB"
                print(f"{k:18s} {c:24s} n={len(vals):4d} avg={avg:16.1f}{extra}")


if __name__ == "__main__":
    for d in sys.argv[1:]:
        main(d)
