#!/usr/bin/env python3
"""Prints the last N kernel dispatches of a rocprofv3 kernel trace as a timeline (us since the first
one shown): start, end, duration, queue, kernel — to see which launches overlap."""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for k in ("k_raster_queue", "k_raster", "k_frame", "k_setup_wave", "k_setup", "k_count_wave", "k_scan", "k_fill_wave", "k_fill"):
            if k in name:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], k))
                break
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
for a, b, q, k in rows:
    print(f"{(a - t0) / 1e3:10.1f} {(b - t0) / 1e3:10.1f} {(b - a) / 1e3:9.1f}us q={q} {k}")
