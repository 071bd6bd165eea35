import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "raster" if "k_raster" in n else "setup" if "k_setup" in n else None
        if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id")))
rows.sort()
t0 = rows[0][0]
for s, e, k, q in rows[-16:]:
    print(f"{k:7s} q={q} start {(s-t0)/1000:9.2f} us  end {(e-t0)/1000:9.2f} us  dur {(e-s)/1000:6.2f}")
