#!/bin/bash
# GPU box: what does the FETCH_SIZE counter report for dword GATHERS of 36-byte records (the raster
# kernel's access pattern on the 10 M-triangle frame)?  scripts/ubench/gather_stride gathers 12 M
# records out of a 10 M-record table at record strides of 36 / 48 / 64 / 128 bytes; the bytes that
# must come from HBM are known (below), so the counter's unit for this pattern follows.
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd)
[ -x scripts/ubench/gather_stride ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/gather_stride scripts/ubench/gather_stride.hip
rm -rf /tmp/fetchcal
(cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fetchcal -- $REPO/scripts/ubench/gather_stride > /tmp/fetchcal.log 2>&1)
cat /tmp/fetchcal.log | grep stride
python - <<'PY'
import csv, glob, math
rows = []
for f in glob.glob("/tmp/fetchcal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gather" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
T, N = 10_000_000, 12_000_000
for k, stride in enumerate((9, 12, 16, 32)):
    vals = [v for _, v in rows[k * 10:(k + 1) * 10]]
    if not vals:
        continue
    kb = sum(vals) / len(vals)
    rec = stride * 4
    # distinct 128-byte lines (and 64-byte halves) touched by 12 M uniform gathers of 36 bytes
    def touched(gran):
        lines = T * rec / gran
        per = 36 / gran + (1 if rec % gran else 0) * 0 + 0     # expected lines per gather ~ 1 + (36 - 1) / gran for random alignment
        per = 1 + 35 / gran if rec % gran else (36 + gran - 1) // gran
        return lines * (1 - math.exp(-N * per / lines)) * gran
    idx_out = N * 8
    print(f"stride {rec:3d} B: FETCH_SIZE {kb:12.0f} per launch | expected HBM bytes: {(touched(128) + idx_out) / 1e6:7.1f} MB at 128-B lines, "
          f"{(touched(64) + idx_out) / 1e6:7.1f} MB at 64-B | counter x 1 KB = {kb * 1024 / 1e6:7.1f} MB -> factor vs 128-B lines {(touched(128) + idx_out) / (kb * 1024):.2f}, vs 64-B {(touched(64) + idx_out) / (kb * 1024):.2f}")
PY
# second part: how large are the L2's read requests to the fabric for this pattern?  (request counts by size;
# bounded: these derived counters need several replays)
rm -rf /tmp/fetchcal2
(cd /tmp && TMPDIR=/tmp timeout -k 10 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d /tmp/fetchcal2 -- $REPO/scripts/ubench/gather_stride > /tmp/fetchcal2.log 2>&1; echo "rc=$?")
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/fetchcal2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gather" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for name, rows in sorted(acc.items()):
    rows.sort()
    for k, stride in enumerate((36, 48, 64, 128)):
        v = [x for _, x in rows[k * 10:(k + 1) * 10]]
        if v:
            print(f"{name:28s} stride {stride:3d} B: {sum(v) / len(v):14.0f} per launch")
PY
