#!/bin/bash
# GPU box: memory-path counters of one workload for a development build and a list of CRENDER_DEBUG values.
#   DBGS="0 2" scripts/pmc_mem.sh synth10m
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd); W=${1:-synth10m}
export CRENDER_LIB=$(scripts/dev_build.sh | tail -1)
for g in ${DBGS:-0}; do
  export CRENDER_DEBUG=$g
  for set in "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    out=/tmp/pmcm_$g; rm -rf $out
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $set --output-format csv -d $out -- python3 $REPO/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-api-calls --no-pipeline > $out.log 2>&1)
    python - $out $g <<'PY'
import csv, glob, sys, collections
d, g = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_raster" if "k_raster" in k else k[:24]
        a = acc[(k, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    if k == "k_raster": print(f"dbg={g:10s} {k:10s} {c:36s} n={n:4d} avg={v / n:16.1f}")
PY
  done
done
