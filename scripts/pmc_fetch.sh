#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of k_raster for a development build under a list of CRENDER_DEBUG values.
#   DBGS="0 2" scripts/pmc_fetch.sh synth10m
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd); W=${1:-synth10m}
export CRENDER_LIB=$(scripts/dev_build.sh --out /tmp/pmc_fetch.so | tail -1)
for g in ${DBGS:-0}; do
  export CRENDER_DEBUG=$g
  for c in FETCH_SIZE WRITE_SIZE; do
    out=/tmp/pmcf_${g}_$c; rm -rf $out
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $c --output-format csv -d $out -- python3 $REPO/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-api-calls --no-pipeline > $out.log 2>&1)
    python - $out $g $c <<'PY'
import csv, glob, sys
d, g, c = sys.argv[1:4]
v = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "k_raster" in r["Kernel_Name"] and r["Counter_Name"] == c]
if v:
    kb = sum(v) / len(v)
    print(f"dbg={g:10s} k_raster {c:10s} n={len(v):3d} avg={kb:12.0f} KB  = {kb * 1024 / 1e6:8.1f} MB raw" + (f", {2 * kb * 1024 / 1e6:8.1f} MB x2" if c == "FETCH_SIZE" else ""))
PY
  done
done
