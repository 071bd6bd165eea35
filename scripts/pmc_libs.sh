#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of k_raster for prebuilt libraries (scripts/ab/*.so, or "cur")
#   LIBS="base cur" COUNTERS="FETCH_SIZE" scripts/pmc_libs.sh synth10m [extra bench args]
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd); W=${1:-synth10m}; shift
for v in ${LIBS:-cur}; do
  if [ $v = cur ]; then unset CRENDER_LIB; else export CRENDER_LIB=$REPO/scripts/ab/$v.so; fi
  for c in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
    out=/tmp/pmcl_${v}_$c; rm -rf $out
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $c --output-format csv -d $out -- python3 $REPO/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-api-calls --no-pipeline "$@" > $out.log 2>&1)
    python - $out $v $c <<'PY'
import csv, glob, sys
d, g, c = sys.argv[1:4]
v = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "k_raster" in r["Kernel_Name"] and r["Counter_Name"] == c]
if v:
    kb = sum(v) / len(v)
    print(f"lib={g:10s} k_raster {c:10s} n={len(v):3d} avg={kb:12.0f} KB  = {kb * 1024 / 1e6:8.1f} MB raw" + (f", {2 * kb * 1024 / 1e6:8.1f} MB x2" if c == "FETCH_SIZE" else ""))
PY
  done
done
