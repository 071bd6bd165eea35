#!/bin/bash
# GPU box: A/B of the development build's knobs (scripts/ab/dev.so, built in the container with -DCRENDER_DEV_KNOBS)
#   CONFIGS="name:CRENDER_DEBUG:CRENDER_ORDER_CELL_SHIFT ..." WORKLOAD=synth10m scripts/ab_dev.sh
cd ${GRAFT_REPO_ROOT:-.}
export CRENDER_LIB=$(pwd)/scripts/ab/${DEVLIB:-dev}.so
line() { python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('%-14s %-10s fps=%9.1f ms=%7.4f single_ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f b2b=%7.4f' % (os.environ['ABNAME'], d['config']['workload'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster'], d['kernel_ms']['raster_back_to_back']))"; }
for c in ${CONFIGS:-base:0:5}; do
  IFS=: read name dbg cell <<< "$c"
  export ABNAME=$name CRENDER_DEBUG=$dbg CRENDER_ORDER_CELL_SHIFT=$cell
  for rep in 1 2; do python bench.py --no-cpu-baseline --no-api-calls --workload ${WORKLOAD:-synth10m} --steps ${STEPS:-20} --warmup ${WARMUP:-5} ${EXTRA:-} 2>/dev/null | line; done
  if [ -n "${PMC:-}" ]; then
    for cn in FETCH_SIZE; do
      out=/tmp/pmcd_${name}_$cn; rm -rf $out
      (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $cn --output-format csv -d $out -- python3 $OLDPWD/bench.py --workload ${WORKLOAD:-synth10m} --steps 10 --warmup 2 --no-cpu-baseline --no-api-calls --no-pipeline > $out.log 2>&1)
      python - $out $name $cn <<'PY'
import csv, glob, sys
d, g, c = sys.argv[1:4]
v = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "k_raster" in r["Kernel_Name"] and r["Counter_Name"] == c]
if v:
    kb = sum(v) / len(v)
    print(f"    {g:14s} k_raster {c:10s} n={len(v):3d} = {kb * 1024 / 1e6:8.1f} MB raw" + (f", {2 * kb * 1024 / 1e6:8.1f} MB x2" if c == "FETCH_SIZE" else ""))
PY
    done
  fi
done
