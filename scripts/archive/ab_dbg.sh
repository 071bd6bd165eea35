#!/bin/bash
# A/B of CRENDER_DEBUG values: DBGS="0 512" WORKLOADS="trex1024" scripts/ab_dbg.sh
cd ${GRAFT_REPO_ROOT:-.}
export CRENDER_LIB=$(scripts/dev_build.sh | tail -1)   # knobs exist in the development build only
run() { python bench.py --no-cpu-baseline --no-api-calls "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('dbg=%-8s %-10s fps=%9.1f ms=%7.4f single_ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f' % (os.environ.get('CRENDER_DEBUG','0'), d['config']['workload'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster']))"; }
for rep in 1 2; do
for g in ${DBGS:-0}; do
export CRENDER_DEBUG=$g
for w in ${WORKLOADS:-trex1024}; do
  s=300; [ $w = bunny4096 ] && s=40; [ $w = trex8192 ] && s=20; [ $w = synth10m ] && s=5
  run --workload $w --steps $s --warmup 3
done
done
done
