#!/bin/bash
# A/B of alternative builds: WORKLOADS="trex1024 cube256" scripts/ab_libs.sh variants/libA.so variants/libB.so ...
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('%-24s %-10s fps=%9.1f ms=%7.4f single_ms=%7.4f raster_ms=%7.4f' % (os.path.basename(os.environ.get('CRENDER_LIB','default')), d['config']['workload'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['raster']))"; }
for rep in 1 2; do
for lib in "$@"; do
export CRENDER_LIB=$PWD/$lib
for w in ${WORKLOADS:-trex1024 cube256}; do
  s=500; [ $w = bunny4096 ] && s=60; [ $w = trex8192 ] && s=30; [ $w = synth10m ] && s=5
  run --workload $w --steps $s --warmup 5
done
done
done
