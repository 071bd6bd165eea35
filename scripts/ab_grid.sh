#!/bin/bash
# A/B of k_raster's grid size (CRENDER_RASTER_GRID: 0 = one workgroup per tile, N = N workgroups
# pulling tiles from the queue, unset = as many as the device holds at once)
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('grid=%-8s %-10s fps=%9.1f ms=%7.4f single_ms=%7.4f raster_ms=%7.4f' % (os.environ.get('CRENDER_RASTER_GRID','auto'), d['config']['workload'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['raster']))"; }
for g in ${GRIDS:-0 auto}; do
if [ $g = auto ]; then unset CRENDER_RASTER_GRID; else export CRENDER_RASTER_GRID=$g; fi
for w in ${WORKLOADS:-trex1024}; do
  s=300; [ $w = bunny4096 ] && s=40; [ $w = trex8192 ] && s=20; [ $w = synth10m ] && s=5
  run --workload $w --steps $s --warmup 3
done
done
