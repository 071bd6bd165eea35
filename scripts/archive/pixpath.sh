#!/bin/bash
# A/B of the pixel-parallel short-batch threshold (CRENDER_DEBUG bits 16..23 = threshold + 1)
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('dbg=%-8s %-10s tile=%-4s fps=%9.1f ms=%7.4f single_ms=%7.4f raster_ms=%7.4f' % (os.environ.get('CRENDER_DEBUG','0'), d['config']['workload'], d['config']['tile'], d['value'], d['ms_per_step'], d.get('ms_per_frame_single_stream', d.get('latency', {}).get('ms_per_frame_single_stream', 0)), d['kernel_ms']['raster']))"; }
for v in 0 4 8 12 16 24 32 48 64; do
export CRENDER_DEBUG=$(( (v + 1) << 16 ))
echo "threshold $v"
run --workload trex1024 --steps 300
run --workload cube256 --steps 300
run --workload bunny4096 --tile 16 --steps 30 --warmup 3
done
