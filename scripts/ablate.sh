#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('dbg=%-2s %-10s tile=%-4s fps=%9.1f ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f' % (os.environ.get('CRENDER_DEBUG','0'), d['config']['workload'], d['config']['tile'], d['value'], d['ms_per_step'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster']))"; }
for dbg in 0 4 1 2 3 5 6; do
export CRENDER_DEBUG=$dbg
run --workload trex1024 --tile 16 --steps 200
run --workload trex1024 --tile 32 --steps 200
run --workload bunny4096 --tile 32 --steps 30 --warmup 3
run --workload trex8192 --tile 32 --steps 20 --warmup 3
done
