#!/bin/bash
# quick A/B on the GPU box: scripts/quick.sh [tiles...]
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-cpu-baseline --no-api-calls "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('dbg=%-2s %-10s tile=%-4s fps=%9.1f ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f frac=%.3f' % (os.environ.get('CRENDER_DEBUG','0'), d['config']['workload'], d['config']['tile'], d['value'], d['ms_per_step'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster'], d['roofline']['frac']))"; }
for t in ${TILES:-16 32 64}; do
run --workload trex1024 --tile $t --steps 200
run --workload bunny4096 --tile $t --steps 30 --warmup 3
run --workload trex8192 --tile $t --steps 20 --warmup 3
run --workload synth10m --tile $t --steps 5 --warmup 2
done
