#!/bin/bash
# GPU box: dev-build A/B lines.  usage: scripts/r3_ab.sh "<dbg values>" "<workloads>" [extra bench args]
cd ${GRAFT_REPO_ROOT:-.}
export CRENDER_LIB=$(scripts/dev_build.sh | tail -1)
DBGS=$1; WLS=$2; shift 2
for rep in 1 2; do for g in $DBGS; do for w in $WLS; do
  s=300; [ $w = bunny4096 ] && s=40; [ $w = trex8192 ] && s=20; [ $w = synth10m ] && s=8
  CRENDER_DEBUG=$g python bench.py --no-cpu-baseline --no-api-calls --workload $w --steps $s --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('dbg=%-7s %-10s fps=%9.1f ms=%7.4f single=%7.4f bin=%7.4f raster=%7.4f %s' % ('$g', d['config']['workload'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster'], {k[:12]:round(v,4) for k,v in r['avg_launch_ms_views'].items()}))"
done; done; done
