#!/bin/bash
# swap-chain depth x GPU_MAX_HW_QUEUES on the bench's timed region
cd ${GRAFT_REPO_ROOT:-.}
for q in ${QUEUES:-default 8}; do
for d in ${DEPTHS:-3 4 5}; do
for w in ${WORKLOADS:-trex1024}; do
  s=1000; [ $w = bunny4096 ] && s=60; [ $w = trex8192 ] && s=30; [ $w = synth10m ] && s=5
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --no-cpu-baseline --workload $w --steps $s --warmup 20 --pipeline-depth $d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('queues=$q depth=$d %-10s fps=%9.1f ms=%7.4f single=%7.4f' % ('$w', d['value'], d['ms_per_step'], d['ms_per_frame_single_stream']))"
done; done; done
