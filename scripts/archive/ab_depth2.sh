#!/bin/bash
# swap-chain depth on the bench's timed region, long run and the driver's short run
cd ${GRAFT_REPO_ROOT:-.}
export GPU_MAX_HW_QUEUES=${Q:-8}
for rep in 1 2; do
for d in ${DEPTHS:-3 4 5 6 7}; do
for k in 200 20; do
  python bench.py --no-cpu-baseline --workload ${W:-trex1024} --steps $k --warmup 5 --pipeline-depth $d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('depth=$d K=%-4d fps=%9.1f ms=%7.4f single=%7.4f' % ($k, d['value'], d['ms_per_step'], d['ms_per_frame_single_stream']))"
done; done; done
