#!/bin/bash
# swap chain with and without look-ahead (one launch per frame), long run and the driver's short run
cd ${GRAFT_REPO_ROOT:-.}
for w in ${WL:-trex1024 cube256 bunny4096 trex8192}; do
for la in off on; do
for k in ${KS:-200 20}; do
  s=$k; [ $w = bunny4096 ] && [ $k = 200 ] && s=50; [ $w = trex8192 ] && [ $k = 200 ] && s=30
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --workload $w --steps $s --warmup 5 --lookahead $la 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-10s lookahead=%-3s K=%-4d fps=%9.1f ms=%7.4f single=%7.4f' % ('$w', '$la', $s, d['value'], d['ms_per_step'], d['ms_per_frame_single_stream']))"
  done
done; done; done
