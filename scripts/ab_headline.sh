#!/bin/bash
# GPU box: the headline's two commands (K = 200 and the driver's K = 20, W = 5) for the swap chain on
# 16- and on 32-pixel tiles, product library, same box; also cube 256^2.
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s chain tile=%-4s K=%-4d fps=%9.1f ms=%7.4f %s frac=%.3f views=%s' % (d['config']['workload'], d['config']['pipeline_tile'], d['steps'], d['value'], d['ms_per_step'], r['kernel'], r['frac'], {k[:12]: round(v,4) for k,v in r['avg_launch_ms_views'].items() if v}))"; }
for rep in 1 2; do
for pt in 16 32; do
  python bench.py --no-cpu-baseline --no-api-calls --pipeline-tile $pt --steps 200 --warmup 20 2>/dev/null | line
  python bench.py --no-cpu-baseline --no-api-calls --pipeline-tile $pt --steps 20 --warmup 5 2>/dev/null | line
done
done
for pt in 16 32; do python bench.py --no-cpu-baseline --no-api-calls --workload cube256 --pipeline-tile $pt --steps 200 --warmup 20 2>/dev/null | line; done
