#!/bin/bash
# GPU box: is the driver's 20-step burst slow because 0.2 ms after a synchronisation is too short for the clocks (round 5's
# hypothesis, profiles/r05/k20_probe.txt)?  The same 20 timed steps behind warm-ups of 5 .. 5000 frames (each warm-up ends in
# the same synchronisation + barrier as ever), and for comparison 200 timed steps.
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/${OUTDIR:-r6w}; mkdir -p $OUT
for rep in 1 2 3; do
  for w in 5 50 500 5000; do
    python bench.py --no-cpu-baseline --no-api-calls --steps 20 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('K=%3d W=%4d  frames/s=%9.1f  us/frame=%6.2f' % (d['steps'], d['warmup'], d['value'], d['ms_per_step']*1e3))" | tee -a $OUT/k20_warm.txt
  done
  python bench.py --no-cpu-baseline --no-api-calls --steps 200 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('K=%3d W=%4d  frames/s=%9.1f  us/frame=%6.2f' % (d['steps'], d['warmup'], d['value'], d['ms_per_step']*1e3))" | tee -a $OUT/k20_warm.txt
done
