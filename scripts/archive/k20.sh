cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s fps=%9.1f ms/step=%7.4f single=%7.4f' % (sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream']))" "$*"; }
for rep in 1 2; do
run --steps 20 --warmup 5
run --steps 20 --warmup 500
run --steps 20 --warmup 5000
run --steps 200 --warmup 5
run --steps 20 --warmup 5 --pipeline-depth 2
run --steps 20 --warmup 5000 --pipeline-depth 2
run --steps 20 --warmup 5000 --no-pipeline
done
