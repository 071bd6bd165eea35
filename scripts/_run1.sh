cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w1
W=${W:-synth10m}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w1/prof_$W -- python bench.py --no-cpu-baseline --workload $W --steps ${K:-5} --warmup 2 ${EXTRA:-} > gpurun_out/w1/prof_bench_$W.json 2>/dev/null
python - <<PY
import csv,glob,json
d=json.loads(open('gpurun_out/w1/prof_bench_$W.json').read().strip().splitlines()[-1])
print('$W (under rocprof) fps', d['value'], 'ms', d['ms_per_step'], 'single', d['ms_per_frame_single_stream'])
for p in glob.glob('gpurun_out/w1/prof_$W/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if float(r['Percentage'])>0.5: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
