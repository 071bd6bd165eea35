#!/bin/bash
# Runs on the GPU box: the full evidence set of a round -> gpurun_out/report/
# (copy what is to be judged into profiles/rNN/ afterwards: gpurun_out/ is scratch)
#   PART=a  smoke, pytest -m gpu, the bench lines
#   PART=b  rocprofv3: kernel trace + PMC passes, single-stream (k_raster) and pipelined (k_frame)
#   PART=c  the bench lines alone
#   PART=all (default) everything
cd ${GRAFT_REPO_ROOT:-.}
PART=${PART:-all}
OUT=gpurun_out/report; mkdir -p $OUT
# (the profiles first: bench.py reads profiles/kernel_avg.json and profiles/traffic.json of THIS run)
if [ $PART = b ] || [ $PART = all ]; then
for w in trex1024 bunny4096 trex8192 synth10m; do
  rm -rf gpurun_out/prof_$w
  s=20; [ $w = trex1024 ] && s=100; [ $w = synth10m ] && s=40     # (enough launches that the cold first frames do not set the average)
  scripts/profile_gpu.sh $w $s > $OUT/profile_$w.log 2>&1
  python scripts/summarize_prof.py gpurun_out/prof_$w | grep -v "at::native\|rocclr\|^void" > $OUT/rocprof_$w.txt
  cp gpurun_out/prof_$w/trace/*/*kernel_stats.csv $OUT/${w}_kernel_stats.csv 2>/dev/null
done
# the pipelined run — the driver's own command shape — for the workloads whose frames are k_frame launches
for w in trex1024 bunny4096 trex8192; do
  rm -rf gpurun_out/prof_${w}_pipelined
  s=20; [ $w = trex1024 ] && s=200
  MODE=pipelined scripts/profile_gpu.sh $w $s > $OUT/profile_${w}_pipelined.log 2>&1
  python scripts/summarize_prof.py gpurun_out/prof_${w}_pipelined | grep -v "at::native\|rocclr\|^void" > $OUT/rocprof_${w}_pipelined.txt
  cp gpurun_out/prof_${w}_pipelined/trace/*/*kernel_stats.csv $OUT/${w}_pipelined_kernel_stats.csv 2>/dev/null
done
# the same box's launches on the OTHER raster kernel, for the workloads whose plans choose the pixel owners (kernel trace only)
for w in bunny4096 trex8192; do
  rm -rf gpurun_out/prof_${w}_general
  SUFFIX=_general TRACE_ONLY=1 scripts/profile_gpu.sh $w 20 --raster-path 0 > $OUT/profile_${w}_general.log 2>&1
  python scripts/summarize_prof.py gpurun_out/prof_${w}_general | grep -E "^k_raster|^k_frame|^## kernel stats" > $OUT/rocprof_${w}_general_kernel.txt
done
python scripts/make_kernel_avg_json.py gpurun_out > $OUT/kernel_avg.log 2>&1
cp profiles/kernel_avg.json $OUT/kernel_avg.json
python scripts/make_traffic_json.py $OUT > $OUT/traffic.log 2>&1
cp profiles/traffic.json $OUT/traffic.json
# who overlaps whom on 10 M triangles (three frames of the swap chain march in step)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/tl_synth -- python3 $OLDPWD/bench.py --workload synth10m --steps 6 --warmup 2 --no-cpu-baseline --no-api-calls > $OLDPWD/$OUT/tl_synth.log 2>&1)
python scripts/timeline.py gpurun_out/tl_synth 400 2>/dev/null | sed -n 1,60p > $OUT/timeline_synth10m.txt
fi
if [ $PART = a ] || [ $PART = all ]; then
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
fi
if [ $PART = a ] || [ $PART = c ] || [ $PART = all ]; then      # (c: the bench lines alone)
python bench.py > $OUT/bench_trex1024.json 2> $OUT/bench_trex1024.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > $OUT/bench_trex1024_k20.json 2>/dev/null       # the driver's command line
python bench.py --workload bunny4096 --steps 50 --warmup 5 > $OUT/bench_bunny4096.json 2>/dev/null
python bench.py --workload trex8192 --steps 30 --warmup 3 > $OUT/bench_trex8192.json 2>/dev/null
python bench.py --workload synth10m --steps 40 --warmup 10 > $OUT/bench_synth10m.json 2>/dev/null     # (the first ~20 frames after the 1.5 GB upload run up to 15 % slower: K = 10, W = 2 gave 1 062-1 104 frames/s)
python bench.py --workload cube256 --steps 200 > $OUT/bench_cube256.json 2>/dev/null
fi
ls $OUT
