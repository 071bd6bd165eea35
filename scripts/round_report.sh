#!/bin/bash
# Runs on the GPU box: the full evidence set of a round -> gpurun_out/report/
# (copy what is to be judged into profiles/rNN/ afterwards: gpurun_out/ is scratch)
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/report; rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_trex1024.json 2> $OUT/bench_trex1024.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > $OUT/bench_trex1024_k20.json 2>/dev/null       # the driver's command line
python bench.py --workload bunny4096 --steps 50 --warmup 5 > $OUT/bench_bunny4096.json 2>/dev/null
python bench.py --workload trex8192 --steps 30 --warmup 3 > $OUT/bench_trex8192.json 2>/dev/null
python bench.py --workload synth10m --steps 10 --warmup 2 > $OUT/bench_synth10m.json 2>/dev/null
python bench.py --workload cube256 --steps 200 > $OUT/bench_cube256.json 2>/dev/null
WL="trex1024 cube256" scripts/ab_lookahead.sh > $OUT/lookahead.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/clear_shapes.hip -o /tmp/clear_shapes && /tmp/clear_shapes > $OUT/clear_shapes.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/frame_shape.hip -o /tmp/frame_shape && /tmp/frame_shape > $OUT/frame_shape.txt 2>&1
python scripts/stamps_overlap.py trex1024 > $OUT/stamps_overlap_trex1024.txt 2>&1
python scripts/hostcost.py > $OUT/hostcost.txt 2>/dev/null
python scripts/k20_host.py > $OUT/k20_host.txt 2>/dev/null
for w in trex1024 bunny4096 trex8192 synth10m; do
  rm -rf gpurun_out/prof_$w
  s=20; [ $w = trex1024 ] && s=100; [ $w = synth10m ] && s=5
  scripts/profile_gpu.sh $w $s > $OUT/profile_$w.log 2>&1
  python scripts/summarize_prof.py gpurun_out/prof_$w | grep -v "at::native\|rocclr\|^void" > $OUT/rocprof_$w.txt
  cp gpurun_out/prof_$w/trace/*/*kernel_stats.csv $OUT/${w}_kernel_stats.csv 2>/dev/null
done
STAMPS_DEFS="-DCRENDER_DEV_KNOBS" python scripts/stamps.py trex1024 > $OUT/stamps_raster_trex1024.txt 2>&1
python scripts/stamps_setup.py > $OUT/stamps_setup_trex1024.txt 2>&1
ls $OUT
