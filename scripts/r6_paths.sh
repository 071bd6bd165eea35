#!/bin/bash
# GPU box: the parity suite under each raster kernel (CRENDER_RASTER_PATH: crender_set_default_raster_path), then
# a same-box A/B of the kernels on the workloads they are for -> gpurun_out/r6a/
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/${OUTDIR:-r6a}; mkdir -p $OUT
for p in ${PATHS:-auto 1 0}; do
  if [ $p = auto ]; then unset CRENDER_RASTER_PATH; else export CRENDER_RASTER_PATH=$p; fi
  timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_path_$p.log 2>&1; rc=$?
  echo "pytest path=$p rc=$rc: $(tail -1 $OUT/pytest_path_$p.log)"
  [ $rc -ne 0 ] && grep -E "^FAILED|^E  " $OUT/pytest_path_$p.log | cut -c1-300 | head -20
done
unset CRENDER_RASTER_PATH
line() { python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
v=d['roofline']['avg_launch_ms_views']
print('%-10s path=%-4s fps=%9.1f ms=%7.4f single_ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f b2b=%7.4f | %s %s' % (d['config']['workload'], os.environ['ABP'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster'], d['kernel_ms']['raster_back_to_back'], d['roofline']['kernel'], d['config']['raster_path']['last_launch_of_each_plan']))"; }
for spec in ${BENCH:-bunny4096:0,1,auto trex8192:0,1,auto synth10m:0,auto trex1024:0,auto}; do
  w=${spec%%:*}; ps=$(echo ${spec#*:} | tr ',' ' ')
  s=200; [ $w = bunny4096 ] && s=50; [ $w = trex8192 ] && s=30; [ $w = synth10m ] && s=30
  for rep in 1 2; do for p in $ps; do
    export ABP=$p
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-calls --workload $w --steps $s --warmup 10 --raster-path $p 2>$OUT/bench_err.log | line | tee -a $OUT/ab_paths.txt
  done; done
done
