#!/bin/bash
# GPU box: A/B of prebuilt libraries (scripts/ab/*.so; "cur" = the in-tree product library) x raster path
#   ARMS="cur:1 own8:1" WORKLOADS="bunny4096 trex8192" scripts/r6_variants.sh
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd)
OUT=gpurun_out/${OUTDIR:-r6v}; mkdir -p $OUT
line() { python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
v=d['roofline']['avg_launch_ms_views']
print('%-10s %-7s path=%-4s fps=%9.1f ms=%7.4f single_ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f b2b=%7.4f | %s events=%.4f b2b=%.4f' % (d['config']['workload'], os.environ['ABNAME'], os.environ['ABP'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster'], d['kernel_ms']['raster_back_to_back'], d['roofline']['kernel'], v.get('hip_events_around_each_launch') or 0, v.get('frames_back_to_back_on_one_stream') or v.get('single_stream_frame_minus_event_measured_bin_passes') or 0))"; }
for w in ${WORKLOADS:-bunny4096}; do
  s=${STEPS:-200}; [ $w = bunny4096 ] && s=50; [ $w = trex8192 ] && s=30; [ $w = synth10m ] && s=30
  for rep in 1 2 ${REPS:-}; do for arm in ${ARMS:-cur:auto}; do
    v=${arm%%:*}; p=${arm#*:}
    export ABNAME=$v ABP=$p
    if [ $v = cur ]; then unset CRENDER_LIB; else export CRENDER_LIB=$REPO/scripts/ab/$v.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-calls --workload $w --steps $s --warmup ${WARMUP:-10} --raster-path $p 2>$OUT/bench_err.log | line | tee -a $OUT/ab_variants.txt
  done; done
done
unset CRENDER_LIB
