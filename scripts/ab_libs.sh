#!/bin/bash
# GPU box: A/B of prebuilt libraries (scripts/ab/*.so, built in the container; git-ignored, they travel with gpurun)
#   LIBS="base cur" WORKLOADS="synth10m" TILES="0 16" STEPS=20 scripts/ab_libs.sh
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
v=d['roofline']['avg_launch_ms_views']
print('%-8s %-10s tile=%-4s fps=%9.1f ms=%7.4f single_ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f b2b=%7.4f | %s events=%.4f b2b=%.4f' % (os.environ['ABNAME'], d['config']['workload'], d['config']['tile'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster'], d['kernel_ms']['raster_back_to_back'], d['roofline']['kernel'], v.get('hip_events_around_each_launch') or 0, v.get('frames_back_to_back_on_one_stream') or v.get('single_stream_frame_minus_event_measured_bin_passes') or 0))"; }
for w in ${WORKLOADS:-synth10m}; do
  for t in ${TILES:-0}; do
    for v in ${LIBS:-base cur}; do
      export ABNAME=$v
      if [ $v = cur ]; then unset CRENDER_LIB; else export CRENDER_LIB=$(pwd)/scripts/ab/$v.so; fi
      for rep in 1 2; do python bench.py --no-cpu-baseline --no-api-calls --workload $w --tile $t --steps ${STEPS:-20} --warmup ${WARMUP:-5} 2>/dev/null | line; done
    done
  done
done
