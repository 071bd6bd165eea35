#!/usr/bin/env python3
"""profiles/kernel_avg.json from the rocprofv3 kernel traces of a round's profiling runs
(gpurun_out/prof_<workload>[_pipelined]/trace/**/kernel_trace.csv): per workload the average launch
duration in ns of
  k_raster             MODE=single run (frames one at a time on one stream)
  k_frame_one_stream   MODE=pipelined run, the dispatches that overlap no other k_frame (bench.py's
                       probe pass: the same launches one after another on one stream)
  k_frame_overlapped   the same run, the dispatches in flight together on the swap chain's streams
bench.py never reports a launch duration below these (roofline.avg_launch_ms_views)."""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import split_by_overlap  # noqa: E402

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = {}
for wl in ("trex1024", "bunny4096", "trex8192", "synth10m", "cube256"):
    e = {}
    for f in glob.glob(os.path.join(root, f"prof_{wl}", "trace", "**", "*kernel_trace.csv"), recursive=True):
        r = split_by_overlap(f, quiet=True)
        for key in ("k_raster", "k_raster_owners"):          # (general kernel / pixel owners' kernel: csrc/raster.hip kPath*)
            if key in r:
                v = r[key][0] + r[key][1]
                e[key] = sum(v) / len(v)
                e[key + "_calls"] = len(v)
    for f in glob.glob(os.path.join(root, f"prof_{wl}_pipelined", "trace", "**", "*kernel_trace.csv"), recursive=True):
        r = split_by_overlap(f, quiet=True)
        for key in ("k_frame", "k_frame_owners"):
            if key not in r:
                continue
            alone, over, scratch = r[key]
            if alone:
                e[key + "_one_stream"] = sum(alone) / len(alone)
                e[key + "_one_stream_calls"] = len(alone)
            if over:
                e[key + "_overlapped"] = sum(over) / len(over)
                e[key + "_overlapped_calls"] = len(over)
            e[key + "_scratch_bytes"] = scratch
    if e:
        e["source"] = "rocprofv3 --kernel-trace --stats of bench.py (scripts/profile_gpu.sh), ns"
        out[wl] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cython3dmodelrenderer_amd import _build  # noqa: E402
out["csrc_sha16"] = _build.source_sha16()       # the kernels these figures were measured on (bench.py checks it)
json.dump(out, open("profiles/kernel_avg.json", "w"), indent=1)
print(json.dumps(out, indent=1))
