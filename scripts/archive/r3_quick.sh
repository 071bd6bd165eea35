#!/bin/bash
# GPU box: tests + the five bench lines -> gpurun_out/$1
OUT=gpurun_out/${1:-r3}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
python bench.py > $OUT/bench_trex1024.json 2> $OUT/bench.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-calls > $OUT/bench_trex1024_k20.json 2>/dev/null
python bench.py --workload bunny4096 --steps 50 --warmup 5 --no-api-calls > $OUT/bench_bunny4096.json 2>/dev/null
python bench.py --workload trex8192 --steps 30 --warmup 3 --no-api-calls > $OUT/bench_trex8192.json 2>/dev/null
python bench.py --workload synth10m --steps 10 --warmup 2 --no-api-calls > $OUT/bench_synth10m.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read())
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]
    print(f.split("/")[-1], "fps=%.1f ms=%.4f single=%.4f kernel=%s frac=%.3f views=%s bin=%.4f ras=%.4f" % (d["value"], d["ms_per_step"], d["ms_per_frame_single_stream"], r["kernel"], r["frac"], {k:round(v,4) for k,v in r["avg_launch_ms_views"].items()}, d["kernel_ms"]["binning_passes"], d["kernel_ms"]["raster"]))
PY
