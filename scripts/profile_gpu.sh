#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + HBM counters for one workload.
#   [MODE=single|pipelined] scripts/profile_gpu.sh <workload> <steps> [extra bench args]
# MODE=single (default): --no-pipeline, every frame k_setup_wave / k_count_wave.. -> k_raster on one
#   stream, so a kernel's duration is its own.
# MODE=pipelined: the bench command as the driver runs it — the swap chain's k_frame launches of the
#   timed region (overlapping: each looks ~4x longer than its share of the machine) AND, in the same
#   process, bench.py's probe pass that runs the same k_frame launches one after another on one
#   stream; scripts/summarize_prof.py tells the two apart by whether a dispatch overlaps another.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit together,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); no trace domain is combined with --pmc.
set -u
WL=${1:-trex1024}; STEPS=${2:-50}; shift 2 || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
MODE=${MODE:-single}
OUT=$REPO/gpurun_out/prof_$WL${SUFFIX:-}; [ $MODE = pipelined ] && OUT=$REPO/gpurun_out/prof_${WL}_pipelined${SUFFIX:-}
# (SUFFIX=_general with extra args "--raster-path 0": the same command on the other raster kernel, into a directory of its own;
#  TRACE_ONLY=1: the kernel trace alone)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
NOPIPE=--no-pipeline; [ $MODE = pipelined ] && NOPIPE=
ARGS="$REPO/bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline --no-api-calls $NOPIPE $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
[ -n "${TRACE_ONLY:-}" ] && exit 0
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 $ARGS > "$OUT/pmc_$C.log" 2>&1
  echo "pmc $C rc=$?"
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 $ARGS > "$OUT/pmc_sq.log" 2>&1
echo "pmc sq rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_FLAT --output-format csv -d "$OUT/pmc_lds" -- python3 $ARGS > "$OUT/pmc_lds.log" 2>&1
echo "pmc lds rc=$?"
find "$OUT" -name "*.csv" | head -40
