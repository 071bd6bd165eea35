#!/bin/bash
# GPU box: does crender_plan_debug_check find the cross-frame-state defects of rounds 4 and 5 at the frame that
# CAUSES them?  Two libraries with one defect each compiled back in (scripts/ab/fault1.so: hand-off words not reset
# when the pixel owners take a split tile, commit 30ca3df's fix removed; fault2.so: a binning discarded by the
# look-ahead leaves its split flags and helper slots, f924058's fix removed), the tests that found them by pixels.
# The two libraries are built in the container first (git-ignored, they travel with gpurun):
#   scripts/dev_build.sh --no-dev-knobs -DCRENDER_FAULT=1 --out scripts/ab/fault1.so
#   scripts/dev_build.sh --no-dev-knobs -DCRENDER_FAULT=2 --out scripts/ab/fault2.so
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/${OUTDIR:-r6c}; mkdir -p $OUT
for f in 1 2; do
  export CRENDER_LIB=$(pwd)/scripts/ab/fault$f.so
  CRENDER_RASTER_PATH=0 timeout -k 10 600 python -m pytest tests/test_hip_parity_gpu.py -m gpu -q \
     -k "test_dispatch_order_hint_never_changes_pixels or test_lone_chain_through_changing_scenes or test_fuzz_many_frames_on_the_same_plans" \
     > $OUT/pytest_fault$f.log 2>&1
  echo "fault $f: rc=$? $(tail -1 $OUT/pytest_fault$f.log)"
  grep -E "^FAILED|^E .*plan state" $OUT/pytest_fault$f.log | cut -c1-420 | head -12
done
unset CRENDER_LIB
