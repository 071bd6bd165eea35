#!/bin/bash
# GPU box: the small records' kernel with and without row spans against the general kernel; PMC instruction counts
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd)
OUT=gpurun_out/${OUTDIR:-r6b}; mkdir -p $OUT
CRENDER_RASTER_PATH=2 timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_path_2.log 2>&1; rc=$?
echo "pytest path=2 rc=$rc: $(tail -1 $OUT/pytest_path_2.log)"
[ $rc -ne 0 ] && exit 1
line() { python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('%-10s %-8s path=%-4s fps=%9.1f ms=%7.4f single_ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f b2b=%7.4f | %s' % (d['config']['workload'], os.environ['ABNAME'], os.environ['ABP'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster'], d['kernel_ms']['raster_back_to_back'], d['roofline']['kernel']))"; }
for w in ${WORKLOADS:-synth10m trex1024}; do
  s=200; [ $w = bunny4096 ] && s=50; [ $w = trex8192 ] && s=30; [ $w = synth10m ] && s=30
  for rep in 1 2; do for arm in ${ARMS:-cur:0 cur:2 nospans:2}; do
    v=${arm%%:*}; p=${arm#*:}
    export ABNAME=$v ABP=$p
    if [ $v = cur ]; then unset CRENDER_LIB; else export CRENDER_LIB=$REPO/scripts/ab/$v.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-calls --workload $w --steps $s --warmup 10 --raster-path $p 2>$OUT/bench_err.log | line | tee -a $OUT/ab_spans.txt
  done; done
done
unset CRENDER_LIB
# instruction counts per launch
for w in ${PMCW:-synth10m}; do
  for p in 0 2; do
    out=$REPO/$OUT/pmc_${w}_$p; rm -rf $out
    s=10; [ $w = trex1024 ] && s=50
    (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- python3 $REPO/bench.py --workload $w --steps $s --warmup 3 --no-cpu-baseline --no-api-calls --no-pipeline --raster-path $p > $out.log 2>&1
     TMPDIR=/tmp timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_lds -- python3 $REPO/bench.py --workload $w --steps $s --warmup 3 --no-cpu-baseline --no-api-calls --no-pipeline --raster-path $p >> $out.log 2>&1)
    python scripts/summarize_prof.py $out 2>/dev/null | grep -E "^k_raster|^k_frame|^k_bin|^## " | sed "s/^/[$w path=$p] /" | tee -a $OUT/pmc_summary.txt
  done
done
