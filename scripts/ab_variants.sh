#!/bin/bash
# A/B of library builds x CRENDER_DEBUG values on the GPU box, one process per arm:
#   VARIANTS="name[:extra hipcc defines]..." DBGS="0 1024" WORKLOADS="trex1024" scripts/ab_variants.sh
# every name builds the working tree with -DCRENDER_DEV_KNOBS plus the given defines.  Per arm: bench.py line (pipelined
# frames/s, single-stream frame, event-timed passes), rocprofv3 kernel averages of a single-stream
# run, and one PMC pass (instruction counts).
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd)
mkdir -p /tmp/abv
for v in ${VARIANTS:-cur}; do
  name=${v%%:*}; defs=""; [ "$v" != "$name" ] && defs=$(echo "${v#*:}" | tr ',' ' ')
  nodev=""; [ -n "${NODEV:-}" ] && nodev=--no-dev-knobs
  scripts/dev_build.sh $nodev $defs --out /tmp/abv/$name.so > /dev/null || echo "BUILD FAILED $name"
done
line() { python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('%-10s dbg=%-5s %-10s fps=%9.1f ms=%7.4f single_ms=%7.4f bin_ms=%7.4f raster_ms=%7.4f' % (os.environ['ABNAME'], os.environ.get('CRENDER_DEBUG','0'), d['config']['workload'], d['value'], d['ms_per_step'], d['ms_per_frame_single_stream'], d['kernel_ms']['binning_passes'], d['kernel_ms']['raster']))"; }
for w in ${WORKLOADS:-trex1024}; do
  s=300; [ $w = bunny4096 ] && s=40; [ $w = trex8192 ] && s=20; [ $w = synth10m ] && s=5
  for v in ${VARIANTS:-cur}; do
    name=${v%%:*}; export ABNAME=$name CRENDER_LIB=/tmp/abv/$name.so
    for g in ${DBGS:-0}; do
      export CRENDER_DEBUG=$g
      for rep in 1 2; do python bench.py --no-cpu-baseline --no-api-calls --workload $w --steps $s --warmup 3 2>/dev/null | line; done
      if [ -n "${PROF:-}" ]; then
        out=/tmp/abv/prof_${name}_$g; rm -rf $out
        (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $REPO/bench.py --workload $w --steps 100 --warmup 3 --no-cpu-baseline --no-api-calls --no-pipeline > $out.log 2>&1
         TMPDIR=/tmp rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- python3 $REPO/bench.py --workload $w --steps 50 --warmup 3 --no-cpu-baseline --no-api-calls --no-pipeline >> $out.log 2>&1)
        [ -n "${PROF2:-}" ] && (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_lds -- python3 $REPO/bench.py --workload $w --steps 50 --warmup 3 --no-cpu-baseline --no-api-calls --no-pipeline >> $out.log 2>&1)
        python scripts/summarize_prof.py $out 2>/dev/null | grep -E "^k_|^## kernel" | sed "s/^/   [$name dbg=$g] /"
      fi
    done
  done
done
