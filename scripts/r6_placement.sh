#!/bin/bash
# GPU box: T-Rex 8192^2 framebuffer placement (scripts/placement_probe.py) -> gpurun_out/r6p/
cd ${GRAFT_REPO_ROOT:-.}
REPO=$(pwd)
OUT=$REPO/gpurun_out/${OUTDIR:-r6p}; mkdir -p $OUT
for m in ${MODES:-torch arena late reuse torch}; do
  timeout -k 10 200 python scripts/placement_probe.py $m 4 2>/dev/null | tee -a $OUT/placement.txt
done
# which counters does this box have?  (translation, write-request stalls)
(cd /tmp && TMPDIR=/tmp timeout -k 10 120 rocprofv3 -L > $OUT/counters_list.txt 2>&1)
grep -o "\b\(TCP_UTCL1[A-Z0-9_]*\|TCP_UTCL2[A-Z0-9_]*\|TCC_EA0_WRREQ[A-Z0-9_]*\|TCC_TOO_MANY_EA_WRREQS_STALL[A-Z0-9_]*\|TCP_TCC_WRITE_REQ[A-Z0-9_]*\|TCP_PENDING_STALL_CYCLES[A-Z0-9_]*\|TCC_WRITEBACK[A-Z0-9_]*\|TCC_EA0_WR_UNCACHED_32B[A-Z0-9_]*\|TCC_EA0_ATOMIC[A-Z0-9_]*\|TCC_NORMAL_WRITEBACK[A-Z0-9_]*\|TCC_TAG_STALL[A-Z0-9_]*\|TCC_BUSY[A-Z0-9_]*\|TCP_UTCL1_TRANSLATION_MISS\)\b" $OUT/counters_list.txt | sort -u > $OUT/counters_have.txt
echo "counters on this box: $(wc -l < $OUT/counters_have.txt)"; head -60 $OUT/counters_have.txt | tr '\n' ' '; echo
pass() {  # name, counters...
  local name=$1; shift
  local have=""
  for c in "$@"; do grep -qx "$c" $OUT/counters_have.txt && have="$have $c"; done
  [ -z "$have" ] && { echo "pass $name: none of its counters exist here"; return; }
  rm -rf $OUT/pmc_$name
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc $have --output-format csv -d $OUT/pmc_$name -- python3 $REPO/scripts/placement_probe.py pmc 4 > $OUT/pmc_$name.log 2>&1)
  python3 - $OUT/pmc_$name <<'PY' | tee -a $OUT/placement.txt
import csv, glob, sys, collections
d = sys.argv[1]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not tr or not cc:
    print("  (no csv in", d, ")"); sys.exit(0)
dur = {}
for r in csv.DictReader(open(tr[0])):
    if "k_raster" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = collections.OrderedDict()
for r in csv.DictReader(open(cc[0])):
    if "k_raster" in r["Kernel_Name"]:
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(rows)
K = 4
names = sorted({c for v in rows.values() for c in v})
print("  per set (dispatch i goes into set i mod %d), averages over %d clears each:" % (K, len(ids) // K))
for k in range(K):
    sel = ids[k::K][1:]        # (skip the first, cold one)
    t = [dur[i] for i in sel if i in dur]
    line = "  set %d  dur_us %7.1f" % (k, (sum(t) / len(t) / 1e3) if t else -1)
    for c in names:
        v = [rows[i].get(c, 0.0) for i in sel]
        line += "  %s %.4g" % (c, sum(v) / len(v))
    print(line)
PY
}
if [ -z "${PMC_ONLY:-}" ]; then
pass utcl1 TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_PERMISSION_MISS
fi
pass wrreq TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL
pass wrdest TCC_EA0_WRREQ_WRITE_DRAM TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_GMI_32B TCC_EA0_WRREQ_WRITE_IO_32B
pass wrcredit TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_IO_CREDIT_STALL TCC_EA0_WRREQ_GMI_CREDIT_STALL TCC_EA0_WRREQ_DRAM
pass tcp TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCC_EA0_WRREQ_LEVEL
pass tcc TCC_BUSY TCC_TAG_STALL TCC_WRITEBACK TCC_NORMAL_WRITEBACK
