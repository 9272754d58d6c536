#!/bin/bash
# PMC passes over tools/conv_one.py (one process per pass; counters only).  Usage: tools/pmc_conv.sh <tag> [n,C,H,K]
TAG=${1:-conv}; export CONV_SHAPE=${2:-3,1920,32,640}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C -d $OUT/pass$i -o p --output-format csv -- python3 $ROOT/tools/conv_one.py 4 > $OUT/pass$i.log 2>&1 || echo "pass $i failed (see $OUT/pass$i.log)"
done
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/tools/conv_one.py 20 > $OUT/trace.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT all > gpurun_out/pmc_$TAG.md 2>&1
python3 - <<PY >> gpurun_out/pmc_$TAG.md
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
print("\n## kernel trace (rocprofv3 --kernel-trace --stats, 20 repetitions)\n\n| kernel | calls | avg us |\n|---|---|---|")
if f:
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        if "conv" in n:
            print(f"| \`{n[:70]}\` | {r['Calls']} | {float(r['AverageNs']) / 1e3 if 'AverageNs' in r else float(r.get('AverageUs', 0)):.1f} |")
PY
grep -v "^$" gpurun_out/pmc_$TAG.md | head -70
