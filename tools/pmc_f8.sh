#!/bin/bash
# PMC passes over tools/f8_one.py (counters only).  Usage: tools/pmc_f8.sh <tag> [BH]   -> gpurun_out/pmc_<tag>.md
TAG=${1:-f8}; export BH=${2:-32}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C -d $OUT/pass$i -o p --output-format csv -- python3 $ROOT/tools/f8_one.py 5 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT all > gpurun_out/pmc_$TAG.md 2>&1
grep -A32 "k_attn_fwd_f8" gpurun_out/pmc_$TAG.md | head -40
