#!/bin/bash
# PMC passes over tools/attn_one.py (one process per pass; counters only, no tracing domains).  Usage: tools/pmc_attn.sh <tag> [BH] [cfg]
# Writes gpurun_out/pmc_<tag>/pass*/ and gpurun_out/pmc_<tag>.md
TAG=${1:-x}; export BH=${2:-32}; if [ -n "$3" ]; then export GD_ATTN_CFG=$3; fi; export QS=${4:-0}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C -d $OUT/pass$i -o p --output-format csv -- python3 $ROOT/tools/attn_one.py 6 > $OUT/pass$i.log 2>&1 || echo "pass $i failed (see $OUT/pass$i.log)"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > gpurun_out/pmc_$TAG.md 2>&1
cat gpurun_out/pmc_$TAG.md
