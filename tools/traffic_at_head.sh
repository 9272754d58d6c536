#!/bin/bash
# HBM traffic + pipe-utilisation counters of the 64^2 attention forward at HEAD, every launch form of an edit, one gpurun call:
#   tools/traffic_at_head.sh <round tag, e.g. r06>      (on the GPU box; writes gpurun_out/pmc_<tag>_*.md and gpurun_out/<tag>_attn_traffic.json)
# Forms: 5 / 15 / 32 plain heads with pre-scaled queries (5 = the inversion pass's launch), the CFG pass's launch (20 heads, 4 segments,
# fused warp + row list), the optimisation pass's launch (15 heads, 3 segments, LSE, matrix-pipe row sums).
R=${1:-r06}
bash tools/pmc_attn.sh ${R}_qs_bh5 5 "" 1 > /dev/null
bash tools/pmc_attn.sh ${R}_qs_bh15 15 "" 1 > /dev/null
bash tools/pmc_attn.sh ${R}_qs_bh32 32 "" 1 > /dev/null
FORM=cfg bash tools/pmc_attn.sh ${R}_cfg20 20 > /dev/null
FORM=opt bash tools/pmc_attn.sh ${R}_opt15 15 > /dev/null
python3 tools/attn_traffic.py gpurun_out/${R}_attn_traffic.json 5=gpurun_out/pmc_${R}_qs_bh5 15_plain=gpurun_out/pmc_${R}_qs_bh15 32=gpurun_out/pmc_${R}_qs_bh32 \
    20=gpurun_out/pmc_${R}_cfg20 15=gpurun_out/pmc_${R}_opt15
find gpurun_out -name "pass*" -path "*pmc_${R}_*" -prune -exec rm -rf {} + 2>/dev/null   # raw counter CSVs: tens of MB
