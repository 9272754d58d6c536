#!/bin/bash
# Probe: does MIOpen's tuning search (MIOPEN_FIND_ENFORCE=4) beat its heuristic kernel configurations on the UNet's convolution shapes,
# and what does the search cost per shape?  Writes gpurun_out/miopen_tune_probe.log.
mkdir -p gpurun_out /tmp/mi_a /tmp/mi_b
L=gpurun_out/miopen_tune_probe.log; : > $L
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
run() {  # n c hw k
  ARGS="convbfp16 -n $1 -c $2 -H $3 -W $3 -k $4 -y 3 -x 3 -p 1 -q 1 -u 1 -v 1 -l 1 -j 1 -F 1 -t 1 -i 30 -V 0 --in_layout NHWC --fil_layout NHWC --out_layout NHWC"
  echo "== n=$1 c=$2 hw=$3 k=$4" >> $L
  s=$(date +%s.%N)
  MIOPEN_USER_DB_PATH=/tmp/mi_a MIOPEN_CUSTOM_CACHE_DIR=/tmp/mi_a/cache timeout 300 /opt/rocm/bin/MIOpenDriver $ARGS 2>&1 | grep -E "Elapsed|Algorithm|Solution" >> $L
  e=$(date +%s.%N); echo "default find: $(echo "$e - $s" | bc) s wall" >> $L
  s=$(date +%s.%N)
  MIOPEN_FIND_ENFORCE=4 MIOPEN_USER_DB_PATH=/tmp/mi_b MIOPEN_CUSTOM_CACHE_DIR=/tmp/mi_b/cache timeout 600 /opt/rocm/bin/MIOpenDriver $ARGS 2>&1 | grep -E "Elapsed|Algorithm|Solution" >> $L
  e=$(date +%s.%N); echo "tuned find:   $(echo "$e - $s" | bc) s wall" >> $L
  s=$(date +%s.%N)
  MIOPEN_USER_DB_PATH=/tmp/mi_b MIOPEN_CUSTOM_CACHE_DIR=/tmp/mi_b/cache timeout 300 /opt/rocm/bin/MIOpenDriver $ARGS 2>&1 | grep -E "Elapsed|Algorithm|Solution" >> $L
  e=$(date +%s.%N); echo "after tuning: $(echo "$e - $s" | bc) s wall" >> $L
}
run 1 320 64 320
run 3 320 64 320
run 3 640 32 640
run 3 1280 16 1280
run 3 960 64 320
cat /tmp/mi_b/*.udb.txt >> $L 2>/dev/null
