#!/bin/bash
# kernel + memory-copy trace of the benchmark's timed region -> gpurun_out/<tag>_gap_causes.md   (tools/gap_causes.py; on the GPU box)
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/gaps_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export GD_BENCH_MARK=1 GD_PREPASS_THREAD=${GD_PREPASS_THREAD:-0}
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o bench --output-format csv -- python3 $ROOT/bench.py --steps ${STEPS:-2} --warmup ${WARMUP:-3} --no-cpu-baseline --no-fp16-leg > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
KT=$(find $OUT -name "*kernel_trace.csv" | head -1); MT=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
python3 tools/gap_causes.py $KT $MT > gpurun_out/${TAG}_gap_causes.md
head -3 $MT > gpurun_out/${TAG}_memcopy_head.csv
rm -rf $OUT
head -70 gpurun_out/${TAG}_gap_causes.md
