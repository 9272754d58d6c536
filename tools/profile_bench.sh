#!/bin/bash
# rocprofv3 kernel trace of the benchmark's timed region + summary.  Usage: tools/profile_bench.sh <tag>   (run on the GPU box)
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export GD_BENCH_MARK=1
# one traced run with two launching host threads died inside the tracer at exit (heap corruption, then a hung signal handler), so the
# traced run keeps the pre-pass on the caller's thread (side stream only); profiles/README.md has the bench A/B of the thread
export GD_PREPASS_THREAD=${GD_PREPASS_THREAD:-0}
rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python3 $ROOT/bench.py --steps ${STEPS:-2} --warmup ${WARMUP:-3} --no-cpu-baseline ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
TR=$(find $OUT -name "*kernel_trace.csv" | head -1)
GD_PROF_EDITS=${STEPS:-2} python3 tools/prof_summary.py $TR gpurun_out/${TAG}_bench_summary.md gpurun_out/${TAG}_bench_kernel_stats.csv "$TAG - rocprofv3 --kernel-trace --stats of bench.py --steps ${STEPS:-2} --warmup ${WARMUP:-3} --no-cpu-baseline (MI355X, bf16)"
rm -f $TR   # hundreds of MB
cp $OUT/bench.json gpurun_out/${TAG}_bench_traced.json 2>/dev/null   # the bench line of the SAME (traced) run: its replay times beside the in-situ ones above
head -60 gpurun_out/${TAG}_bench_summary.md
