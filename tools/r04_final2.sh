#!/bin/bash
# Round-4 closing run at the last commit (GPU box): the bench lines, the kernel trace, the cross-attention micro-benchmark, the tests.  The PMC tables
# and kernel micro-benchmarks of tools/r04_final.sh are not repeated (those kernels did not change afterwards).
cd "$(dirname "$0")/.."
O=gpurun_out/r04g2; mkdir -p $O
timeout 600 python3 bench.py > $O/r04_bench_final.json 2> $O/bench_final.err
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_20steps.json 2> $O/bench_20.err
timeout 600 python3 bench.py --dtype fp16 --no-cpu-baseline > $O/r04_bench_fp16.json 2> $O/bench_fp16.err
timeout 600 python3 bench.py --size 768 --no-cpu-baseline > $O/r04_bench_768.json 2> $O/bench_768.err
timeout 900 python3 bench.py --model sdxl --size 1024 --no-cpu-baseline > $O/r04_bench_sdxl_1024.json 2> $O/bench_sdxl.err
STEPS=2 WARMUP=4 timeout 900 tools/profile_bench.sh r04 > $O/profile_bench.log 2>&1
timeout 300 python3 tools/bench_cross.py > $O/r04_cross_attention.log 2>&1
timeout 300 python3 tools/opt_pass_kernels.py > $O/r04_opt_pass_kernels.log 2>&1
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/r04_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
cp gpurun_out/r04_bench_summary.md gpurun_out/r04_bench_kernel_stats.csv $O/ 2>/dev/null
tail -2 $O/r04_gpu_tests.log; tail -1 $O/smoke.log; grep -n "span\|hooked layer\|idle gaps >= 1" $O/r04_bench_summary.md
