#!/bin/bash
# Round-3 evidence at HEAD (on the GPU box): bench lines, kernel trace, PMC tables, kernel micro-benchmarks, phase timelines.
# Results under gpurun_out/r03f/ (copy into profiles/).  tools/build_dbg.sh 32 must have run before the snapshot (phase timelines).
cd "$(dirname "$0")/.."
O=gpurun_out/r03f; mkdir -p $O
timeout 600 python3 bench.py > $O/r03_bench_final.json 2> $O/bench_final.err
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_20steps.json 2> $O/bench_20.err
timeout 600 python3 bench.py --dtype fp16 --no-cpu-baseline > $O/r03_bench_fp16.json 2> $O/bench_fp16.err
timeout 600 python3 bench.py --size 768 --no-cpu-baseline > $O/r03_bench_768.json 2> $O/bench_768.err
timeout 900 python3 bench.py --model sdxl --size 1024 --no-cpu-baseline > $O/r03_bench_sdxl_1024.json 2> $O/bench_sdxl.err
GD_ATTN_FP8=1 timeout 900 python3 bench.py --model sdxl --size 1024 --no-cpu-baseline > $O/r03_bench_sdxl_1024_fp8.json 2> $O/bench_sdxl_fp8.err
timeout 900 tools/profile_bench.sh r03 > $O/profile_bench.log 2>&1
for bh in 5 15 20 32; do timeout 600 tools/pmc_attn.sh r03_qs_bh$bh $bh "" 1 > $O/pmc_qs_bh$bh.log 2>&1; done
FORM=cfg timeout 600 tools/pmc_attn.sh r03_cfg20 20 "" 1 > $O/pmc_cfg20.log 2>&1
timeout 300 python3 tools/bench_sk.py > $O/r03_attention_kernels.log 2>&1
timeout 300 python3 tools/bench_cfg20.py > $O/r03_cfg20.log 2>&1
timeout 300 python3 tools/bench_opt15.py > $O/r03_opt15.log 2>&1
timeout 300 python3 tools/bench_handoff.py > $O/r03_handoff.log 2>&1
GD_LIB=tools/ub/build/libgd_dbg32.so timeout 300 python3 tools/w64_phases.py > $O/r03_w64_phases.log 2>&1
GD_LIB=tools/ub/build/libgd_dbg32.so timeout 300 python3 tools/mp_phases.py > $O/r03_mp_phases.log 2>&1
cp gpurun_out/r03_bench_summary.md gpurun_out/r03_bench_kernel_stats.csv gpurun_out/pmc_r03_*.md $O/ 2>/dev/null
ls $O
