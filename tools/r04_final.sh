#!/bin/bash
# Round-4 evidence at HEAD (on the GPU box): bench lines, kernel trace, PMC tables, kernel micro-benchmarks.  Results under gpurun_out/r04f/ (copy into profiles/).
cd "$(dirname "$0")/.."
O=gpurun_out/r04f; mkdir -p $O
timeout 600 python3 bench.py > $O/r04_bench_final.json 2> $O/bench_final.err
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_20steps.json 2> $O/bench_20.err
timeout 600 python3 bench.py --dtype fp16 --no-cpu-baseline > $O/r04_bench_fp16.json 2> $O/bench_fp16.err
timeout 600 python3 bench.py --size 768 --no-cpu-baseline > $O/r04_bench_768.json 2> $O/bench_768.err
timeout 900 python3 bench.py --model sdxl --size 1024 --no-cpu-baseline > $O/r04_bench_sdxl_1024.json 2> $O/bench_sdxl.err
STEPS=2 WARMUP=4 timeout 900 tools/profile_bench.sh r04 > $O/profile_bench.log 2>&1
for bh in 5 15 32; do timeout 600 tools/pmc_attn.sh r04_qs_bh$bh $bh "" 1 > $O/pmc_qs_bh$bh.log 2>&1; done
FORM=cfg timeout 600 tools/pmc_attn.sh r04_cfg20 20 "" 1 > $O/pmc_cfg20.log 2>&1
FORM=opt timeout 600 tools/pmc_attn.sh r04_opt15 15 "" 1 > $O/pmc_opt15.log 2>&1
python3 tools/attn_traffic.py $O/r04_attn_traffic.json 5=gpurun_out/pmc_r04_qs_bh5 15_plain=gpurun_out/pmc_r04_qs_bh15 32=gpurun_out/pmc_r04_qs_bh32 20=gpurun_out/pmc_r04_cfg20 15=gpurun_out/pmc_r04_opt15 > $O/attn_traffic.log 2>&1
timeout 900 tools/pmc_bwd.sh r04_bwd > $O/pmc_bwd.log 2>&1
timeout 900 tools/pmc_hbm.sh r04 > $O/pmc_hbm.log 2>&1
timeout 300 python3 tools/bench_corr.py > $O/r04_corr_max.log 2>&1
timeout 300 python3 tools/bench_corr2.py 24 > $O/r04_corr_max_rounds.log 2>&1
timeout 300 python3 tools/bench_bwd.py > $O/r04_bwd_kernels.log 2>&1
timeout 300 python3 tools/bench_dq2.py > $O/r04_dq2_runs.log 2>&1
timeout 300 python3 tools/bench_cross.py > $O/r04_cross_attention.log 2>&1
timeout 300 python3 tools/opt_pass_kernels.py > $O/r04_opt_pass_kernels.log 2>&1
timeout 300 python3 tools/bench_lsum.py > $O/r04_lsum_default.log 2>&1
GD_ATTN_LSUM=2 timeout 300 python3 tools/bench_lsum.py > $O/r04_lsum_all.log 2>&1
GD_ATTN_LSUM=0 timeout 300 python3 tools/bench_lsum.py > $O/r04_lsum_off.log 2>&1
timeout 600 python3 tools/parity_report.py > $O/r04_parity_report.md 2>&1
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/r04_gpu_tests.log
cp gpurun_out/r04_bench_summary.md gpurun_out/r04_bench_kernel_stats.csv gpurun_out/pmc_r04_*.md gpurun_out/r04_hbm.md $O/ 2>/dev/null
ls $O
