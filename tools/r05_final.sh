#!/bin/bash
# Round 5 evidence at one commit, one gpurun call: tools/r05_final.sh   (run on the GPU box; results under gpurun_out/r05f/)
O=gpurun_out/r05f; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/gpu_tests.log
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp16-leg > $O/bench_20steps.json 2> /dev/null
for B in 2 4 8; do python bench.py --edits-per-pass $B --steps $((B > 4 ? 3 : 4)) --warmup 3 --no-cpu-baseline > $O/bench_epp$B.json 2> $O/bench_epp$B.err; done
STEPS=2 WARMUP=3 bash tools/profile_bench.sh r05 > $O/prof.log 2>&1
STEPS=1 WARMUP=3 BENCH_ARGS="--edits-per-pass 4" bash tools/profile_bench.sh r05_epp4 > $O/prof_epp4.log 2>&1
cp gpurun_out/r05_bench_summary.md gpurun_out/r05_bench_kernel_stats.csv gpurun_out/r05_epp4_bench_summary.md gpurun_out/r05_epp4_bench_kernel_stats.csv $O/ 2>/dev/null
python tools/probe_batch_scaling.py bf16 > $O/batch_scaling.log 2>&1
python tools/loop_error_budget.py --kinds cfg1_t50 cfg1_full_t50 --reps 3 --out $O/loop_error_budget.md > $O/budget.log 2>&1
python tools/parity_report.py > $O/parity_report.md 2> $O/parity_report.err
python tools/corr_variants.py > $O/corr_variants.log 2>&1
FORM=cfgb4 bash tools/pmc_attn.sh r05_cfg_b4 80 "" 1 > $O/pmc_cfg_b4.log 2>&1; cp gpurun_out/pmc_r05_cfg_b4.md $O/ 2>/dev/null; rm -rf gpurun_out/pmc_r05_cfg_b4
rm -rf gpurun_out/prof_r05 gpurun_out/prof_r05_epp4
ls -la $O
