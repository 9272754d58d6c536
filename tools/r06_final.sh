#!/bin/bash
# Round 6 evidence at one commit, one gpurun call: tools/r06_final.sh   (run on the GPU box; results under gpurun_out/r06f/)
O=gpurun_out/r06f; mkdir -p $O
# 1. HBM traffic + pipe counters of every 64^2 launch form at HEAD (bench.py's `roofline.traffic` reads the table this writes)
bash tools/traffic_at_head.sh r06 > $O/traffic.log 2>&1
cp gpurun_out/r06_attn_traffic.json profiles/r06_attn_traffic.json 2>/dev/null     # (on the box: so that the bench runs below find it)
cp gpurun_out/r06_attn_traffic.json $O/ 2>/dev/null; cp gpurun_out/pmc_r06_*.md $O/ 2>/dev/null
# 2. the bench lines
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20steps.json 2> /dev/null
GD_REF_AHEAD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp16-leg > $O/bench_20steps_no_ref_ahead.json 2> /dev/null
GD_REF_AHEAD=2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp16-leg > $O/bench_20steps_ref_carried.json 2> /dev/null
GD_PASS_TIMES=1 python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-fp16-leg 2>&1 > /dev/null | grep -A7 "device time per pass" > $O/pass_times.log
python bench.py --size 768 --steps 4 --warmup 3 --no-cpu-baseline --no-fp16-leg > $O/bench_768.json 2> /dev/null
python bench.py --model sdxl --size 1024 --steps 4 --warmup 3 --no-cpu-baseline --no-fp16-leg > $O/bench_sdxl_1024.json 2> /dev/null
for B in 4 8; do python bench.py --edits-per-pass $B --no-cpu-baseline --no-fp16-leg > $O/bench_epp$B.json 2> $O/bench_epp$B.err; done
# 3. the launch forms in both dtypes, back to back; the hand-off modes
bash tools/fp16_vs_bf16.sh 2>&1 | grep "us per launch" > $O/fp16_vs_bf16.log
( export TIME=1; for HS in 1 0; do echo "handoff=$HS"; HS=$HS QS=1 BH=5 python3 tools/attn_one.py 300; HS=$HS QS=1 BH=20 python3 tools/attn_one.py 300; done;
  for F in cfg cfg4n cfg4n_split; do FORM=$F python3 tools/attn_one.py 300; done ) 2>&1 | grep "us per launch\|handoff" > $O/attn_forms.log
# 4. the trace of the timed region (kernel table, in-situ 64^2 launch times, gaps) and the gap causes
STEPS=2 WARMUP=3 bash tools/profile_bench.sh r06 > $O/prof.log 2>&1
cp gpurun_out/r06_bench_summary.md gpurun_out/r06_bench_kernel_stats.csv $O/ 2>/dev/null
bash tools/profile_gaps.sh r06 > /dev/null 2>&1; cp gpurun_out/r06_gap_causes.md $O/ 2>/dev/null
# 5. parity table
python tools/parity_report.py > $O/parity_report.md 2>&1
rm -rf gpurun_out/prof_r06 gpurun_out/gaps_r06
ls -la $O
