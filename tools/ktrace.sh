#!/bin/bash
# kernel durations of a small driver script under rocprofv3 --kernel-trace --stats.  Usage: tools/ktrace.sh <tag> <script.py> [args]   (GPU box)
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ktrace_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $ROOT/$@ > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print("| kernel | calls | avg us | min us | max us |\n|---|---|---|---|---|")
for r in rows:
    n = r["Name"]
    if n.startswith(("k_", "void k_", "_Z")) and "at::" not in n:
        print(f"| \`{n[:64]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
PY
