#!/bin/bash
# round 4, after a change: the GPU tests, the bench line and the kernel trace of the timed region.  Usage: tools/r04_check.sh <tag> [pytest -k expr]   (GPU box)
TAG=${1:-r04x}; KEXPR=${2:-}
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then python -m pytest tests -q -m gpu -k "$KEXPR" 2>&1 | tail -12 > gpurun_out/${TAG}_tests.log
else python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/${TAG}_tests.log; fi
python bench.py --steps 8 --warmup 4 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
tail -4 gpurun_out/${TAG}_tests.log; python3 - <<PY
import json
l = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
r = l.get("roofline", {})
print("ms/edit", round(l["ms_per_step"], 1), "edits/min", round(60 * l["value"], 2), "frac", round(r.get("frac", 0), 3), "alg", round(r.get("frac_algorithmic", 0), 3))
PY
grep -n "hooked layer\|idle gaps >= 1\|span " gpurun_out/${TAG}_bench_summary.md
