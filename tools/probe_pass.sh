#!/bin/bash
# per-kernel totals of a captured vanilla UNet pass at two batch sizes (default 3 and 4) -> gpurun_out/probe_pass_<a>_<b>.md
A=${1:-3}; B=${2:-4}; ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
for NB in $A $B; do
  export NB REPS=40
  rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/pp_$NB -o p --output-format csv -- python3 $ROOT/tools/probe_pass.py > $ROOT/gpurun_out/pp_$NB.log 2>&1
done
cd $ROOT
python3 - "$A" "$B" > gpurun_out/probe_pass_${A}_${B}.md <<'PY'
import csv, glob, sys, collections
sys.path.insert(0, "tools")
from prof_summary import base_name
a, b = sys.argv[1], sys.argv[2]
def load(nb):
    f = glob.glob(f"gpurun_out/pp_{nb}/**/*kernel_stats.csv", recursive=True)[0]
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Name"][:60] if r["Name"].startswith("Cijk") else base_name(r["Name"])
        d[k][0] += int(r["Calls"]); d[k][1] += float(r["TotalDurationNs"]) * 1e-6
    return d
da, db = load(a), load(b)
print(open(f"gpurun_out/pp_{a}.log").read().strip().splitlines()[-1]); print(open(f"gpurun_out/pp_{b}.log").read().strip().splitlines()[-1])
print(f"\n| kernel | calls b{a} | ms b{a} | calls b{b} | ms b{b} | delta ms per pass |\n|---|---|---|---|---|---|")
rows = sorted(set(da) | set(db), key=lambda k: -(db.get(k, [0, 0])[1] - da.get(k, [0, 0])[1]))
n = 45.0   # 2 eager + capture(0) + 3 + 40 replays: per-pass figures are approximate
for k in rows[:25] + rows[-8:]:
    ca, ta = da.get(k, [0, 0.0]); cb, tb = db.get(k, [0, 0.0])
    print(f"| `{k}` | {ca} | {ta:.1f} | {cb} | {tb:.1f} | {(tb - ta) / n:+.3f} |")
print(f"\ntotal kernel ms: b{a} {sum(v[1] for v in da.values()):.1f}, b{b} {sum(v[1] for v in db.values()):.1f}")
PY
rm -rf gpurun_out/pp_$A gpurun_out/pp_$B
cat gpurun_out/probe_pass_${A}_${B}.md
