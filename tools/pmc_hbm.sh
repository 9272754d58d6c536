#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (one counter per pass: counters only, no tracing domains) + a kernel trace of tools/hbm_one.py: the HBM-bound rows of the
# path with counter bytes beside algorithmic bytes.  Usage: tools/pmc_hbm.sh <tag>   (GPU box) -> gpurun_out/<tag>_hbm.md
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_hbm_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C -d $OUT/pass$i -o p --output-format csv -- python3 $ROOT/tools/hbm_one.py 4 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/tools/hbm_one.py 20 > $OUT/trace.log 2>&1
cd $ROOT
python3 tools/hbm_one.py 0 alg > $OUT/alg.json 2>/dev/null
python3 - <<PY > gpurun_out/${TAG}_hbm.md
import csv, glob, json, collections, re
out = "$OUT"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
def base(n):
    m = re.match(r"_Z(\d+)", n)
    if m: return n[m.end():m.end() + int(m.group(1))]
    m = re.match(r"(?:void\s+)?([A-Za-z_][\w:]*)\s*[<(]", n)
    return m.group(1) if m else n
print("# $TAG - HBM-bound kernels of the path: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) + --kernel-trace of tools/hbm_one.py (bf16, 64^2 x 5 heads unless named)\n")
print("Counter bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies wide reads at half, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE x 1024, mean per dispatch without the first;")
print("narrow accesses are uncalibrated (the guide): read the columns as an upper bound on over-fetch, not as exact bytes.  GB/s = counter bytes / average duration.\n")
print("| kernel | dispatches | avg us | FETCH KB | WRITE KB | counter MB | GB/s (counter) | % of 8 TB/s |\n|---|---|---|---|---|---|---|---|")
rows = []
for k, d in acc.items():
    b = base(k)
    if not (b.startswith("k_") or b.startswith("gd_")): continue
    fe = d.get("FETCH_SIZE", [0]); wr = d.get("WRITE_SIZE", [0])
    fe = fe[1:] if len(fe) > 2 else fe; wr = wr[1:] if len(wr) > 2 else wr
    fe = sum(fe) / len(fe); wr = sum(wr) / len(wr)
    du = [v for n, v in dur.items() if base(n) == b and (n == k or True)]
    # several template variants of one base name: match on the full printed name first
    dd = dur.get(k) or (du[0] if du else None)
    mb = (2 * fe + wr) * 1024 / 1e6
    rows.append((b, k, dd, fe, wr, mb))
for b, k, dd, fe, wr, mb in sorted(rows, key=lambda r: r[0]):
    us = dd[1] if dd else float("nan")
    gbs = mb * 1e6 / (us * 1e-6) / 1e9 if dd else float("nan")
    print(f"| \`{k[:60]}\` | {dd[0] if dd else '?'} | {us:.1f} | {fe:,.0f} | {wr:,.0f} | {mb:.2f} | {gbs:,.0f} | {gbs / 80:.1f} |")
print("\n## algorithmic bytes per launch (SURVEY 8d; tools/hbm_one.py)\n\n| launch | MB |\n|---|---|")
for k, v in json.load(open(out + "/alg.json")).items():
    print(f"| {k} | {v / 1e6:.2f} |")
PY
cat gpurun_out/${TAG}_hbm.md
