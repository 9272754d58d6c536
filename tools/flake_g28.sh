#!/bin/bash
# Spread of the loop tests' last-pass loss terms over repeated runs (the convolutions of the harness are not run-to-run deterministic):
# how close the G28 / G29 loop tests come to their bounds.  Usage (GPU box): tools/flake_g28.sh [runs]
N=${1:-6}
for t in "cfg1_t50 and bf16" "cfg1_t50 and fp16" "rem768_t75 and bf16" "rem768_t75 and fp16"; do
  for i in $(seq $N); do python -m pytest tests/test_end_to_end.py -q -m gpu -s -k "g18 and $t" 2>&1 | grep -E "last pass|edit-latent|^E  .*Assertion|passed|failed" | sed "s/^/$t | /"; done
done > gpurun_out/flake_g28.log 2>&1
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list); res = collections.Counter(); fin = collections.defaultdict(list)
for l in open("gpurun_out/flake_g28.log"):
    t, _, rest = l.partition(" | ")
    m = re.search(r"last pass \(step (\d+)\) (\w+)/(\w+): ([-\d.e]+) vs ([-\d.e]+)", rest)
    if m:
        v, r = float(m.group(4)), float(m.group(5)); d[(t, m.group(2) + "/" + m.group(3))].append((abs(v - r), abs(r)))
    m = re.search(r"edit-latent rel_l2 vs the reference driver: ([\d.]+) \(ideal \w+ storage: ([\d.]+)", rest)
    if m: fin[t].append((float(m.group(1)), float(m.group(2))))
    if "passed" in rest or "failed" in rest: res[(t, "failed" if "failed" in rest else "passed")] += 1
    if "Assertion" in rest: print(l.strip()[:200])
print(dict(res))
for t, v in fin.items(): print(f"{t:24s} edit-latent distance {min(a for a, _ in v):.4f} .. {max(a for a, _ in v):.4f} (ideal storage {v[0][1]:.4f}), {len(v)} runs")
print("| test | term | reference | max abs dev | max rel dev |"); print("|---|---|---|---|---|")
for (t, k), v in sorted(d.items()):
    print(f"| {t} | {k} | {v[0][1]:.5f} | {max(a for a, _ in v):.2e} | {max(a / max(r, 1e-9) for a, r in v):.3f} |")
PY
