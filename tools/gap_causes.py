"""What sits in the idle gaps of the timed region: a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py (GD_BENCH_MARK=1).
For every gap of 5 us .. 1 ms between consecutive kernels: did a memory copy (the runtime's copy engine: no kernel row) run inside it,
and which kernels surround it.  Development aid for the "eager gaps" item.
    python tools/gap_causes.py <kernel_trace.csv> [<memory_copy_trace.csv>] > out.md"""
import collections, csv, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from prof_summary import base_name

kt = sys.argv[1]
mt = sys.argv[2] if len(sys.argv) > 2 else None
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt)))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r[2]]
if len(marks) >= 2:
    rows = rows[marks[0] + 1:marks[-1]]
copies = []
if mt:
    for r in csv.DictReader(open(mt)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction") or r.get("Kind") or "?"))
    copies.sort()
t0, t1 = rows[0][0], rows[-1][1]
copies = [c for c in copies if t0 <= c[0] <= t1]
print(f"# idle gaps of the timed region by cause\n\n{len(rows)} kernels, {len(copies)} memory copies in the region, span {(t1 - t0) * 1e-6:.1f} ms\n")
import bisect
cstarts = [c[0] for c in copies]
groups = collections.defaultdict(lambda: [0, 0.0])
examples = {}
cur_end, prev = rows[0][1], base_name(rows[0][2])
hist = collections.deque(maxlen=3)
for i, (s, e, n) in enumerate(rows[1:], 1):
    b = base_name(n)
    if s > cur_end:
        g = (s - cur_end) * 1e-3
        if 5 <= g < 1000:
            lo = bisect.bisect_left(cstarts, cur_end - 2000)
            inside = [c for c in copies[lo:lo + 8] if c[0] < s and c[1] > cur_end - 2000]
            cause = "copy: " + ",".join(sorted({c[2] for c in inside})) if inside else "no copy"
            k = (prev[:44], b[:44], cause)
            groups[k][0] += 1; groups[k][1] += g
            if k not in examples:
                examples[k] = [base_name(r[2])[:28] for r in rows[max(0, i - 3):i + 3]]
        cur_end, prev = e, b
    elif e > cur_end:
        cur_end, prev = e, b
by_cause = collections.defaultdict(lambda: [0, 0.0])
for (p, n, c), (cnt, us) in groups.items():
    by_cause[c][0] += cnt; by_cause[c][1] += us
print("| cause | gaps | total ms |\n|---|---|---|")
for c, (cnt, us) in sorted(by_cause.items(), key=lambda kv: -kv[1][1]):
    print(f"| {c} | {cnt} | {us * 1e-3:.1f} |")
print("\n| after kernel | before kernel | cause | gaps | total ms | avg us | neighbourhood (3 before, 3 after) |\n|---|---|---|---|---|---|---|")
for (p, n, c), (cnt, us) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"| `{p}` | `{n}` | {c} | {cnt} | {us * 1e-3:.1f} | {us / cnt:.1f} | {' > '.join(examples[(p, n, c)])} |")
