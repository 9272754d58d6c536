"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel table of the region between the two marker kernels that
`GD_BENCH_MARK=1 bench.py` launches around its timed region (whole trace if absent), GPU busy time and the idle gaps.

    python tools/prof_summary.py <kernel_trace.csv> <out.md> <out_stats.csv> [title]
"""
import collections
import csv
import sys


def main():
    trace, out_md, out_csv = sys.argv[1:4]
    title = sys.argv[4] if len(sys.argv) > 4 else trace
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) // max(1, int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 256))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "spin_kernel" in r[2]]
    if len(marks) >= 2:
        rows = rows[marks[0] + 1:marks[-1]]
        region = "between the bench's two marker kernels (the timed region)"
    else:
        region = "whole trace (no markers)"
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    span = (t_last - t_first) * 1e-6
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    busy, cur_end = 0.0, t_first
    gaps = collections.Counter()
    by_grid = collections.defaultdict(lambda: [0, 0.0])
    big = [0, 0.0]                      # 64^2 self-attention launches: >= 320 workgroups AND >= 35 us (77-key cross attention at 64^2 has
    #                                     the same workgroup count but runs ~8 us, 32^2 self-attention with 40 heads ~29 us)
    for s, e, n, wgs in rows:
        if "k_attn_fwd" in n:
            g = by_grid[wgs]
            g[0] += 1; g[1] += (e - s) * 1e-3
            # r02: the pipelined kernels serve every self-attention launch; 64^2 launches run >= 28 us (5 heads), 32^2 ones <= 22 us
            if ("k_attn_fwd_mp" in n and (e - s) >= 26000) or ("k_attn_fwd_mp" not in n and wgs >= 320 and (e - s) >= 35000):
                big[0] += 1; big[1] += (e - s) * 1e-3
        a = agg[n]
        d = (e - s) * 1e-3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        if s > cur_end:
            g = (s - cur_end) * 1e-3
            gaps["<5us" if g < 5 else "5-20us" if g < 20 else "20-100us" if g < 100 else "0.1-1ms" if g < 1000 else ">1ms"] += g
            busy += (e - s) * 1e-6
            cur_end = e
        elif e > cur_end:
            busy += (e - cur_end) * 1e-6
            cur_end = e
    tot = sum(a[1] for a in agg.values()) * 1e-3
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out_csv, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
        for n, a in items:
            w.writerow([n, a[0], f"{a[1]:.1f}", f"{a[1] / a[0]:.2f}", f"{a[2]:.2f}", f"{a[3]:.2f}", f"{100 * a[1] * 1e-3 / tot:.2f}"])
    own = [(n, a) for n, a in items if n.startswith("_Z") and ("k_" in n[:8] or "gd_" in n[:10])]
    with open(out_md, "w") as fh:
        fh.write(f"# {title}\n\nRegion: {region}.\n\n")
        fh.write(f"* span {span:.1f} ms, GPU busy {busy:.1f} ms ({100 * busy / span:.1f} %), sum of kernel durations {tot:.1f} ms over "
                 f"{len(rows)} launches\n")
        fh.write("* idle time by gap length (us): " + ", ".join(f"{k}: {v * 1e-3:.1f} ms" for k, v in sorted(gaps.items())) + "\n\n")
        fh.write("| kernel | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|\n")
        for n, a in items[:32]:
            fh.write(f"| `{n[:72]}` | {a[0]} | {a[1] * 1e-3:.1f} | {a[1] / a[0]:.1f} | {100 * a[1] * 1e-3 / tot:.2f} |\n")
        fh.write("\n## own HIP kernels (libgeodiff_hip.so)\n\n| kernel | calls | total ms | avg us | min us | max us |\n|---|---|---|---|---|---|\n")
        for n, a in own:
            fh.write(f"| `{n[:72]}` | {a[0]} | {a[1] * 1e-3:.1f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} |\n")
        fh.write(f"\nown kernels total {sum(a[1] for _, a in own) * 1e-3:.1f} ms of {tot:.1f} ms\n")
        if by_grid:
            fh.write("\n## k_attn_fwd by launch size (workgroups = 128-query tiles x heads x key splits)\n\n"
                     "bench.py's `roofline` is computed from the 64^2 SELF-attention launches (N = M = 4096): 32 query tiles x 5-20 heads x 1-4 key "
                     "splits = 320-640 workgroups.  64^2 CROSS attention (77 keys) has the same workgroup counts but runs ~8 us, so the classes "
                     "below mix the two (and the 320 class also holds the 40-head 32^2 launches, ~29 us); the launches of those classes that take >= 35 us are the 64^2 self-attention ones:\n\n"
                     f"* **64^2 self-attention launches: {big[0]}, total {big[1] * 1e-3:.1f} ms, average {big[1] / max(1, big[0]):.1f} us** "
                     "(bench.py's `avg_launch_us` additionally contains the ~6 us split-KV merge kernel of the inversion-pass launches)\n\n"
                     "| workgroups | launches | total ms | avg us |\n|---|---|---|---|\n")
            for wgs, (cnt, us) in sorted(by_grid.items(), key=lambda kv: -kv[1][1])[:16]:
                fh.write(f"| {wgs} | {cnt} | {us * 1e-3:.1f} | {us / cnt:.1f} |\n")


if __name__ == "__main__":
    main()
