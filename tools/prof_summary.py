"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel table of the region between the two marker kernels that
`GD_BENCH_MARK=1 bench.py` launches around its timed region (whole trace if absent), GPU busy time and the idle gaps.

    python tools/prof_summary.py <kernel_trace.csv> <out.md> <out_stats.csv> [title]
"""
import collections
import csv
import re
import sys


def base_name(n):
    """rocprofv3 prints some rows demangled (`void k_x<...>(Args)`) and some mangled (`_Z3k_xI...`): one key for both."""
    m = re.match(r"_Z(\d+)", n)
    if m:
        k = int(m.group(1))
        return n[m.end():m.end() + k]
    m = re.match(r"(?:void\s+)?([A-Za-z_][\w:]*)\s*[<(]", n)
    return m.group(1) if m else n


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
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])          # per printed name (template variants apart)
    fam = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])          # own kernels per base name (k_x / gd_x), both spellings merged
    busy, cur_end, prev_name = 0.0, t_first, "(region start)"
    gaps = collections.Counter()
    gap_after = collections.defaultdict(lambda: [0, 0.0])       # idle time by the kernel that ran BEFORE the gap (gaps of 5 us .. 1 ms)
    top_gaps = []
    self64 = collections.defaultdict(lambda: [0, 0.0])                  # 64^2 self-attention launches by (kernel, workgroups)
    for s, e, n, wgs in rows:
        b = base_name(n)
        d = (e - s) * 1e-3
        # 64^2 self-attention (N = M = 4096): k_attn_fwd_w64 launches of >= 40 us (its 32^2 x 40-head CFG launches run 27-29 us);
        # k_attn_fwd_mp launches of >= 26 us (its 32^2 launches run <= 22 us; the 77-key cross-attention launches go to k_attn_fwd)
        # (the 5-head inversion launch runs on k_attn_fwd_w64 in 240 workgroups = 80 units x 3 parts, 30-33 us)
        if (b == "k_attn_fwd_w64" and (d >= 40.0 or (wgs == 240 and d >= 26.0))) or (b == "k_attn_fwd_mp" and d >= 26.0):
            g = self64[(b, wgs)]
            g[0] += 1; g[1] += d
        for tab, key in ((agg, n), (fam, b)):
            a = tab[key]
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        if s > cur_end:
            g = (s - cur_end) * 1e-3
            gaps["<5us" if g < 5 else "5-20us" if g < 20 else "20-100us" if g < 100 else "0.1-1ms" if g < 1000 else ">1ms"] += g
            if 5 <= g < 1000:
                ga = gap_after[(prev_name, b)]
                ga[0] += 1; ga[1] += g
            if g >= 300:
                top_gaps.append((g, (cur_end - t_first) * 1e-6, prev_name, b))
            busy += (e - s) * 1e-6
            cur_end = e
            prev_name = b
        elif e > cur_end:
            busy += (e - cur_end) * 1e-6
            cur_end = e
            prev_name = b
    tot = sum(a[1] for a in agg.values()) * 1e-3
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out_csv, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
        for n, a in items:
            w.writerow([n, a[0], f"{a[1]:.1f}", f"{a[1] / a[0]:.2f}", f"{a[2]:.2f}", f"{a[3]:.2f}", f"{100 * a[1] * 1e-3 / tot:.2f}"])
    own = sorted(((n, a) for n, a in fam.items() if n.startswith("k_") or n.startswith("gd_")), key=lambda kv: -kv[1][1])
    with open(out_md, "w") as fh:
        fh.write(f"# {title}\n\nRegion: {region}.\n\n")
        fh.write(f"* span {span:.1f} ms, GPU busy {busy:.1f} ms ({100 * busy / span:.1f} %), sum of kernel durations {tot:.1f} ms over "
                 f"{len(rows)} launches\n")
        fh.write("* idle time by gap length (us): " + ", ".join(f"{k}: {v * 1e-3:.1f} ms" for k, v in sorted(gaps.items())) + "\n\n")
        fh.write("| kernel (as rocprofv3 prints it: demangled and mangled rows of one kernel are separate lines here, merged in the next table) "
                 "| calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|\n")
        for n, a in items[:32]:
            fh.write(f"| `{n[:72]}` | {a[0]} | {a[1] * 1e-3:.1f} | {a[1] / a[0]:.1f} | {100 * a[1] * 1e-3 / tot:.2f} |\n")
        fh.write("\n## own HIP kernels (libgeodiff_hip.so), all template variants of a kernel together\n\n"
                 "| kernel | calls | total ms | avg us | min us | max us |\n|---|---|---|---|---|---|\n")
        for n, a in own:
            fh.write(f"| `{n}` | {a[0]} | {a[1] * 1e-3:.1f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} |\n")
        fh.write(f"\nown kernels total {sum(a[1] for _, a in own) * 1e-3:.1f} ms of {tot:.1f} ms\n")
        # the hooked layer OUTSIDE the 64^2 self-attention forward: every kernel of the controllers except the 64^2
        # forward launches counted below and the UNet harness (conv / norm / GEMM glue)
        hooked = ("k_attn_fwd", "k_attn_fwd_mp", "k_blend", "k_blend_rows", "k_attn_bwd_dq", "k_attn_bwd_dq64", "k_removal_bwd", "k_corr_max", "k_corr_max2",
                  "k_attn_probs", "k_attn_probs2", "k_losses_fwd", "k_losses_bwd", "k_gauss5", "k_removal_rowdot", "k_attn_bwd_dk", "k_removal_reduce",
                  "k_attn_bwd_dk_reduce", "k_zero_u32", "k_removal_dq_fold", "k_losses_fold", "k_loss_assemble", "k_rows_merge",
                  "k_removal_dk", "k_amodal_interp", "k_attn_bwd_dq_fold", "k_amodal_fused", "k_losses_fused", "k_edit_dq_fold", "k_losses_bwd_rowdot",
                  "k_blend_merge", "k_attn_bwd_dq2", "k_removal_bwd2", "k_heads_split", "k_heads_merge", "k_attn_fwd_pair")
        n_edits = max(1, int(__import__("os").environ.get("GD_PROF_EDITS", "2")))
        w64_small = sum(a[1] for n, a in fam.items() if n == "k_attn_fwd_w64") - sum(v[1] for (b, _), v in self64.items() if b == "k_attn_fwd_w64")
        mp_big = sum(v[1] for (b, _), v in self64.items() if b == "k_attn_fwd_mp")
        hk = sum(a[1] for n, a in fam.items() if n in hooked) - mp_big + w64_small
        hl = sum(a[0] for n, a in fam.items() if n in hooked)
        fh.write(f"\n**hooked layer outside the 64^2 forward: {hk * 1e-3 / n_edits:.1f} ms per edit** ({hl // n_edits} launches per edit; "
                 f"the kernels of `geodiffuser_amd/attention_processors.py` except the 64^2 self-attention launches below; {n_edits} edits in the region)\n")
        big = sum(g for g, *_ in top_gaps if g >= 1000.0)
        fh.write(f"\n**idle gaps >= 1 ms: {big * 1e-3 / n_edits:.1f} ms per edit; GPU idle {100 * (1 - busy / span):.1f} % of the region**\n")
        if self64:
            cnt = sum(v[0] for v in self64.values()); us = sum(v[1] for v in self64.values())
            fh.write("\n## 64^2 self-attention launches (N = M = 4096: the launches bench.py's `roofline` is computed from)\n\n"
                     "The `k_attn_fwd_w64` launches of >= 40 us and its 240-workgroup launches of >= 26 us (5-head inversion passes: 80 units x 3 parts; "
                     "the 32^2 x 40-head launches run 27-29 us in 160 workgroups), and any `k_attn_fwd_mp` launch of >= 26 us (its 32^2 launches run <= 22 us).  Workgroups: w64 = 256-query units (or their even split), mp = 128-query units x key ranges.\n\n"
                     f"* **{cnt} launches, total {us * 1e-3:.1f} ms, average {us / max(1, cnt):.1f} us** — compare bench.py's `roofline.avg_launch_us`\n\n"
                     "| kernel | workgroups | launches | total ms | avg us |\n|---|---|---|---|---|\n")
            for (b, wgs), (c, u) in sorted(self64.items(), key=lambda kv: -kv[1][1]):
                fh.write(f"| `{b}` | {wgs} | {c} | {u * 1e-3:.1f} | {u / c:.1f} |\n")
        if gap_after:
            fh.write("\n## idle time in gaps of 5 us .. 1 ms by the kernels around the gap (top 14)\n\n| after kernel | before kernel | gaps | total ms | avg us |\n|---|---|---|---|---|\n")
            for (pn, nn), (c, u) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:14]:
                fh.write(f"| `{pn[:40]}` | `{nn[:40]}` | {c} | {u * 1e-3:.1f} | {u / c:.1f} |\n")
        if top_gaps:
            fh.write("\n## idle gaps >= 0.3 ms (host work between launches)\n\n| at ms | gap ms | after kernel | before kernel |\n|---|---|---|---|\n")
            for g, at, pn, nn in sorted(top_gaps, key=lambda x: x[1])[:80]:
                fh.write(f"| {at:.1f} | {g * 1e-3:.2f} | `{pn[:48]}` | `{nn[:48]}` |\n")


if __name__ == "__main__":
    main()
