"""Which part of the device path supplies the loop-level distance from the reference's fp32 driver?  

Re-runs the device loop of a long fixture (default G28: BASELINE configs[1] at its stated length, narrow model; also G29 / G30) with ONE
switch flipped at a time and prints the edit-latent distance to the REFERENCE's recorded fp32 latents for each, next to the yardsticks
of tests/golden/fp16_emulation.json (ideal 16-bit storage of module outputs; every aten op's result rounded; + 16-bit probabilities).
Every variant is run `--reps` times: the harness convolutions are not bit-reproducible, so a switch only counts when it moves the
distance by more than the spread of the baseline.

    python tools/loop_error_budget.py [--kinds cfg1_t50 rem768_t75] [--dtypes fp16 bf16] [--reps 3] [--out profiles/r05_loop_error_budget.md]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

from _loop import LOOP_KINDS, run_device_loop  # noqa: E402
from _util import rel_l2  # noqa: E402


def switches():
    """name -> (what it replaces, [(module, attribute, value)])."""
    from geodiffuser_amd import attention_processors as ap, graphs, unet_sd21 as un
    return {
        "baseline": ("the product path as benchmarked", []),
        "unscaled_q": ("queries WITHOUT the folded scale: q = 16-bit(xW), scores scaled inside the kernels (Q' = 16-bit(c q) off)",
                       [(ap, "SCALED_Q", False), (ap, "SCALED_Q_OPT", False)]),
        "opt_rescue": ("optimisation pass on the exact-scale rescue forward instead of the pre-scaled LSUM variant", [(ap, "OPT_PRE", False)]),
        "no_row_lists": ("warped edit attention over all rows (no row list + merge)", [(ap, "WARP_ROWS", False)]),
        "unfused_layer": ("stand-alone loss / probability / fold launches (GD_FUSED_LAYER=0, GD_TAIL_SUMS=0, GD_TOK_OPT=0)",
                          [(ap, "FUSED_LAYER", False), (ap, "TAIL_SUMS", False), (ap, "TOK_OPT", False)]),
        "head_major": ("no token-major passes / batched QKV / pair-blend launches (the reference's head_to_batch_dim layout everywhere)",
                       [(ap, "TOKEN_MAJOR", False), (ap, "BATCHED_QKV", False), (ap, "PAIR_BLEND", False)]),
        "miopen_convs": ("MIOpen convolutions instead of k_conv3x3 / the 1x1-as-linear route", [(un, "CONV3X3", False), (un, "CONV1X1", False)]),
        "stock_unet_ops": ("torch GroupNorm / LayerNorm / GEGLU / bias-residual instead of the fused glue kernels", [(un, "FUSED", False)]),
        "eager": ("no hipGraphs (same kernels, eager dispatch)", [(graphs, "ENABLED", False)]),
        "everything_stock": ("ALL of the above at once: torch / MIOpen for every non-attention op, head-major unfused hooked layers, eager",
                             [(ap, "SCALED_Q", False), (ap, "SCALED_Q_OPT", False), (ap, "OPT_PRE", False), (ap, "WARP_ROWS", False),
                              (ap, "FUSED_LAYER", False), (ap, "TAIL_SUMS", False), (ap, "TOK_OPT", False), (ap, "TOKEN_MAJOR", False),
                              (ap, "BATCHED_QKV", False), (ap, "PAIR_BLEND", False), (un, "CONV3X3", False), (un, "CONV1X1", False),
                              (un, "FUSED", False), (graphs, "ENABLED", False)]),
    }


def main():
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--kinds", nargs="+", default=["cfg1_t50"])
    ap_.add_argument("--dtypes", nargs="+", default=["fp16", "bf16"])
    ap_.add_argument("--reps", type=int, default=3)
    ap_.add_argument("--only", nargs="*", default=None)
    ap_.add_argument("--out", default=None)
    args = ap_.parse_args()
    sw = switches()
    names = [n for n in sw if args.only is None or n in args.only or n == "baseline"]
    emu_all = json.load(open(os.path.join(ROOT, "tests", "golden", "fp16_emulation.json")))
    lines = []

    def emit(s=""):
        print(s, flush=True)
        lines.append(s)

    for kind in args.kinds:
        fixture = LOOP_KINDS[kind][0]
        emu = emu_all.get(fixture, {})
        for dn in args.dtypes:
            dtype = torch.float16 if dn == "fp16" else torch.bfloat16
            emit(f"\n### {fixture}, {dn}: edit-latent rel-L2 vs the reference driver's fp32 latents ({args.reps} runs per row, 3-row CFG batch)\n")
            emit("yardsticks (CPU, reference driver re-run): " + ", ".join(
                f"{k.replace('emulated_' + dn, 'ideal ' + dn + ' storage')} {emu[k]:.4%}" for k in sorted(emu)
                if k.startswith("emulated_" + dn) and isinstance(emu[k], float) and "first_update" not in k)
                + (f", fp32 under another thread partition {emu['fp32_other_partition']:.4%}" if "fp32_other_partition" in emu else ""))
            emit("\n| variant | what changed | final latent (mean, min-max) | first update (mean) | s per run |")
            emit("|---|---|---|---|---|")
            base_mean = None
            for n in names:
                what, sets = sw[n]
                saved = [(m, a, getattr(m, a)) for m, a, _ in sets]
                for m, a, v in sets:
                    setattr(m, a, v)
                try:
                    fin, upd, secs = [], [], []
                    for _ in range(args.reps):
                        t0 = time.perf_counter()
                        g, _, runs = run_device_loop(kind, dtype, skip_refs=("ahead",))
                        torch.cuda.synchronize()
                        secs.append(time.perf_counter() - t0)
                        lat, log, w_rm, first_update, w_traj = runs[0][:5]
                        fin.append(rel_l2(lat[1], torch.from_numpy(g["latents"])[1]))
                        upd.append(rel_l2(first_update, torch.from_numpy(g["first_update"])))
                except Exception as e:  # noqa: BLE001
                    emit(f"| {n} | {what} | FAILED: {e!r} | | |")
                    continue
                finally:
                    for m, a, v in saved:
                        setattr(m, a, v)
                mean = sum(fin) / len(fin)
                if n == "baseline":
                    base_mean = mean
                emit(f"| {n} | {what} | {mean:.4%} ({min(fin):.4%} - {max(fin):.4%})"
                     + (f" = {mean / base_mean:.2f} x baseline" if base_mean and n != "baseline" else "")
                     + f" | {sum(upd) / len(upd):.4%} | {sum(secs) / len(secs):.1f} |")
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
