"""End-to-end experiment folders per minute of the batch driver (geodiffuser_amd.large_scale_editor.run_work_list) on one GPU: the edit
PLUS reading every folder's files and writing its result files (seven PNGs, loss.log, loss.pkl), which bench.py leaves out by contract
(inputs resident when the timed region starts).

    python tools/folder_driver_bench.py [--folders 16] [--warmup 8]

Writes N synthetic experiment folders (geodiffuser_amd.synthetic.make_edit through ui_utils.save_exp: the reference's folder format)
under a temporary directory, runs `--warmup` of them untimed (graph captures), then times the whole work list for every combination of
--edits-per-pass in {1, 4, 8} and --io-threads in {0, 8}.  One line per combination.
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--folders", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--per-pass", type=int, nargs="*", default=[1, 4, 8])
    ap.add_argument("--io-threads", type=int, nargs="*", default=[0, 8])
    args = ap.parse_args()
    from geodiffuser_amd import editor, large_scale_editor as L, miopen_cache
    from geodiffuser_amd.diffusion import load_model
    from geodiffuser_amd.synthetic import make_edit
    from geodiffuser_amd.ui_utils import save_exp
    miopen_cache.configure()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    editor.DEVICE = torch.device(dev)
    pipe, tok, sched = load_model(device=dev, dtype=torch.bfloat16)
    root = tempfile.mkdtemp(prefix="gd_folders_")
    try:
        for j in range(args.folders):
            image, depth, mask, T = make_edit(j, size=512, kind="rotate")
            dv = np.stack([(depth / depth.max() * 255).astype(np.uint8)] * 3, -1)
            m = np.stack([(mask * 255).astype(np.uint8)] * 3, -1)
            save_exp(root, image, depth, dv, m, T.numpy(), transformed_image=image, h=512, w=512, exp_transform_type="Rotation_3D")
        work = L.list_experiments(root)
        assert len(work) == args.folders, (len(work), args.folders)
        print(f"# {len(work)} folders of 512^2 edits (geometry_editor column), bf16, one MI355X; host cores {os.cpu_count()}", flush=True)
        for per_pass in args.per_pass:
            L.run_work_list(work[:max(per_pass, min(args.warmup, len(work)))], pipe, tok, sched, edits_per_pass=per_pass, io_threads=0)   # captures
            for threads in args.io_threads:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                L.run_work_list(work, pipe, tok, sched, edits_per_pass=per_pass, io_threads=threads)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(f"edits-per-pass {per_pass}  io-threads {threads}: {dt / len(work) * 1e3:7.1f} ms per folder = {60 * len(work) / dt:6.1f} folders/min",
                      flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
