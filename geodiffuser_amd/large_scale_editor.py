"""Batch driver over experiment folders, sharded one edit per GPU.

Mirror of ``large_scale_editor.py`` of the reference (repo root): ``perform_exp`` :196-317 (the per-edit-type configuration
table), ``save_results`` :133-176, ``run_exp_on_folder_single`` :179-192, the folder walk of ``__main__`` :320-402
(category folder -> edit type: ``Removal`` -> geometry_remover, ``Rotation_2D`` / ``Scaling`` skipped, everything else ->
geometry_editor), ``log_dictionary_to_file`` / ``save_dictionary`` :42-84 and ``generate_output_plot_and_save`` :88-130.

MI355X extension: the list of experiment folders is sharded over ranks, edit j -> rank j mod W, one process per
GPU, no collective in the data path (weights are broadcast once at start-up by ``geodiffuser_amd.dist``).

    python -m geodiffuser_amd.large_scale_editor --root <dataset root> [--exp-type geometry_editor]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m geodiffuser_amd.large_scale_editor --root ...
"""
from __future__ import annotations

import argparse
import glob
import logging
import os
import pickle
from typing import List, Optional, Tuple

import numpy as np
import torch

from .ui_utils import check_if_exp_root, complete_path, imsave, list_exp_details, read_exp

log = logging.getLogger("geodiffuser_amd.large_scale")

# perform_exp :196-317 — what each edit type overrides on top of the signature defaults
_DEFAULTS = dict(cross_replace_steps={"default_": 0.9}, self_replace_steps=0.9, optimize_steps=0.85, lr=0.03, latent_replace=0.4,
                 optimize_embeddings=True, optimize_latents=True, obj_edit_step=1.0, perform_inversion=False, skip_optim_steps=2,
                 guidance_scale=5.0, num_ddim_steps=50, splatting_tau=1.0, splatting_points_per_pixel=15, splatting_radius=1.3,
                 loss_weights_dict=None)
EDIT_CONFIGS = {
    "geometry_remover": dict(guidance_scale=5.0,
                             loss_weights_dict={"self": {"sim": 55, "removal": 4.6, "smoothness": 30.0},
                                                "cross": {"sim": 45, "removal": 4.6, "smoothness": 15.0}}),
    "geometry_editor": dict(loss_weights_dict={"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30.0, "amodal": 80.5},
                                               "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15.0, "amodal": 3.5}},
                            splatting_radius=1.3, splatting_tau=1.0, optimize_steps=0.65, latent_replace=0.1,
                            splatting_points_per_pixel=15, guidance_scale=3.0, cross_replace_steps={"default_": 0.95},
                            self_replace_steps=0.95, obj_edit_step=0.9),
}


def edit_config(edit_type: str, **overrides) -> dict:
    """Keyword arguments ``perform_exp`` passes to ``perform_geometric_edit`` for ``edit_type``."""
    import copy
    if edit_type not in EDIT_CONFIGS:
        # geometry_stitch* need controller classes that do not exist in the reference either (SURVEY.md F-list)
        raise NameError(f"edit_type {edit_type!r}: no controller for it (reference: NameError at editor.py:618-621)")
    kw = copy.deepcopy(_DEFAULTS)
    kw.update(overrides)
    kw.update(copy.deepcopy(EDIT_CONFIGS[edit_type]))
    return kw


def perform_exp(exp_dict, prompt="", ldm_stable_model=None, tokenizer_model=None, scheduler_in=None, edit_type="geometry_editor",
                **overrides):
    """:196-317 -> (images, loss_dict, step_store=None)."""
    from .editor import perform_geometric_edit
    kw = edit_config(edit_type, **overrides)
    image = exp_dict["input_image_png"]
    image_mask = (exp_dict["input_mask_png"] / 255.0)[..., 0]
    depth = exp_dict["depth_npy"]
    transform_in = torch.tensor(exp_dict["transform_npy"]).float()
    images, loss_dict = perform_geometric_edit(
        image, depth, image_mask, transform_in, prompt, ldm_stable_model=ldm_stable_model, tokenizer_model=tokenizer_model,
        scheduler_in=scheduler_in, edit_type=edit_type, image_stitch=None, fast_start_steps=0.0, num_first_optim_steps=1,
        return_loss_log_dict=True, use_adaptive_optimization=True, return_attention_maps=False, **kw)
    return images, loss_dict, None


def save_dictionary(d, pkl_path):
    with open(pkl_path, "wb") as fh:
        pickle.dump(d, fh)


def load_dictionary(pkl_path):
    with open(pkl_path, "rb") as fh:
        return pickle.load(fh)


def log_dictionary_to_file(log_dict, file_path):
    """:53-84 — one block of DEBUG lines per optimisation step (own file handler instead of re-pointing the root logger)."""
    with open(file_path, "w") as fh:
        for i in log_dict:
            log_dict[i]["optimization_step"] = i
            fh.write("optimization_step: %d\n" % i)
            fh.write("logged_dict_self: %s\n" % (log_dict[i]["self"],))
            fh.write("logged_dict_cross: %s\n" % (log_dict[i]["cross"],))
            fh.write("num_layers: %d\n" % log_dict[i]["num_layers"])


def resize_image(image: np.ndarray, aspect_ratio) -> np.ndarray:
    """image_processing.py:100-113 (``cv2.resize`` bilinear to the original aspect ratio; cv2 is absent: PARITY UNPINNED,
    half-pixel-centre bilinear resampling via torch)."""
    h, w = image.shape[:2]
    ratio = aspect_ratio[1] / aspect_ratio[0]
    new_h, new_w = (h / ratio, w) if ratio < 1 else (h, ratio * w)
    new_h, new_w = int(new_h), int(new_w)
    if (new_h, new_w) == (h, w):
        return image.copy()
    t = torch.from_numpy(np.ascontiguousarray(image)).float()
    t = t[None, None] if t.dim() == 2 else t.permute(2, 0, 1)[None]
    t = torch.nn.functional.interpolate(t, size=(new_h, new_w), mode="bilinear", align_corners=False)
    out = t[0, 0] if image.ndim == 2 else t[0].permute(1, 2, 0)
    if image.dtype == np.uint8:
        return out.round().clamp(0, 255).to(torch.uint8).numpy()
    return out.numpy().astype(image.dtype)


def generate_output_plot_and_save(exp_dict):
    """:88-130 — input | [target] | result side by side, 20 px apart -> experiment.png."""
    ims = [exp_dict["input_image_png"]]
    if exp_dict.get("transformed_image_png") is not None:
        ims.append(exp_dict["transformed_image_png"])
    ims.append(exp_dict["final_result"])
    h, w = ims[0].shape[:2]
    gap = 20
    canvas = np.zeros((h, len(ims) * w + gap * (len(ims) - 1), ims[0].shape[-1]))
    x = 0
    for im in ims:
        canvas[:, x:x + w] = im[:h, :w] if im.shape[:2] != (h, w) else im
        x += w + gap
    imsave(exp_dict["path_name"] + "experiment.png", canvas.astype("uint8"))


def save_results(exp_dict, out_image, loss_dict, edit_type="geometry_editor", step_store=None):
    """:133-176 — loss.log, loss.pkl, [attention.pkl], result_ls.png, resized_result_ls.png, resized_<key>.png, experiment.png."""
    from scipy.ndimage import maximum_filter
    path = exp_dict["path_name"]
    log_dictionary_to_file(loss_dict, path + "loss.log")
    save_dictionary(loss_dict, path + "loss.pkl")
    if step_store is not None:
        for k in step_store:
            if isinstance(step_store[k], list):
                step_store[k] = [a.detach().cpu().numpy() for a in step_store[k]]
        save_dictionary(step_store, path + "attention.pkl")
    aspect = exp_dict["image_shape_npy"]
    out_image = np.asarray(out_image)
    if out_image.dtype != np.uint8:                        # the edit result is the float64 histogram-matched image
        out_image = np.clip(out_image, 0, 255).astype(np.uint8)
    out_resized = resize_image(out_image, aspect)
    imsave(path + "result_ls.png", out_image)
    imsave(path + "resized_result_ls.png", out_resized)
    if edit_type == "geometry_remover":                    # :152-158 highlight of the removed region as the "target" panel
        mask = maximum_filter(exp_dict["input_mask_png"][..., :1] / 255.0, 20)
        t = exp_dict["input_image_png"]
        exp_dict["transformed_image_png"] = (t * (1.0 - mask) + mask * (0.5 * 255 + 0.5 * t)).astype("uint8")
    for k in list(exp_dict):
        if k.split("_")[-1] == "png" and exp_dict[k] is not None:
            im = resize_image(exp_dict[k], aspect)
            imsave(path + "resized_" + k + ".png", im, gray=k in ("input_mask_png", "depth_png"))
            exp_dict[k] = im
    exp_dict["final_result"] = out_resized
    generate_output_plot_and_save(exp_dict)


class FolderIO:
    """Read-ahead / write-behind for a rank's work list.  One experiment's files cost ~0.1 s to read and ~1 s to write on the host
    (seven PNGs through zlib, :133-176) — more than the edit itself takes on an MI355X — and the reference does both in line with the
    edit.  Here ``read_exp`` of the NEXT group and ``save_results`` of finished experiments run on worker threads (PNG decode / encode
    and the numpy resizes release the GIL) while the device runs the current group.  Same files, same bytes; at most ``max_pending``
    unfinished writes (the oldest is waited for beyond that); an exception in a worker is raised by the next call or by ``close()``.
    ``threads = 0``: everything in line, as in the reference."""

    def __init__(self, threads: int = 8, max_pending: int = 32):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="gd-folder-io") if threads > 0 else None
        self.max_pending = max(1, max_pending)
        self.pending = []

    def read(self, folders):
        """-> a handle whose ``result()`` is [read_exp(folder) for folder in folders] (started now)."""
        folders = [complete_path(f) for f in folders]
        if self.pool is None:
            return _Done([read_exp(f) for f in folders])
        futs = [self.pool.submit(read_exp, f) for f in folders]
        return _Gather(futs)

    def save(self, exp_dict, out_image, loss_dict, edit_type, step_store=None):
        if self.pool is None:
            return save_results(exp_dict, out_image, loss_dict, edit_type, step_store=step_store)
        self._reap(block_over=self.max_pending - 1)
        self.pending.append(self.pool.submit(save_results, exp_dict, out_image, loss_dict, edit_type, step_store=step_store))

    def _reap(self, block_over=None):
        while self.pending and (self.pending[0].done() or (block_over is not None and len(self.pending) > block_over)):
            self.pending.pop(0).result()                       # raises what the worker raised

    def close(self):
        try:
            while self.pending:
                self.pending.pop(0).result()
        finally:
            if self.pool is not None:
                self.pool.shutdown(wait=True)
                self.pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class _Done:
    def __init__(self, v):
        self.v = v

    def result(self):
        return self.v


class _Gather:
    def __init__(self, futs):
        self.futs = futs

    def result(self):
        return [f.result() for f in self.futs]


def run_exp_on_folder_single(exp_folder, exp_type, ldm_stable, tokenizer, scheduler, *, exp_dict=None, io: Optional[FolderIO] = None,
                             **overrides):
    """:179-192.  ``exp_dict``: the folder's files if already read (FolderIO.read); ``io``: write the result files behind the next edit."""
    log.info("Performing edit on: %s with exp type: %s", exp_folder, exp_type)
    exp_folder = complete_path(exp_folder)
    if exp_dict is None:
        exp_dict = read_exp(exp_folder)
    list_exp_details(exp_dict, printer=lambda *a: log.debug(" ".join(str(x) for x in a)))
    images, loss_dict, step_store = perform_exp(exp_dict, edit_type=exp_type, ldm_stable_model=ldm_stable, tokenizer_model=tokenizer,
                                                scheduler_in=scheduler, **overrides)
    (io.save if io is not None else save_results)(exp_dict, images[-1], loss_dict, exp_type, step_store=step_store)
    return images


def run_exp_on_folders_batched(exp_folders, exp_type, ldm_stable, tokenizer, scheduler, *, exp_dicts=None, io: Optional[FolderIO] = None,
                               **overrides):
    """``run_exp_on_folder_single`` for a group of folders of ONE edit type whose edits share every UNet pass
    (geodiffuser_amd.batch.perform_geometric_edit_batch): same inputs, same configuration column, same result files per folder."""
    from .batch import perform_geometric_edit_batch
    kw = edit_config(exp_type, **overrides)
    dicts, edits = [], []
    for j, f in enumerate(exp_folders):
        log.info("Performing edit on: %s with exp type: %s (batch of %d)", f, exp_type, len(exp_folders))
        d = exp_dicts[j] if exp_dicts is not None else read_exp(complete_path(f))
        dicts.append(d)
        edits.append(dict(image=d["input_image_png"], image_mask=(d["input_mask_png"] / 255.0)[..., 0], depth=d["depth_npy"],
                          transform_in=torch.tensor(d["transform_npy"]).float(), prompt=""))
    # a batch shares one image size: folders of other sizes in this group run as batches of their own (a dataset is normally uniform)
    by_size = {}
    for j, e in enumerate(edits):
        by_size.setdefault(tuple(np.asarray(e["image"]).shape), []).append(j)
    res = [None] * len(edits)
    for idx in by_size.values():
        part = perform_geometric_edit_batch([edits[j] for j in idx], ldm_stable_model=ldm_stable, tokenizer_model=tokenizer, scheduler_in=scheduler,
                                            edit_type=exp_type, return_loss_log_dict=True, **kw)
        for j, r in zip(idx, part):
            res[j] = r
    for d, (images, loss_dict) in zip(dicts, res):
        (io.save if io is not None else save_results)(d, images[-1], loss_dict, exp_type, step_store=None)
    return [r[0] for r in res]


def run_work_list(work: List[Tuple[str, str]], ldm_stable, tokenizer, scheduler, edits_per_pass: int = 1, io_threads: int = 8, **overrides):
    """A rank's whole work list: groups of ``edits_per_pass`` folders per pass (1: the reference's one edit at a time, in work-list
    order), the next group's files read ahead and the result files written behind by ``io_threads`` workers (FolderIO)."""
    groups = group_for_batches(work, edits_per_pass) if edits_per_pass > 1 else [([f], t) for f, t in work]
    with FolderIO(io_threads) as io:
        ahead = io.read(groups[0][0]) if groups else None
        for i, (folders, etype) in enumerate(groups):
            dicts = ahead.result()
            ahead = io.read(groups[i + 1][0]) if i + 1 < len(groups) else None
            if len(folders) == 1:
                run_exp_on_folder_single(folders[0], etype, ldm_stable, tokenizer, scheduler, exp_dict=dicts[0], io=io, **overrides)
            else:
                run_exp_on_folders_batched(folders, etype, ldm_stable, tokenizer, scheduler, exp_dicts=dicts, io=io, **overrides)
            for f in folders:
                log.info("Completed: %s", f)           # the edit; its files are on disk when this function returns


def group_for_batches(work: List[Tuple[str, str]], per_pass: int) -> List[Tuple[List[str], str]]:
    """A rank's work list -> groups of at most ``per_pass`` folders of one edit type, in work-list order within a type."""
    by_type = {}
    for folder, etype in work:
        by_type.setdefault(etype, []).append(folder)
    return [(fs[i:i + per_pass], etype) for etype, fs in by_type.items() for i in range(0, len(fs), per_pass)]


def list_experiments(exp_root_folder: str, exp_type: Optional[str] = None) -> List[Tuple[str, str]]:
    """The folder walk of ``__main__`` :358-402 as a flat, sorted work list of (experiment folder, edit type)."""
    folder_list = sorted(glob.glob(complete_path(exp_root_folder) + "**/"))
    work = []
    if check_if_exp_root(exp_root_folder, folder_list):
        for f in folder_list:
            cat = f.split("/")[-2]
            if cat in ("Rotation_2D", "Scaling"):          # skipped by the reference driver (:378-381)
                continue
            etype = "geometry_remover" if cat == "Removal" else "geometry_editor"
            for e in sorted(glob.glob(complete_path(f) + "**/")):
                work.append((e, etype))
    else:
        if exp_type is None:
            raise ValueError("not a category root: --exp-type is required (the reference relies on a module-level exp_type here)")
        work = [(e, exp_type) for e in folder_list]
    return work


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--root", required=True, help="dataset root (category folders) or a folder of experiment folders")
    ap.add_argument("--exp-type", default=None, choices=[None, "geometry_editor", "geometry_remover"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--limit", type=int, default=0, help="process at most this many experiments in total (0 = all)")
    ap.add_argument("--gpus", type=int, default=0, help="start this many GPUs' worth of ranks here (0: run as the one process / rank this is)")
    ap.add_argument("--edits-in-flight", type=int, default=1,
                    help="with --gpus: independent edits in flight per GPU = ranks per device (4 gives 1.85 x the edits/min of 1 on an MI355X)")
    ap.add_argument("--edits-per-pass", type=int, default=1,
                    help="B > 1: every rank runs its folders B at a time in ONE process, sharing every UNet pass (geodiffuser_amd/batch.py; one "
                         "copy of the weights; 4 gives 1.8 x, 8 gives 2.1 x, 16 gives 2.2 x the edits/min of 1 on an MI355X)")
    ap.add_argument("--io-threads", type=int, default=8,
                    help="worker threads that read the next experiments' files and write finished results while the GPU edits (0: in line)")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    from . import dist as gdist
    if args.gpus > 0 and "WORLD_SIZE" not in os.environ:
        # our own launcher (before anything touches the GPU): gpus x edits-in-flight ranks of this module, experiments sharded j mod W
        import sys
        tail = [a for a in (sys.argv[1:] if argv is None else list(argv))]
        for flag in ("--gpus", "--edits-in-flight"):
            while flag in tail:
                i = tail.index(flag)
                del tail[i:i + 2]
        raise SystemExit(gdist.launch_ranks(args.gpus, max(1, args.edits_in_flight), ["-m", "geodiffuser_amd.large_scale_editor"], tail,
                                            run=getattr(main, "_run", None)))
    from . import editor
    from .diffusion import load_model
    from . import miopen_cache
    miopen_cache.configure()                                                  # committed find-db: no per-rank solver search
    rank, world, local = gdist.init()
    local = gdist.local_device_index(local)              # GD_EDITS_IN_FLIGHT = P: P ranks (independent edits) share every GPU
    # (the edit itself needs the HIP extension and fails loudly without a GPU; the driver around it — sharding, broadcast, result
    #  files — is device-agnostic so that the world-size-2 gloo test can run it on CPU with a stubbed edit)
    dev = f"cuda:{local}" if torch.cuda.is_available() else "cpu"
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    editor.DEVICE = torch.device(dev)
    pipe, tok, sched = load_model(device=dev, dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float16)
    gdist.broadcast_model([pipe.unet, pipe.vae, pipe.text_encoder], src=0)     # one RCCL broadcast; no-op for one rank
    work = list_experiments(args.root, args.exp_type)
    if args.limit:
        work = work[:args.limit]
    mine = gdist.shard(work, rank, world)
    log.info("rank %d/%d: %d of %d experiments", rank, world, len(mine), len(work))
    run_work_list(mine, pipe, tok, sched, edits_per_pass=max(1, args.edits_per_pass), io_threads=max(0, args.io_threads))
    gdist.barrier()


if __name__ == "__main__":
    main()
