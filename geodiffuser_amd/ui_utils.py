"""Experiment-folder wire format and transform composition (the callers either side of the hot path).

Mirror of the parts of GeoDiffuser/utils/ui_utils.py that the batch driver needs — same names, argument meaning and file
layout, so the authors' dataset folders are consumed unchanged:

    <root>/<Category>/<n>/input_image.png  input_mask.png  depth.npy  transform.npy  image_shape.npy
                          [background_image.png  depth.png  transformed_image.png  result.png ...]

``read_exp`` / ``save_exp`` :52-159, ``read_image`` :39-50, ``list_exp_details`` :173-181, ``get_exp_types`` /
``check_if_exp_root`` :901-920, and the transform composition of ``get_transformed_mask`` :529-555 (``compose_transform``).
The gradio UI, SAM / depth predictors and camera capture in the same reference file are out of scope.  PNG files are read /
written with Pillow; the reference goes through ``matplotlib.pyplot.imread / imsave`` (float32 round trip), whose arithmetic
``read_image`` reproduces.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Optional

import numpy as np
import torch
from PIL import Image

from . import vis_utils


def count_folders(directory):
    return len([name for name in os.listdir(directory) if os.path.isdir(os.path.join(directory, name))])


def create_folder(directory):
    os.makedirs(directory, exist_ok=True)


def complete_path(directory):
    return os.path.join(directory, "")


def file_exists(f_path):
    return os.path.exists(f_path)


def read_image(im_path) -> np.ndarray:
    """:39-50.  ``plt.imread`` yields float32 ``v / 255`` for 8-bit PNGs (uint8 for other formats); the reference then drops alpha
    and, if the maximum is <= 1.0, converts back with ``(im * 255.0).astype("uint8")`` — a truncation, so levels whose float32
    round trip lands just below the integer come back one lower.  Reproduced exactly (fixture G15)."""
    with Image.open(im_path) as pil:
        is_png = (pil.format == "PNG")
        if pil.mode in ("P", "1", "PA") or "transparency" in pil.info:
            pil = pil.convert("RGBA")
        elif pil.mode == "LA":
            pil = pil.convert("RGBA")
        elif pil.mode.startswith("I;16") or pil.mode == "I":
            arr16 = np.asarray(pil).astype(np.float32) / 65535.0 if is_png else np.asarray(pil)
            pil = None
        if pil is not None:
            raw = np.asarray(pil)
            im = np.divide(raw, 255, dtype=np.float32) if is_png else raw
        else:
            im = arr16
    if im.ndim == 3:
        im = im[..., :3]
    if im.max() <= 1.0:
        im = (im * 255.0).astype("uint8")
    return im


def _to_rgba8(arr: np.ndarray, gray: bool) -> np.ndarray:
    """What ``plt.imsave`` writes: RGB(A) uint8 as is (alpha 255); RGB float in [0,1] scaled; a 2-D array min/max-normalised
    through the 256-level gray colormap."""
    arr = np.asarray(arr)
    if arr.ndim == 2:                              # imsave(..., cmap="gray") on a 2-D array
        a = arr.astype(np.float64)
        lo, hi = float(a.min()), float(a.max())
        n = (a - lo) / (hi - lo) if hi > lo else np.zeros_like(a)
        idx = np.clip((n * 256).astype(np.int64), 0, 255)
        # matplotlib turns its float gray table into bytes by truncation: level i is written as trunc(i/255 * 255), which is
        # i - 1 for the i whose float round trip lands just below the integer (33, 37, 41, ...)
        g = (np.linspace(0.0, 1.0, 256) * 255).astype(np.uint8)[idx]
        return np.stack([g, g, g, np.full_like(g, 255)], -1)
    if arr.dtype != np.uint8:                      # float RGB in [0, 1]: matplotlib scales in the array's own precision and truncates
        if arr.max() > 1 or arr.min() < 0:
            raise ValueError("Floating point image RGB values must be in the 0..1 range.")
        arr = (arr * 255).astype(np.uint8)
    if arr.shape[-1] == 3:
        arr = np.concatenate([arr, np.full(arr.shape[:2] + (1,), 255, np.uint8)], -1)
    return arr


def imsave(path: str, arr, gray: bool = False):
    Image.fromarray(_to_rgba8(arr, gray), "RGBA").save(path, format="PNG")


def save_exp(save_location_in, input_img, input_depth, input_depth_vis, input_mask, transform_in, transformed_image=None,
             edited_image=None, background_image=None, h=512, w=512, exp_transform_type="Mix", download_input=None,
             download_edit=None):
    """:52-110.  Creates ``<save_location_in>/<exp_transform_type>/<count+1>/`` and writes the experiment files; returns the folder
    (the reference returns nothing)."""
    h, w = int(h), int(w)
    save_location = complete_path(save_location_in) + exp_transform_type
    create_folder(save_location)
    folder_num = count_folders(save_location) + 1
    save_folder = complete_path(complete_path(save_location) + str(folder_num))
    create_folder(save_folder)
    imsave(save_folder + "input_image.png", input_img)
    for name, im in (("transformed_image", transformed_image), ("result", edited_image), ("background_image", background_image),
                     ("download_input", download_input), ("download_edit", download_edit)):
        if im is not None:
            imsave(save_folder + name + ".png", im)
    imsave(save_folder + "input_mask.png", input_mask, gray=True)
    imsave(save_folder + "depth.png", input_depth_vis, gray=True)
    np.save(save_folder + "depth.npy", input_depth)
    np.save(save_folder + "transform.npy", transform_in)
    np.save(save_folder + "image_shape.npy", np.array([h, w]))
    return save_folder


_EXP_FILES = ("input_image.png", "depth.npy", "input_mask.png", "background_image.png", "depth.png", "transform.npy",
              "transformed_image.png", "result.png", "image_shape.npy", "resized_result_ls.png",
              "zero123/lama_followed_by_zero123_result.png", "resized_input_image_png.png", "object_edit/result_object_edit.png",
              "resized_input_mask_png.png", "dragon_diffusion/result_dragon_diffusion.png", "diffhandles/im_edited_diffhandles.png",
              "free_drag/result_free_drag_resized.png")


def read_exp(d_path) -> Dict[str, Optional[np.ndarray]]:
    """:117-159.  Keys ``<basename>_png`` / ``<basename>_npy`` (None when the file is absent), ``image_shape_npy`` defaulting to
    [512, 512], ``path_name``."""
    save_folder = complete_path(d_path)
    out = {}
    for rel in _EXP_FILES:
        f_name = save_folder + rel
        base = os.path.basename(f_name)
        key, f_type = base.split(".")[0], base.split(".")[1]
        if file_exists(f_name):
            out[key + "_" + f_type] = read_image(f_name) if f_type == "png" else np.load(f_name)
        else:
            out[key + "_" + f_type] = None
    if out["image_shape_npy"] is None:
        out["image_shape_npy"] = np.array([512, 512])
    out["path_name"] = d_path
    return out


def list_exp_details(exp_dict, printer=print):
    for k, v in exp_dict.items():
        if v is None:
            printer(k, " None")
        elif k != "path_name":
            printer(k, " ", v.shape, " ", v.min(), " ", v.max())
        else:
            printer(k, " ", v)


def get_exp_types():
    return ["Removal", "Rotation_3D", "Rotation_2D", "Translation_3D", "Scaling", "Mix", "Translation_2D"]


def check_if_exp_root(exp_root_folder, folder_list=None):
    if folder_list is None:
        folder_list = glob.glob(complete_path(exp_root_folder) + "**/")
    types = get_exp_types()
    return any(f.split("/")[-2] in types for f in folder_list)


def compose_transform(translation_x=0.0, translation_y=0.0, translation_z=0.0, rotation_x=0.0, rotation_y=0.0, rotation_z=0.0,
                      scale_x=1.0, scale_y=1.0, scale_z=1.0) -> torch.Tensor:
    """The 4x4 edit transform as ``get_transformed_mask`` builds it (:529-555): T · Sx · Sy · Sz · Rx · Ry · Rz (degrees), float32."""
    t = torch.eye(4).float()
    t = t @ vis_utils.translateMatrix(translation_x, translation_y, translation_z).type_as(t)
    for axis, s in enumerate((scale_x, scale_y, scale_z)):
        if s != 1.0:
            f = torch.eye(4).type_as(t)
            f[axis, axis] = s
            t = t @ f
    for axis, deg in enumerate((rotation_x, rotation_y, rotation_z)):
        if deg != 0.0:
            t = t @ vis_utils.rotateAxis(deg, axis).type_as(t)
    return t
