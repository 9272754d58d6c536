"""ORACLE tooling (build container only; never runs on the GPU box, never shipped).

Imports the reference's own Python (``/root/reference/GeoDiffuser``) on CPU so that
``oracle/gen_golden.py`` can record golden input/output vectors from it.  The reference cannot be
imported directly here: torchvision, diffusers, pytorch3d, cupy, cv2, IPython,
pytorch_lightning, xformers and skimage are absent.  This module

  * registers stand-in modules in ``sys.modules`` for those imports.  They contain NO reference
    code; the only arithmetic they carry is (a) ``T.Resize`` := ``F.interpolate(bilinear,
    align_corners=False, antialias=False)`` and (b) our own restatement of pytorch3d's
    ``Pointclouds`` / ``rasterize_points`` / ``compositing.alpha_composite`` (oracle/ref_cpu.py),
    which is why the splat boundary stays "parity unpinned";
  * maps ``Tensor.to("cuda*")`` to CPU (the reference hard-codes "cuda" at
    U/warp_utils.py:810-812, U/generic_torch.py:132);
  * wraps ``RasterizePointsXYsBlending.forward`` to receive ``pts3D.clone()`` — on an fp32 CPU run
    the reference's in-place x/y negation (U/warp_utils.py:90-91) would otherwise mutate the cached
    coordinates and mirror every second call; on the real GPU path q is fp16 so ``.to(float32)``
    copies and the flip never happens.
"""
from __future__ import annotations

import os
import sys
import types

import torch
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "GeoDiffuser", "utils"))


class _Anything:
    """Placeholder for any class/function the reference imports by name but never calls here."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("stub called")

    def __getattr__(self, name):
        return _Anything()


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)

    def _getattr(attr, _n=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything

    m.__getattr__ = _getattr  # PEP 562
    sys.modules[name] = m
    return m


def install_stubs() -> None:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ref_cpu  # our restatement (for the pytorch3d stand-ins only)

    # ---- torchvision.transforms ---------------------------------------------------------
    class InterpolationMode:
        BILINEAR = "bilinear"
        NEAREST = "nearest"

    class Resize:
        def __init__(self, size, antialias=None, interpolation=InterpolationMode.BILINEAR):
            self.size = size if isinstance(size, (tuple, list)) else (size, size)

        def __call__(self, x):
            return F.interpolate(x, size=tuple(self.size), mode="bilinear", align_corners=False, antialias=False)

    tv = _module("torchvision")
    tvt = _module("torchvision.transforms", Resize=Resize, InterpolationMode=InterpolationMode)
    tv.transforms = tvt
    tvt.v2 = _module("torchvision.transforms.v2")

    # ---- diffusers --------------------------------------------------------------------------
    d = _module("diffusers")
    dm = _module("diffusers.models")
    dma = _module("diffusers.models.attention_processor", USE_PEFT_BACKEND=False)
    d.models = dm
    dm.attention_processor = dma

    # ---- pytorch3d (stand-ins carrying OUR restatement) ---------------------------------------
    class Pointclouds:
        def __init__(self, points, features=None):
            self._points = points          # [B, P, 3]
            self._features = features      # [B, P, F]

        def points_padded(self):
            return self._points

        def features_packed(self):
            B, P, Fc = self._features.shape
            return self._features.reshape(B * P, Fc)

    def rasterize_points(pointclouds, image_size, radius, points_per_pixel, *a, **k):
        return ref_cpu.rasterize_points(pointclouds.points_padded(), image_size, radius, points_per_pixel)

    class compositing:
        @staticmethod
        def alpha_composite(idx, alphas, feat):
            return ref_cpu.alpha_composite(idx, alphas, feat)

    p3 = _module("pytorch3d")
    p3.structures = _module("pytorch3d.structures", Pointclouds=Pointclouds)
    p3.renderer = _module("pytorch3d.renderer", compositing=compositing)
    p3.renderer.points = _module("pytorch3d.renderer.points", rasterize_points=rasterize_points)
    p3.renderer.mesh = _module("pytorch3d.renderer.mesh")
    p3.renderer.mesh.rasterizer = _module("pytorch3d.renderer.mesh.rasterizer")

    # ---- misc absent packages ------------------------------------------------------------------
    cp = _module("cupy")
    cp.memoize = lambda *a, **k: (lambda fn: fn)
    _module("cv2")
    ip = _module("IPython")
    ip.display = _module("IPython.display", display=lambda *a, **k: None)
    _module("pytorch_lightning", seed_everything=lambda *a, **k: None)
    xf = _module("xformers")
    xf.ops = _module("xformers.ops")
    sk = _module("skimage")
    sk.exposure = _module("skimage.exposure")
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        mp = _module("matplotlib")
        mp.pyplot = _module("matplotlib.pyplot")

    # ---- "cuda" -> cpu --------------------------------------------------------------------------
    if not getattr(torch.Tensor, "_gd_to_patched", False):
        _orig_to = torch.Tensor.to

        def _to(self, *args, **kwargs):
            args = tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else
                         (torch.device("cpu") if isinstance(a, torch.device) and a.type == "cuda" else a)
                         for a in args)
            if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
                kwargs["device"] = "cpu"
            return _orig_to(self, *args, **kwargs)

        torch.Tensor.to = _to
        torch.Tensor._gd_to_patched = True


_REF = None


def import_reference():
    """Returns a namespace with the reference modules used for golden generation."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError("reference not present (this tool only runs in the build container)")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import GeoDiffuser.utils.warp_utils as warp_utils
    import GeoDiffuser.utils.generic_torch as generic_torch
    import GeoDiffuser.utils.attention_sharing as attention_sharing
    import GeoDiffuser.utils.attention_processors as attention_processors
    import GeoDiffuser.utils.loss as loss
    import GeoDiffuser.utils.optimization as optimization
    import GeoDiffuser.utils.generic as generic
    import GeoDiffuser.utils.image_processing as image_processing

    # clone-wrapper (see module docstring)
    cls = warp_utils.RasterizePointsXYsBlending
    if not getattr(cls, "_gd_wrapped", False):
        orig = cls.forward

        def fwd(self, pts3D, src):
            return orig(self, pts3D.clone(), src)

        cls.forward = fwd
        cls._gd_wrapped = True
    # the distance singleton defaults to device="cuda" (U/generic_torch.py:130); .to is patched above.
    _REF = types.SimpleNamespace(warp_utils=warp_utils, generic_torch=generic_torch,
                                 attention_sharing=attention_sharing, attention_processors=attention_processors,
                                 loss=loss, optimization=optimization, generic=generic,
                                 image_processing=image_processing)
    return _REF


def import_reference_ui():
    """``GeoDiffuser.utils.ui_utils`` (experiment-folder I/O, transform composition) with its UI / perception / camera imports
    (gradio, pyrealsense2, the SAM + depth front-end module) replaced by empty stand-ins — none of them is touched by
    ``save_exp`` / ``read_exp`` / ``read_image`` / ``check_if_exp_root`` or by the matrix composition of ``get_transformed_mask``."""
    import_reference()
    for name in ("gradio", "pyrealsense2"):
        if name not in sys.modules:
            _module(name)
    if "GeoDiffuser.utils.depth_predictor" not in sys.modules:
        _module("GeoDiffuser.utils.depth_predictor")
    import GeoDiffuser.utils.ui_utils as ui_utils
    return ui_utils


def import_reference_batch_driver():
    """The reference's ``large_scale_editor.py`` (repo root) with its perception imports (SAM, DepthAnything, DPT/MiDaS) replaced by
    empty stand-ins; only ``perform_exp`` (the per-edit-type configuration table) is exercised."""
    import_reference_ui()
    for name in ("GeoDiffuser.segment_anything", "GeoDiffuser.depth_anything", "GeoDiffuser.depth_anything.util",
                 "GeoDiffuser.depth_anything.util.transform", "GeoDiffuser.depth_anything.dpt", "GeoDiffuser.dpt",
                 "GeoDiffuser.dpt.models", "GeoDiffuser.dpt.midas_net", "GeoDiffuser.dpt.transforms"):
        if name not in sys.modules:
            _module(name)
    sys.modules["torchvision.transforms"].Compose = object
    import large_scale_editor
    return large_scale_editor
