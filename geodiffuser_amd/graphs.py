"""hipGraph capture of launch-bound UNet passes.

One UNet pass is ~1,700 small kernels; in eager mode the host dispatch (~16 ms) exceeds the GPU time of the pass at the
batch sizes of this path (2-3 latents).  The inversion passes and the no-grad CFG passes have a static launch sequence for a
given (batch, resolution, controller regime), so they are captured once into a hipGraph (``torch.cuda.CUDAGraph``) and
replayed; the HIP kernels behind ``include/geodiff_hip.h`` take an explicit stream, never allocate or synchronise, and are
capture-safe.  Python-side controller state (layer / step counters) only advances while capturing, so the caller re-applies
it after a replay.

A pass is run eagerly the first time a key is seen (warm-up: MIOpen / rocBLAS workspaces, per-resolution tables of the
controller, which need one host sync), captured the second time, replayed afterwards.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Hashable, Optional, Tuple

import torch

ENABLED = os.environ.get("GD_GRAPHS", "1") == "1"


class _Entry:
    __slots__ = ("seen", "graph", "x", "t", "ctx", "out")

    def __init__(self):
        self.seen = 0
        self.graph = None


class GraphedUNet:
    """``runner(key, x, t, ctx) -> noise_pred``; the returned tensor is a static buffer that the next replay overwrites."""

    def __init__(self, unet):
        self.unet = unet
        self.entries: Dict[Hashable, _Entry] = {}
        self.replays = 0

    def reset(self):
        self.entries.clear()

    @torch.no_grad()
    def __call__(self, key: Hashable, x: torch.Tensor, t, ctx: torch.Tensor) -> Tuple[torch.Tensor, bool]:
        """-> (noise_pred, replayed).  ``replayed`` tells the caller that no Python side effect ran."""
        if not ENABLED or torch.is_grad_enabled():
            return self.unet(x, t, encoder_hidden_states=ctx)["sample"], False
        key = (key, tuple(x.shape), x.dtype, tuple(ctx.shape), ctx.dtype)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = _Entry()
        e.seen += 1
        if e.seen == 1:                                        # warm-up pass, eager
            return self.unet(x, t, encoder_hidden_states=ctx)["sample"], False
        tval = int(t)
        if e.graph is None:                                    # capture (the captured pass also executes once: replay below)
            e.x = x.clone()
            e.ctx = ctx.clone()
            e.t = torch.tensor([tval], device=x.device, dtype=torch.long)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                e.out = self.unet(e.x, e.t, encoder_hidden_states=e.ctx)["sample"]
            e.graph = g
            e.graph.replay()
            return e.out, False                                # Python side effects DID run (during capture)
        e.x.copy_(x)
        e.ctx.copy_(ctx)
        e.t.fill_(tval)
        e.graph.replay()
        self.replays += 1
        return e.out, True


class GraphedOptPass:
    """The optimisation pass — UNet forward with the geometry controller's losses, then autograd back to the latent and the
    text embedding — as one hipGraph per edit.

    ``grads(...) -> (d loss / d latents, d loss / d context)`` with ``controller.loss`` / ``controller.loss_log_dict`` left
    as the eager pass leaves them.  The first pass of an edit runs eagerly (it builds the controller's per-resolution tables,
    which needs one host sync per resolution), the second is captured, later ones replay.  What changes from pass to pass
    is read from device memory: latents, embedding, timestep and the adaptive loss weights
    (``controller.loss_weights_device``); the launch sequence itself is fixed while ``controller.graph_key()`` is.
    The graph lives on the controller, i.e. for one edit."""

    def __init__(self, model, transform_coords, guidance_scale):
        self.model = model
        self.transform_coords = transform_coords
        self.guidance_scale = guidance_scale

    def _eager(self, controller, lat, ctx, t, skip_scheduler=False):
        from .diffusion import diffusion_step
        from .optimization import _latent_grads
        with torch.enable_grad():
            diffusion_step(self.model, controller, lat, ctx[2:], t, self.guidance_scale, transform_coords=self.transform_coords,
                           use_cfg=False, return_noise=True, skip_scheduler=skip_scheduler)
            return _latent_grads(lat, controller.loss, ctx)

    def grads(self, controller, latents: torch.Tensor, context: torch.Tensor, t):
        lat = latents.detach().float().requires_grad_(True)                    # editor.py:218
        ctx = context.detach().float().requires_grad_(True)                    # editor.py:221-224
        st = controller.__dict__.setdefault("_opt_graph", {"seen": 0, "key": None, "graph": None})
        key = (controller.graph_key(), tuple(lat.shape), tuple(ctx.shape))
        if not ENABLED or not lat.is_cuda or not hasattr(controller, "graph_key"):
            return self._eager(controller, lat, ctx, t) + (lat, ctx)
        if st["key"] != key:                                                   # regime change (blend / replace window): start over
            release_opt_graph(controller)
            st = controller.__dict__["_opt_graph"] = {"seen": 0, "key": key, "graph": None}
        st["seen"] += 1
        if st["seen"] == 1:
            return self._eager(controller, lat, ctx, t) + (lat, ctx)
        dev = lat.device
        controller.sync_loss_weights(dev)
        if st["graph"] is None:
            st["lat"] = lat.detach().clone().requires_grad_(True)
            st["ctx"] = ctx.detach().clone().requires_grad_(True)
            st["t"] = torch.tensor([int(t)], device=dev, dtype=torch.long)
            layer, step = controller.cur_att_layer, controller.cur_step
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["g_lat"], st["g_ctx"] = self._eager(controller, st["lat"], st["ctx"], st["t"], skip_scheduler=True)
            st["loss"] = controller.loss
            st["log"] = {k: (dict(v) if isinstance(v, dict) else v) for k, v in controller.loss_log_dict.items()}
            st["graph"] = g
            g.replay()                                                         # the capture itself executed nothing
            return st["g_lat"], st["g_ctx"], st["lat"], st["ctx"]              # Python side effects ran during capture
        with torch.no_grad():
            st["lat"].copy_(lat)
            st["ctx"].copy_(ctx)
            st["t"].fill_(int(t))
        st["graph"].replay()
        controller.loss = st["loss"]
        controller.loss_log_dict = {k: (dict(v) if isinstance(v, dict) else v) for k, v in st["log"].items()}
        controller.after_graph_replay()
        return st["g_lat"], st["g_ctx"], st["lat"], st["ctx"]


def release_opt_graph(controller):
    """Free the captured optimisation pass (and its private memory pool) of a controller."""
    st = controller.__dict__.pop("_opt_graph", None)
    if st and st.get("graph") is not None:
        st["graph"].reset()
