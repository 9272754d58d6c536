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

import itertools
import os
import weakref
from typing import Dict, Hashable, Optional, Tuple

import torch

from . import ops

ENABLED = os.environ.get("GD_GRAPHS", "1") == "1"
OPT_PASS_ENABLED = os.environ.get("GD_OPT_GRAPH", "1") == "1"
KV_REFRESH_GRAPH = os.environ.get("GD_KV_REFRESH_GRAPH", "1") == "1"
KV_CACHE = os.environ.get("GD_KV_CACHE", "1") == "1"     # captured no-grad passes: text-row K / V projections outside the graph, per context


IN_FLIGHT = []       # futures of host work that runs on another thread beside the caller (editor.start_ahead)


def wait_in_flight():
    """Before every capture: a stream capture (global mode) forbids allocations and synchronisations on EVERY thread of the process, so
    work running beside the caller has to finish first (its result or exception stays in its future for the caller to collect)."""
    for f in list(IN_FLIGHT):
        f.exception()


class _Entry:
    __slots__ = ("seen", "graph", "x", "t", "ctx", "out", "kv", "kv_src", "kv_ver", "kv_graph", "ahead", "serial")

    def __init__(self):
        self.seen = 0
        self.graph = None
        self.kv = None
        self.kv_src = None
        self.kv_ver = -1
        self.kv_graph = None     # the 16 projections of _refresh_kv as one captured launch (the context changes at every optimisation step)
        self.ahead = None        # reference rows this pass leaves for the NEXT step's optimisation pass (controller.collect_ahead)
        self.serial = None


def _src_version(ctx_src):
    """(source, version) of a context for the K / V cache; (None, -1) = do not cache (no source named, or an inference-mode tensor: it has no
    version counter, so an in-place edit could not be seen)."""
    if ctx_src is None or ctx_src.is_inference():
        return None, -1
    return ctx_src, ctx_src._version


_SERIAL = itertools.count(1)   # names a captured pass whose tensors another captured pass reads by address (never reused, unlike id())
CAPTURES = {"unet": 0, "opt": 0}      # hipGraph captures so far in this process (no-grad passes / optimisation passes): reporting only


class GraphedUNet:
    """``runner(key, x, t, ctx) -> noise_pred``; the returned tensor is a static buffer that the next replay overwrites."""

    def __init__(self, unet):
        self.unet = unet
        self.entries: Dict[Hashable, _Entry] = {}
        self.replays = 0
        self.kv_refreshes = 0       # replays that had to re-project the text rows' K / V (the context changed)

    def reset(self):
        """Drop every captured pass — explicitly and with the device idle: a graph left to the garbage collector may be destroyed in the
        middle of another stream capture, which aborts the process."""
        torch.cuda.synchronize()
        for e in self.entries.values():
            if e.graph is not None:
                e.graph.reset()
            if e.kv_graph is not None:
                e.kv_graph.reset()
        self.entries.clear()
        torch.cuda.synchronize()

    def _cross_modules(self):
        mods = self.__dict__.get("_cross")
        if mods is None:
            mods = self.__dict__["_cross"] = [m for m in self.unet.modules() if getattr(m, "is_cross_attention", False) and hasattr(m, "to_k")]
        return mods

    def _refresh_kv(self, e):
        """K / V of the context rows for every cross-attention layer, into the entry's persistent buffers (allocated on the first call,
        OUTSIDE any capture).  The text rows enter the UNet only through these projections (unet_sd21: ``ctx`` feeds ``attn2`` alone), and
        one context serves 50 inversion passes / the CFG passes after the optimisation window: 16 batched GEMMs per pass that a replay
        does not have to repeat.  Same arithmetic as attention_processors._batched_qkv (one batched GEMM per layer, stacked weights)."""
        from .attention_processors import _stacked_weights
        ctxd = e.ctx.to(self.unet.dtype)
        e2 = ctxd.reshape(-1, ctxd.shape[-1])
        for m in self._cross_modules():
            w2 = _stacked_weights(m, ("to_k", "to_v"), 1.0)
            if w2 is None or w2.shape[1] != e2.shape[1]:
                continue
            buf = e.kv.get(id(m))
            if buf is None:
                e.kv[id(m)] = torch.bmm(e2.unsqueeze(0).expand(2, -1, -1), w2)
            else:
                torch.bmm(e2.unsqueeze(0).expand(2, -1, -1), w2, out=buf)

    @staticmethod
    def _fill_t(buf, t):
        """Timestep(s) of a replay into the pass's device vector without a host-to-device copy: one fill per run of equal values."""
        if isinstance(t, tuple):
            i = 0
            while i < len(t):
                j = i
                while j + 1 < len(t) and int(t[j + 1]) == int(t[i]):
                    j += 1
                buf[i:j + 1].fill_(int(t[i]))
                i = j + 1
        else:
            buf.fill_(int(t))

    @torch.no_grad()
    def __call__(self, key: Hashable, x: torch.Tensor, t, ctx: torch.Tensor, ctx_src: Optional[torch.Tensor] = None,
                 controller=None) -> Tuple[torch.Tensor, bool]:
        """-> (noise_pred, replayed).  ``replayed`` tells the caller that no Python side effect ran.
        ``t``: one timestep, or a tuple with one timestep per batch row (the CFG pass that carries the next step's reference row).
        ``ctx_src``: the tensor ``ctx`` is a pure function of (the caller's context tensor itself, or the one it was sliced / concatenated
        from).  While the SAME tensor object at the SAME ``_version`` comes back, the cached K / V of the text rows stay valid; a strong
        reference is held, so another tensor cannot take its identity.  None: re-compute on every call.
        ``controller`` with ``collect_ahead`` set: the pass leaves row 0's per-layer (q, k, v, out) (attention_processors._leave_ahead); a
        captured pass hands the controller the STATIC tensors of its graph, named by the entry's serial number (an eager pass: serial
        None — its tensors are not addresses a captured consumer may bake in)."""
        collect = controller is not None and getattr(controller, "collect_ahead", False)

        def t_tensor():
            return torch.tensor([int(v) for v in t], device=x.device, dtype=torch.long) if isinstance(t, tuple) else t

        def eager():
            if collect:
                controller._ahead = None
            out = self.unet(x, t_tensor(), encoder_hidden_states=ctx)["sample"]
            if collect:          # eager tensors: no serial (a captured pass must not bake their addresses); the batched form copies out of them at once
                st = controller._ahead
                ok = st is not None and len(st) == controller.num_att_layers and all(isinstance(a, tuple) for a in st)
                controller.ref_stash, controller.ref_stash_serial, controller._ahead = (st if ok else None), None, None
            return out, False

        if not ENABLED or torch.is_grad_enabled():
            return eager()
        key = (key, tuple(x.shape), x.dtype, tuple(ctx.shape), ctx.dtype)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = _Entry()
        e.seen += 1
        if e.seen == 1:                                        # warm-up pass, eager
            return eager()

        def hand_over():
            if collect:
                ok = e.ahead is not None and len(e.ahead) == controller.num_att_layers and all(isinstance(a, tuple) for a in e.ahead)
                controller.ref_stash = e.ahead if ok else None
                controller.ref_stash_serial = e.serial if ok else None

        if e.graph is None:                                    # capture (the captured pass also executes once: replay below)
            e.x = x.clone()
            e.ctx = ctx.clone()
            e.t = torch.zeros(len(t) if isinstance(t, tuple) else 1, device=x.device, dtype=torch.long)
            self._fill_t(e.t, t)
            if KV_CACHE:
                e.kv = {}
                self._refresh_kv(e)
                e.kv_src, e.kv_ver = _src_version(ctx_src)
            wait_in_flight()                                   # first: work the side thread queues after a synchronize would not be drained
            torch.cuda.synchronize()
            CAPTURES["unet"] += 1
            g = torch.cuda.CUDAGraph()
            ops.zero_pool_reset()
            from . import attention_processors as _ap
            _ap.KV_PROVIDER = e.kv
            if collect:
                controller._ahead = None
            try:
                with torch.cuda.graph(g):
                    e.out = self.unet(e.x, e.t, encoder_hidden_states=e.ctx)["sample"]
            finally:
                _ap.KV_PROVIDER = None
            ops.zero_pool_reset()
            if collect:
                e.ahead, e.serial = controller._ahead, next(_SERIAL)       # (the list keeps the graph's tensors alive: never reused inside its pool)
                controller._ahead = None
            e.graph = g
            if e.kv and KV_REFRESH_GRAPH:                      # (same capture event as the pass: not counted separately)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    self._refresh_kv(e)
                e.kv_graph = g2
            e.graph.replay()
            hand_over()
            return e.out, False                                # Python side effects DID run (during capture)
        e.x.copy_(x)
        e.ctx.copy_(ctx)
        src, ver = _src_version(ctx_src)
        if e.kv and (src is None or src is not e.kv_src or ver != e.kv_ver):
            # Inside the optimisation window the edit row's embedding changes at every optimisation step and each of the step forms (2 / 3 /
            # 4 rows) keeps its own K / V: ~35 refreshes per edit, 16 eagerly dispatched batched GEMMs each with ~20 us of host gap between
            # them (profiles/r06_gap_causes.md, a traced run: 11.6 ms of idle device per edit; untraced the host is mostly ahead: 1-4 ms per edit).
            # Captured once, right behind the pass's own capture.
            if e.kv_graph is not None:
                e.kv_graph.replay()
            else:
                self._refresh_kv(e)
            e.kv_src, e.kv_ver = src, ver
            self.kv_refreshes += 1
        self._fill_t(e.t, t)
        e.graph.replay()
        self.replays += 1
        hand_over()
        return e.out, True


# Captured optimisation passes, shared by all edits of the process.  key -> dict(graph, lat, ctx, t, g_lat, g_ctx, loss, log)
_OPT_GRAPHS: Dict[Hashable, dict] = {}
_OPT_GRAPH_LIMIT = 10           # each holds the activations of a batch-1 or batch-2 forward + backward in its private pool
_WARMED = set()                 # (uid, form) whose first pass has run eagerly (library warm-up outside any capture)
_SEEN_LAYERS: Dict[tuple, tuple] = {}   # (id(unet), latent shape) -> (weakref, [(S, heads)] of the hooked layers), learnt from the first eager pass


def reset_opt_graphs():
    """Drop the captured optimisation-pass graphs and what only they keep alive: the backward-data copies of the 3x3 convolution
    weights (~1 GB for SD2.1; both are rebuilt on the next optimisation pass)."""
    for e in _OPT_GRAPHS.values():
        e["graph"].reset()
    _OPT_GRAPHS.clear()
    from .unet_sd21 import release_backward_weights
    release_backward_weights()


def _export_loss_state(controller):
    """What a captured optimisation pass leaves in the controller's Python state (device scalars the graph writes on every replay)."""
    fn = getattr(controller, "export_loss_state", None)
    if fn is not None:                                     # batch.EditBatch: one (loss, log) pair per edit
        return fn()
    return controller.loss, {k: (dict(v) if isinstance(v, dict) else v) for k, v in controller.loss_log_dict.items()}


def _import_loss_state(controller, state):
    fn = getattr(controller, "import_loss_state", None)
    if fn is not None:
        return fn(state)
    controller.loss = state[0]
    controller.loss_log_dict = {k: (dict(v) if isinstance(v, dict) else v) for k, v in state[1].items()}


def _hand_over_ref(controller, st, t):
    """After a captured / replayed optimisation pass: the controller of the CURRENT edit gets the pass's reference-row tensors (static
    addresses of this graph, named by its serial number — never reused, unlike id())."""
    if getattr(controller, "collect_ref", False):
        stash = st.get("ref_stash")
        ok = stash is not None and len(stash) > 0 and all(e is not False for e in stash)
        controller.ref_stash = stash if ok else None
        controller.ref_stash_serial = st["serial"] if ok else None
        controller.ref_stash_t = int(t) if ok else None


class GraphedOptPass:
    """The optimisation pass — UNet forward with the geometry controller's losses, then autograd back to the latent and the
    text embedding — as one hipGraph, reused across edits.

    ``grads(...) -> (d loss / d latents, d loss / d context, latents leaf, context leaf)`` with ``controller.loss`` /
    ``controller.loss_log_dict`` left as the eager pass leaves them.  What changes from pass to pass and from edit to edit is read
    from device memory: latents, embedding, timestep, the adaptive loss weights (``controller.loss_weights_device``) and the
    controller's per-resolution tables, which live in persistent buffers that every new edit overwrites in place
    (``attention_processors._persist``; the inpaint-row list is padded to a bucketed length).  The launch sequence itself is fixed
    for a given ``controller.graph_key()`` + ``controller.table_signature()``; graphs are kept per such key.

    The first optimisation pass a UNet ever sees runs eagerly (MIOpen / rocBLAS warm-up, and it tells which (resolution, heads) pairs
    the hooked layers have); from then on a new edit builds its tables up front (one host sync per resolution) and replays."""

    ctx_text_rows_only = True        # the driver's context is [uncond_ref, uncond_edit, cond_ref, cond_edit]: the pass runs on the last two

    def __init__(self, model, transform_coords, guidance_scale):
        self.model = model
        self.transform_coords = transform_coords
        self.guidance_scale = guidance_scale

    def _eager(self, controller, lat, ctx, t, skip_scheduler=False, edit_row_only=False):
        from .diffusion import diffusion_step
        from .optimization import _latent_grads
        with torch.enable_grad():
            if edit_row_only:          # the reference row's layer tensors come from controller.ref_stash (editor.REF_AHEAD): the UNet runs the edit row
                lat_in, ctx_in = lat[-1:], ctx[-1:]
            else:
                lat_in, ctx_in = lat, (ctx[2:] if self.ctx_text_rows_only else ctx)
            diffusion_step(self.model, controller, lat_in, ctx_in, t, self.guidance_scale,
                           transform_coords=self.transform_coords, use_cfg=False, return_noise=True, skip_scheduler=skip_scheduler)
            return _latent_grads(lat, controller.loss, ctx)

    def _key(self, controller, lat, ctx, edit_row_only=False):
        return (id(self.model.unet), controller.graph_key(), controller.table_signature(), tuple(lat.shape), tuple(ctx.shape),
                self.model.unet.dtype, bool(getattr(controller, "collect_ref", False)),   # (a pass captured without the collection keeps nothing)
                controller.ref_stash_serial if edit_row_only else None)                   # (the captured pass whose tensors this one reads by address)

    def grads(self, controller, latents: torch.Tensor, context: torch.Tensor, t, edit_row_only: bool = False):
        """``edit_row_only``: the reference row of this step went through the UNet in the previous step's CFG pass, which left its per-layer
        q / k / v in ``controller.ref_stash`` (a captured pass's static tensors, ``ref_stash_serial``); forward + backward then run on
        the edit row alone.  Gradients keep the full shapes (zero rows for the reference sample, as before)."""
        lat = latents.detach().float().requires_grad_(True)                    # editor.py:218
        ctx = context.detach().float().requires_grad_(True)                    # editor.py:221-224
        usable = (ENABLED and OPT_PASS_ENABLED and lat.is_cuda and hasattr(controller, "graph_key") and getattr(controller, "persistent_tables", False))
        if not edit_row_only:
            controller.ref_stash_serial = controller.ref_stash_t = None       # (an eager pass leaves tensors no captured CFG pass may read)
        if not usable:
            return self._eager(controller, lat, ctx, t, edit_row_only=edit_row_only) + (lat, ctx)
        uid = (id(self.model.unet), tuple(lat.shape))                          # the hooked layers' resolutions follow the latent size
        seen = _SEEN_LAYERS.get(uid)
        if seen is not None and seen[0]() is not self.model.unet:              # a dead model's id was recycled
            seen = None
            _WARMED.discard((uid, "edit_row_only")); _WARMED.discard((uid, "rows"))
            for k in [k for k in _OPT_GRAPHS if k[0] == uid[0]]:
                _OPT_GRAPHS.pop(k)["graph"].reset()
        form = "edit_row_only" if edit_row_only else "rows"
        if seen is None:                                                       # very first pass on this UNet: eager, learn the layers
            _WARMED.add((uid, form))
            out = self._eager(controller, lat, ctx, t, edit_row_only=edit_row_only)
            _SEEN_LAYERS[uid] = (weakref.ref(self.model.unet),
                                 sorted((S, c["f"], c["D"]) for S, c in controller.masks_cache_dict.items() if "f" in c))
            return out + (lat, ctx)
        if (uid, form) not in _WARMED:
            # the first pass of each FORM (the rows beside the edit row: with or without the reference row) on this UNet runs eagerly too:
            # forward + backward at another batch size meet convolution / GEMM shapes no pass has run yet, and a solver search (MIOpen find)
            # or a library workspace allocation inside a stream capture invalidates it
            _WARMED.add((uid, form))
            return self._eager(controller, lat, ctx, t, edit_row_only=edit_row_only) + (lat, ctx)
        layers = seen[1]
        if not all(S in controller.masks_cache_dict and "f" in controller.masks_cache_dict[S] for S, *_ in layers):
            q_like = torch.empty(1, device=lat.device, dtype=self.model.unet.dtype)
            controller.prebuild_tables(layers, q_like, self.transform_coords)
        dev = lat.device
        controller.sync_loss_weights(dev)
        key = self._key(controller, lat, ctx, edit_row_only)
        st = _OPT_GRAPHS.get(key)
        if st is None:
            while len(_OPT_GRAPHS) >= _OPT_GRAPH_LIMIT:                        # oldest first
                _OPT_GRAPHS.pop(next(iter(_OPT_GRAPHS)))["graph"].reset()
            st = {"lat": lat.detach().clone().requires_grad_(True), "ctx": ctx.detach().clone().requires_grad_(True),
                  "t": torch.tensor([int(t)], device=dev, dtype=torch.long)}
            wait_in_flight()
            torch.cuda.synchronize()
            CAPTURES["opt"] += 1
            g = torch.cuda.CUDAGraph()
            ops.zero_pool_reset()
            with torch.cuda.graph(g):
                st["g_lat"], st["g_ctx"] = self._eager(controller, st["lat"], st["ctx"], st["t"], skip_scheduler=True, edit_row_only=edit_row_only)
            ops.zero_pool_reset()
            st["state"] = _export_loss_state(controller)
            st["graph"] = g
            # the reference rows this pass leaves for the CFG pass of the same step (controller.collect_ref): tensors of THIS graph
            st["ref_stash"], st["serial"] = getattr(controller, "ref_stash", None), next(_SERIAL)
            _OPT_GRAPHS[key] = st
            g.replay()                                                         # the capture itself executed nothing
            _hand_over_ref(controller, st, t)
            return st["g_lat"], st["g_ctx"], st["lat"], st["ctx"]              # Python side effects ran during capture
        with torch.no_grad():
            st["lat"].copy_(lat)
            st["ctx"].copy_(ctx)
            st["t"].fill_(int(t))
        st["graph"].replay()
        _import_loss_state(controller, st["state"])
        controller.after_graph_replay()
        _hand_over_ref(controller, st, t)
        return st["g_lat"], st["g_ctx"], st["lat"], st["ctx"]


def release_opt_graph(controller):
    """Kept for callers of the per-edit design: captured optimisation passes now outlive the controller (``reset_opt_graphs``)."""
    controller.__dict__.pop("_opt_graph", None)
