"""B independent edits in ONE process, sharing every UNet pass — the MI355X-first shape of the reference's batch driver.

The reference processes one edit at a time on a batch of 2 latents (GeoDiffuser/utils/editor.py:603; its batch driver is a sequential
loop, large_scale_editor.py:320-402).  One such edit fills about half of an MI355X (four OS processes on one device give 1.85 x,
profiles/r04_in_flight.md).  Here B edits advance in lockstep: the inversion pass runs at batch B, the optimisation pass at 2 B
(reference rows, then edit rows), the CFG pass at 3 B (uncond_edit, cond_ref, cond_edit rows) — ROLE-MAJOR batches, row = role * B + edit,
so every role is one contiguous slice.  One copy of the weights, 1 / B of the launches per edit, B x the heads in every attention launch,
convolutions and GEMMs out of the launch-bound regime.  Each edit keeps what is its own: controller tables (masks, splat tables, row
lists: persistent buffers named by the edit's ``slot``), loss weights and adaptive schedule, loss log, DDIM trajectory, text context.

Every per-edit quantity is computed by the same kernels on the same values as in the one-edit driver (editor.text2image_ldm_stable),
so an edit's result does not depend on what else is in the batch beyond the UNet's own batch-size-dependent GEMM / convolution kernel
selection (measured in tests/test_batch.py).

    EditBatch                      the controller the processors see: B geometry controllers behind one processor protocol
    text2image_ldm_stable_batch    the per-step loop (editor.py:65-423) for B edits in lockstep
    perform_geometric_edit_batch   pre-pass, batched inversion, loop, decode, post-process for a list of edits
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import numpy as np
import torch

import math

from . import attention_processors as AP
from . import editor as E
from . import graphs, ops, vis_utils, warp_utils
from .attention_processors import (AttentionGeometryEdit, AttentionGeometryRemover, VanillaAttentionProcessor, _persist,
                                   register_attention_control_diffusers, set_attn_processor_for_edit)
from .attention_sharing import AttentionControl, attention, attention_tok
from .diffusion import _sched_step, _unet_nograd, encode_text, latent2image
from .generic_torch import binarize_tensor, reshape_attention_mask, reshape_transform_coords, torch_erode
from .inversion import NullInversion
from .optimization import adaptive_optimization_step_editing, adaptive_optimization_step_remover
from .warp_utils import warp_grid_edit


MERGED = os.environ.get("GD_BATCH_MERGED", "1") == "1"     # 0: every edit's rows through its own controller (gather / scatter copies)
GROUP = 8                                                  # gd_attn_fwd / gd_attn_fwd_pair: one row-list / pair segment per edit, 8 per launch
MAX_EDITS = 16                                             # more than GROUP edits: the UNet passes run on all, a hooked layer launches per group


def _rows(seg, a, b):
    """Rows [a, b) of the q / k / v / out / lse tensors of an attention segment tuple (whatever follows — warp tables, row list — is kept)."""
    return tuple(t[a:b] if t is not None else None for t in seg[:5]) + tuple(seg[5:])


def _groups(B, group=GROUP):
    return [(j0, min(j0 + group, B)) for j0 in range(0, B, group)]


MAX_ROWLIST_HEADS = 256        # GD_ATTN_MAX_ORDER: a launch whose segments carry query row lists addresses its heads through order[]


def _group_size(plain, per_edit, tails, unit, B, hpr):
    """Edits per launch: GROUP, fewer where the launch carries query row lists and its heads would exceed the kernel's order table (10-head
    64^2 layers of SDXL-shaped UNets: 4 * B * 10 heads — ADVICE r05: this used to end in GD_EUNSUPPORTED in the middle of an edit)."""
    g = min(GROUP, B)
    if not any(len(seg) > 6 and seg[6] is not None for seg in per_edit):
        return max(1, g)
    while g > 1:
        ng = -(-B // g)
        rows = sum(-(-seg[0].shape[0] // ng) for seg in plain) + sum(seg[0].shape[0] for seg in per_edit[:g]) + unit * g * len(tails)
        if rows * hpr <= MAX_ROWLIST_HEADS:
            break
        g -= 1
    return max(1, g)


def _attn_fwd_grouped(plain, per_edit, tails, unit, B, scale, **kw):
    """ops.attn_fwd over plain + per_edit + tails, at most GROUP edits per launch.  ``plain``: segments whose rows are independent of the
    edits' order (vanilla rows: any contiguous share of them goes with any launch); ``per_edit``: one segment per edit; ``tails``: segments
    with ``unit`` rows per edit, in edit order (the replace attention: token-major unit 1, head-major unit f).  B <= GROUP: ONE launch."""
    gs = _groups(B, _group_size(plain, per_edit, tails, unit, B, kw.get("heads") or 1))
    for g, (j0, j1) in enumerate(gs):
        segs = []
        for seg in plain:
            R = seg[0].shape[0]
            a, b = g * R // len(gs), (g + 1) * R // len(gs)
            if b > a:
                segs.append(_rows(seg, a, b) if len(gs) > 1 else seg)
        segs += per_edit[j0:j1]
        segs += [_rows(seg, j0 * unit, j1 * unit) if len(gs) > 1 else seg for seg in tails]
        ops.attn_fwd(segs, scale, **kw)


def _attn_fwd_pair_grouped(plain, a_sides, b_sides, ms, scale, **kw):
    """ops.attn_fwd_pair with one blend pair per edit, at most GROUP pairs per launch (the plain segments' rows shared out as above)."""
    B = len(a_sides)
    gs = _groups(B)
    for g, (j0, j1) in enumerate(gs):
        segs = []
        for seg in plain:
            R = seg[0].shape[0]
            a, b = g * R // len(gs), (g + 1) * R // len(gs)
            if b > a:
                segs.append(_rows(seg, a, b) if len(gs) > 1 else seg)
        ops.attn_fwd_pair(segs + a_sides[j0:j1], b_sides[j0:j1], ms[j0:j1], scale, **kw)


class _EditLayerBatch(torch.autograd.Function):
    """attention_processors._EditLayer for the B edits of a role-major optimisation-pass batch in one autograd node: q / k / v token-major
    [2 B, N | M, heads*64] (reference rows, then edit rows).  What the edits share runs once — the layout change, ONE attention launch
    (the reference rows of all edits as one segment, the replace attention of all edits as one segment, one warped / row-list segment per
    edit), the dq kernel over all edit rows, the layout change back (with every edit's own blend) — what is theirs runs per edit on
    contiguous head-major slices: probability maps, correlation, amodal target, the loss launch with its running sums, the removal
    backward and the fold.  Per edit the same kernels on the same values as _EditLayer.
    Returns (out [2 B, N, heads*64], running loss [B] f32 = the edits' losses so far + this layer's)."""

    @staticmethod
    def forward(ctx, q, k, v, batch, cs, is_cross, scale, q_pre, heads, running, log_accs):
        subs, B, f = batch.subs, batch.B, heads
        remover = batch._is_remover
        tok_shapes = (q.shape, k.shape)
        q, k, v = ops.heads_split((q, k, v), heads)                               # [2 B f, N | M, 64]
        Bf = B * f
        N, D, S = q.shape[1], q.shape[2], cs[0]["S"]
        dev, dt = q.device, q.dtype
        want_losses = N >= 32 ** 2
        blend = batch.cur_step < int(batch.num_steps * batch.obj_edit_step)
        if remover and not blend:
            raise NotImplementedError("the remover's identity attention (cur_step >= obj_edit_step) is never differentiated by the reference driver")
        sl = [slice(j * f, (j + 1) * f) for j in range(B)]
        q_base, k_base, v_base, q_edit, k_edit = q[:Bf], k[:Bf], v[:Bf], q[Bf:], k[Bf:]
        van = torch.empty(Bf, N, D, dtype=dt, device=dev)                          # the reference rows' vanilla outputs
        lse_van = torch.empty(Bf, N, dtype=torch.float32, device=dev) if want_losses else None
        replace_out = torch.empty(Bf, N, D, dtype=dt, device=dev)
        lse_e = torch.empty(Bf, N, dtype=torch.float32, device=dev)
        plain = [(q_base, k_base, v_base, van, lse_van)]
        per_edit = []
        acts = None
        if not remover:
            edit_out = torch.empty(Bf, N, D, dtype=dt, device=dev)
            if AP.WARP_ROWS and (not is_cross) and all("edit_rows" in c for c in cs) and k_base.shape[1] % 256 == 0:
                acts = [torch.empty(f, c["edit_rows"].numel(), D, dtype=dt, device=dev) for c in cs]
                for j, c in enumerate(cs):                                             # only the rows inside each edit's soft mask
                    per_edit.append((q_base[sl[j]], k_base[sl[j]], v_base[sl[j]], acts[j], None, (c["idx"], c["w"], c["m_edit"]), (c["edit_rows"], c["n_edit_rows"])))
            else:
                for j, c in enumerate(cs):
                    per_edit.append((q_base[sl[j]], k_base[sl[j]], v_base[sl[j]], edit_out[sl[j]], None, (c["idx"], c["w"], c["m_edit"])))
            K = k_edit if is_cross else k_base                                         # :432 / :555
        else:
            K = k_base                                                                 # :790,882
        tails = [(q_edit, K, v_base, replace_out, lse_e)]                              # :433,557 / :791,883
        _attn_fwd_grouped(plain, per_edit, tails, f, B, scale, q_scaled=2 if (q_pre and AP.OPT_PRE and dt in (torch.bfloat16, torch.float16)) else 0)
        if remover:
            edit_out = van.clone() if want_losses else van
        elif acts is not None:
            for j, c in enumerate(cs):
                ops.blend_merge(van[sl[j]], acts[j], c["edit_pos"], None, None, eo_out=edit_out[sl[j]], out=None)
        new_running = running
        saved = []
        if want_losses:
            kind = "cross" if is_cross else "self"
            new_running = torch.empty(B, dtype=torch.float32, device=dev)
            for j, (s, c) in enumerate(zip(subs, cs)):
                R = c["rows"].numel()
                use_amodal = (not remover) and N > 32 ** 2
                m_edit_l = c["m_edit"] if not remover else c["zeros"]
                wv = s.loss_weights_device(kind, dev)
                best = Pb = Pe = tgt = None
                Kj = K[sl[j]]
                if R > 0:
                    best = torch.empty(f, R, 2, dtype=torch.int64, device=dev)
                    Pb, Pe = ops.attn_probs_pair(q_base[sl[j]], k_base[sl[j]], lse_van[sl[j]], q_edit[sl[j]], Kj, lse_e[sl[j]], c["rows"], c.get("n_rows"),
                                                 scale, zero=best)
                    ops.removal_corr_max_nz(Pe, Pb, c["m_inp"], c["m_wo"], c.get("n_rows"), best)
                if use_amodal:
                    tgt = ops.amodal_target(edit_out[sl[j]], c["nn_idx"], c["nn_w"], c["m_edit"], S)
                res = ops.edit_losses_fused(edit_out[sl[j]], replace_out[sl[j]], tgt, c["m_wo"], m_edit_l, c.get("w_dist"), c.get("m_amodal"), S, best,
                                            c["rows"] if R > 0 else None, c.get("n_rows"), c["inv5"], c["inv_rm"], wv, c["inv5_bwd"], use_amodal,
                                            log_acc=log_accs[j], running=running[j:j + 1].detach(), loss_out=new_running[j:j + 1])
                if res[4] is not None:
                    s._last_removal_aux = res[4]
                saved.append((tgt, Pe, Pb, res[2], res[3], res[4]))
        # output: token-major, the edit rows blended with their own masks (:502-508,617-622 / :831-834)
        srcs = [van[sl[j]] for j in range(B)]
        blends = None
        if (not remover) and blend:
            srcs += [edit_out[sl[j]] for j in range(B)]
            blends = [(B + j, replace_out[sl[j]], cs[j]["m_edit"]) for j in range(B)]
        else:
            srcs += [replace_out[sl[j]] for j in range(B)]
        out = ops.heads_merge(srcs, heads, N, D, dt, dev, blend=blends)
        ctx.save_for_backward(q_edit, K, v_base, replace_out, lse_e, edit_out if want_losses else None)
        ctx.saved, ctx.cs = saved, cs
        ctx.meta = dict(B=B, f=f, is_cross=is_cross, scale=scale, remover=remover, blend=blend, want_losses=want_losses, S=S, heads=heads,
                        tok_shapes=tok_shapes, N=N, D=D)
        return out, new_running

    @staticmethod
    def backward(ctx, g_out, g_run):
        q_edit, K, v_base, replace_out, lse_e, edit_out = ctx.saved_tensors
        m, cs = ctx.meta, ctx.cs
        B, f, S, N, D, heads = m["B"], m["f"], m["S"], m["N"], m["D"], m["heads"]
        dev, dt = q_edit.device, q_edit.dtype
        Bf = B * f
        sl = [slice(j * f, (j + 1) * f) for j in range(B)]
        have_loss = m["want_losses"] and g_run is not None
        eo = edit_out if edit_out is not None else replace_out
        need_dk = m["is_cross"] and not m["remover"]
        dro = torch.empty(Bf, N, D, dtype=dt, device=dev)
        rms = []
        gr = g_run.float().contiguous() if have_loss else None
        for j, c in enumerate(cs):                                                       # [loss backward + row dots] per edit
            m_edit_l = c["m_edit"] if not m["remover"] else c["zeros"]
            gout = g_out[B + j:B + j + 1].contiguous() if g_out is not None else None    # token-major row of this edit's output gradient
            gscale = gr[j:j + 1] if have_loss else None
            rm_args = rm_ws = None
            if have_loss:
                tgt, Pe, Pb, coefs, rm_coef, aux = ctx.saved[j]
                if Pe is not None:
                    rm_args, rm_ws = ops.removal_bwd_args(Pe, Pb, q_edit[sl[j]], K[sl[j]], c["rows"], aux, c["m_inp"], c["m_wo"], 1.0, gscale, rm_coef,
                                                          m["scale"], c.get("n_rows"), need_dk)
                ops.edit_losses_bwd_rowdot(eo[sl[j]], replace_out[sl[j]], tgt, c["m_wo"], m_edit_l, c.get("w_dist"), c.get("m_amodal"), gout, coefs,
                                           gscale, blend=(m["blend"] and not m["remover"]), S=S, rm=rm_args, gout_tok=True, out=dro[sl[j]])
            else:
                ops.edit_losses_bwd(eo[sl[j]], replace_out[sl[j]], None, c["m_wo"], m_edit_l, c.get("w_dist"), c.get("m_amodal"), gout,
                                    AP._zeros5(dev), None, blend=(m["blend"] and not m["remover"]), S=S, gout_tok=True, out=dro[sl[j]])
            rms.append((rm_args, rm_ws))
        dq = torch.empty(Bf, N, D, dtype=dt, device=dev)
        dk32, kchunks, part_ptr, bwd_ws = ops.attn_bwd_nofold(q_edit, K, v_base, replace_out, lse_e, dro, m["scale"], need_dk, dq)   # all edits' rows
        for j, c in enumerate(cs):                                                       # [removal dS K] [fold] per edit
            rm_args, rm_ws = rms[j]
            if rm_args is not None:
                rm_args.dk_f32 = dk32[sl[j]].data_ptr() if dk32 is not None else None
                ops.removal_bwd_nofold(rm_args, dt)
        for j, c in enumerate(cs):
            rm_args, rm_ws = rms[j]
            if kchunks > 1 or rm_ws is not None:
                ops.edit_dq_fold(part_ptr if kchunks > 1 else None, kchunks, Bf, N, D, rm_ws, K.shape[1], c["rows"].numel(), c["inp_pos"],
                                 ctx.saved[j][5]["wgt"] if rm_ws is not None else None, dq, head0=j * f, nheads=f)
        del bwd_ws
        (Bq, Nq, _), (Bk, Mk, _) = m["tok_shapes"]
        grad_q = ops.heads_merge([None] * B + [dq[sl[j]] for j in range(B)], heads, Nq, D, dt, dev)
        grad_k = None
        if dk32 is not None:
            grad_k = ops.heads_merge([None] * B + [dk32[sl[j]] for j in range(B)], heads, Mk, D, dt, dev)
        return grad_q, grad_k, None, None, None, None, None, None, None, g_run, None


class EditBatch(AttentionControl):
    """B geometry controllers of one type behind the processors of one UNet.  Batch rows are role-major (row = role * B + edit); a
    hooked call hands every edit its own rows, its own tables and its own loss state."""

    supports_token_major = True
    supports_scaled_q_head_major = True
    store_attention_maps = False
    local_blend = None

    def __init__(self, subs: Sequence, coords: Sequence[torch.Tensor]):
        super().__init__()
        if len({type(s) for s in subs}) != 1:
            raise ValueError("EditBatch: the edits of a batch must share one controller type (group editors and removers separately)")
        if len(subs) != len(coords):
            raise ValueError("EditBatch: one transform-coordinate tensor per edit")
        self.subs = list(subs)
        self.B = len(self.subs)
        self.coords = list(coords)
        for j, s in enumerate(self.subs):
            s.slot = j                                   # names this edit's persistent device tables (attention_processors._persist)
        self.use_cfg, self.coords_base, self.coords_edit, self.n_batch = True, (2, 3), (3, 4), None
        self.heads_tok = self.heads_opt = 0
        self.q_scaled_tok = self.q_scaled_hm = False
        self._is_remover = self.subs[0]._is_remover
        self.num_steps = self.subs[0].num_steps
        self.obj_edit_step = self.subs[0].obj_edit_step
        self.num_self_replace = self.subs[0].num_self_replace
        # reference rows of the CFG pass from the optimisation pass of the same step (attention_processors.ref_stash, for B edits:
        # token-major [B, N | M, C] views per hooked layer)
        self.collect_ref = self.use_ref_stash = False
        self.ref_stash = self.ref_stash_serial = self.ref_stash_t = None
        self._ref_pos = 0

    def _leave_ref(self, entry):
        if self.collect_ref:
            if self.cur_att_layer == 0 or self.ref_stash is None:
                self.ref_stash = []
            self.ref_stash.append(entry)

    # -- state that the driver / the processors set on "the controller" and every edit must see ---------------------------
    @property
    def persistent_tables(self):
        return all(getattr(s, "persistent_tables", False) for s in self.subs)

    @persistent_tables.setter
    def persistent_tables(self, v):
        for s in self.subs:
            s.persistent_tables = v

    @property
    def masks_cache_dict(self):            # the hooked (resolution, heads, head dim) set is the same for every edit
        return self.subs[0].masks_cache_dict

    @property
    def rows_identical(self):
        return tuple(bool(s.rows_identical) for s in self.subs)

    @property
    def loss(self):
        """Sum over the edits: edits do not interact inside the UNet (per-sample norms, per-sample attention), so d(sum) / d(edit j's
        latent) is d(edit j's loss) / d(its latent) — one backward pass serves all of them."""
        tot = None
        for s in self.subs:
            if torch.is_tensor(s.loss):
                tot = s.loss if tot is None else tot + s.loss
        return 0.0 if tot is None else tot

    def _sync(self, s):
        s.num_att_layers = self.num_att_layers
        s.coords_base, s.coords_edit, s.use_cfg, s.n_batch = self.coords_base, self.coords_edit, self.use_cfg, self.n_batch
        s.heads_tok, s.heads_opt, s.q_scaled_tok, s.q_scaled_hm = self.heads_tok, self.heads_opt, self.q_scaled_tok, self.q_scaled_hm

    # -- the processor protocol --------------------------------------------------------------------------------------------
    def forward(self, q, k, v, is_cross: bool, place_in_unet: str, transform_coords=None, scale=None, mask=None):
        B = self.B
        heads = self.heads_tok or self.heads_opt
        if not heads:
            raise NotImplementedError("EditBatch needs the token-major layer forms (64-wide heads, 16-bit, GD_TOKEN_MAJOR / GD_TOK_OPT on)")
        active = is_cross or (self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1])
        ref = None
        if self.use_ref_stash and self.heads_tok:
            ref = self.ref_stash[self._ref_pos]
            self._ref_pos += 1
        if not active:                                   # (self-attention past its replace window: plain attention for the whole batch)
            for s in self.subs:
                s.cur_att_layer += 1
            if self.heads_tok:
                return attention_tok(q, k, v, scale, heads, q_scaled=self.q_scaled_tok)
            # the optimisation pass (self_replace_steps < optimize_steps): like _GeometryControllerBase.forward and the reference (:646-647) —
            # plain attention with autograd, no loss, nothing left for the CFG pass of this step (this used to fall into the
            # merged edit layer)
            self._leave_ref(None)
            Bq, N, C = q.shape
            sc = AP.LN2 if self.q_scaled_hm else scale
            hm = lambda t: t.view(Bq, t.shape[1], heads, C // heads).permute(0, 2, 1, 3).reshape(Bq * heads, t.shape[1], C // heads)
            return attention(hm(q), hm(k), hm(v), sc).view(Bq, heads, N, C // heads).permute(0, 2, 1, 3).reshape(Bq, N, C)
        if MERGED and AP.FUSED_WARP and AP.FUSED_LAYER and q.is_cuda and q.shape[2] == heads * 64 and B <= MAX_EDITS \
                and not any(self.rows_identical):
            for s in self.subs:
                self._sync(s)
            out = self._forward_cfg(q, k, v, is_cross, float(scale), heads, ref=ref) if self.heads_tok else self._forward_opt(q, k, v, is_cross, scale, heads)
            if out is not None:
                for s in self.subs:
                    s.cur_att_layer += 1
                    s.heads_tok = s.heads_opt = 0
                return out
        if self.use_ref_stash:
            raise RuntimeError("EditBatch: a CFG pass without its reference rows needs the merged layer forms (disable GD_REF_FROM_OPT)")
        if not self.heads_tok:
            self._leave_ref(False)                       # the per-edit route leaves nothing for the CFG pass: the collection is void
        outs = []
        for j, s in enumerate(self.subs):
            self._sync(s)
            # rows j, B + j, 2 B + j ... of the role-major batch: this edit's [ref, edit] / [uncond_edit, cond_ref, cond_edit] rows
            outs.append(s(q[j::B].contiguous(), k[j::B].contiguous(), v[j::B].contiguous(), is_cross, place_in_unet,
                          transform_coords=self.coords[j], scale=scale, mask=None))
            s.heads_tok = s.heads_opt = 0
        return torch.stack(outs, 1).reshape(outs[0].shape[0] * B, *outs[0].shape[1:])

    def _forward_cfg(self, q, k, v, is_cross: bool, scale: float, heads: int, ref=None):
        """The no-grad CFG pass of all edits, token-major q / k / v [(cb + 1) B, N | M, heads*64] role-major (cb vanilla roles, then the edit
        rows): what _GeometryControllerBase._forward_tok does for one edit, with ONE attention launch for the batch — the vanilla rows of all
        edits as one segment, the replace attention of all edits as one segment (both read contiguous role slices), one warped / row-list
        segment (or one blend pair) per edit with that edit's tables."""
        B, subs = self.B, self.subs
        b0, e0, cb = self.coords_base[0], self.coords_edit[0], self.coords_base[-1]
        if ref is not None:                               # rows [uncond_edit x B | cond_edit x B]: the reference rows come from ref
            cb = e0
        N, C = q.shape[1], q.shape[2]
        S = int(math.isqrt(N))
        cs = [s._tables(S, heads, q, self.coords[j]) for j, s in enumerate(subs)]
        remover = self._is_remover
        blend = self.cur_step < int(self.num_steps * self.obj_edit_step)
        qs = self.q_scaled_tok
        out_full = torch.empty((cb + 1) * B, N, C, dtype=q.dtype, device=q.device)
        van = (q[:cb * B], k[:cb * B], v[:cb * B], out_full[:cb * B], None)
        if ref is None:
            q_base, k_base, v_base = q[b0 * B:(b0 + 1) * B], k[b0 * B:(b0 + 1) * B], v[b0 * B:(b0 + 1) * B]
            van_base = out_full[b0 * B:(b0 + 1) * B]
        else:
            if not qs or ref[0].shape != (B, N, C) or ref[0].dtype != q.dtype:
                raise RuntimeError("ref_stash does not match this pass (query scaling / shape / dtype): disable GD_REF_FROM_OPT")
            q_base, k_base, v_base, van_base = ref
        q_edit, k_edit, v_edit = q[e0 * B:(e0 + 1) * B], k[e0 * B:(e0 + 1) * B], v[e0 * B:(e0 + 1) * B]
        o_edit = out_full[cb * B:]
        r = lambda t, j: t[j:j + 1]
        if is_cross:
            for s in subs:
                _ = s.cross_replace_alpha[self.cur_step]                                  # :654 (indexing only; value unused)
        if AP.PAIR_BLEND and k.shape[1] <= 128 and ((not remover and blend) or (remover and not blend)):
            # short key lists: both attention outputs of every edit's rows and their blend inside the launch (one pair per edit)
            if not remover:
                a_sides = [(r(q_base, j), r(k_base, j), r(v_base, j), r(o_edit, j), None, (c["idx"], c["w"], c["m_edit"])) for j, c in enumerate(cs)]
                Kb = k_edit if is_cross else k_base
                b_sides = [(r(q_edit, j), r(Kb, j), r(v_base, j)) for j in range(B)]
                ms = [c["m_edit"] for c in cs]
            else:
                a_sides = [(r(q_edit, j), r(k_edit, j), r(v_edit, j), r(o_edit, j), None) for j in range(B)]
                b_sides = [(r(q_edit, j), r(k_base, j), r(v_base, j)) for j in range(B)]
                ms = [c["m_inp"] for c in cs]
            _attn_fwd_pair_grouped([van], a_sides, b_sides, ms, scale, heads=heads, q_scaled=qs)
            return out_full
        per_edit, tails = [], []
        acts = edit_outs = ident_out = None
        replace_out = o_edit                                                              # no blend to come: straight into the edit rows
        if not remover:
            K = k_edit if is_cross else k_base
            if blend:
                replace_out = torch.empty(B, N, C, dtype=q.dtype, device=q.device)
                if AP.WARP_ROWS and (not is_cross) and all("edit_rows" in c for c in cs) and k_base.shape[1] % 256 == 0:
                    acts = [torch.empty(1, c["edit_rows"].numel(), C, dtype=q.dtype, device=q.device) for c in cs]
                    for j, c in enumerate(cs):
                        per_edit.append((r(q_base, j), r(k_base, j), r(v_base, j), acts[j], None, (c["idx"], c["w"], c["m_edit"]), (c["edit_rows"], c["n_edit_rows"])))
                else:
                    edit_outs = torch.empty(B, N, C, dtype=q.dtype, device=q.device)
                    for j, c in enumerate(cs):
                        per_edit.append((r(q_base, j), r(k_base, j), r(v_base, j), r(edit_outs, j), None, (c["idx"], c["w"], c["m_edit"])))
        else:
            K = k_base
            if not blend:
                ident_out = torch.empty(B, N, C, dtype=q.dtype, device=q.device)
                replace_out = torch.empty(B, N, C, dtype=q.dtype, device=q.device)
                tails.append((q_edit, k_edit, v_edit, ident_out, None))
        tails.append((q_edit, K, v_base, replace_out, None))
        _attn_fwd_grouped([van], per_edit, tails, 1, B, scale, heads=heads, q_scaled=qs)
        for j, c in enumerate(cs):
            if acts is not None:          # rows outside the soft edit mask: the reference row's output — merged and blended in one pass
                ops.blend_merge(r(van_base, j), acts[j], c["edit_pos"], r(replace_out, j), c["m_edit"], eo_out=None, out=r(o_edit, j))
            elif edit_outs is not None:
                ops.blend_tokens(r(edit_outs, j), r(replace_out, j), c["m_edit"], out=r(o_edit, j))
            elif ident_out is not None:
                ops.blend_tokens(r(ident_out, j), r(replace_out, j), c["m_inp"], out=r(o_edit, j))
        return out_full

    def _forward_opt(self, q, k, v, is_cross: bool, scale, heads: int):
        """The optimisation pass (token-major q / k / v [2 B, N | M, heads*64]: reference rows, then edit rows): _EditLayerBatch with every
        edit's running loss and log sums.  None: a precondition of the tail-sum path does not hold — the caller runs the edits one by one."""
        B, subs = self.B, self.subs
        if not (AP.TAIL_SUMS and AP.TOK_OPT) or self.coords_base != (0, 1) or self.coords_edit != (1, 2):
            return None
        N = q.shape[1]
        S = int(math.isqrt(N))
        q_pre = bool(self.q_scaled_hm)
        if q_pre:
            scale = AP.LN2
        if is_cross:
            for s in subs:
                _ = s.cross_replace_alpha[self.cur_step]
        cs = [s._tables(S, heads, q, self.coords[j], 64) for j, s in enumerate(subs)]
        lossy = N >= 32 ** 2
        kind = "cross" if is_cross else "self"
        running, log_accs = None, [None] * B
        if lossy:
            runs = []
            for j, s in enumerate(subs):                       # _GeometryControllerBase.forward's tail-sum bookkeeping, per edit
                log = s.loss_log_dict[kind]
                acc = s.__dict__.get("_log_acc_" + kind)
                if all((not torch.is_tensor(x)) and x == 0.0 for x in log.values()):
                    acc = s.__dict__["_log_acc_" + kind] = ops.zeros_f32(4, q.device)
                    for i, key in enumerate(AP.LOG_KEYS):
                        if key in log:
                            log[key] = acc[i]
                views = acc is not None and all(torch.is_tensor(log[key]) and log[key].data_ptr() == acc[i].data_ptr()
                                                for i, key in enumerate(AP.LOG_KEYS) if key in log)
                lo = s.loss
                if torch.is_tensor(lo):
                    ok = lo.is_cuda and lo.dtype == torch.float32 and lo.numel() == 1
                    lo = lo.reshape(1)
                else:
                    ok, lo = (lo == 0.0), ops.zeros_f32(1, q.device)
                if not (views and ok):
                    return None
                runs.append(lo)
                log_accs[j] = acc
            running = torch.cat(runs)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, new_running = _EditLayerBatch.apply(q, k, v, self, cs, is_cross, float(scale), q_pre, heads, running, log_accs)
        if self.collect_ref:                              # the reference rows of this layer, for the CFG pass of the same step
            ok = q_pre and q.dtype in (torch.float16, torch.bfloat16)
            self._leave_ref((q[:B].detach(), k[:B].detach(), v[:B].detach(), out[:B].detach()) if ok else False)
        if lossy:
            for j, s in enumerate(subs):
                s.loss = new_running[j]
                s.loss_log_dict["num_layers"] += 1
        return out

    def __call__(self, q, k, v, is_cross: bool, place_in_unet: str, transform_coords=None, scale=None, mask=None):
        out = self.forward(q, k, v, is_cross, place_in_unet, transform_coords=transform_coords, scale=scale, mask=mask)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            for s in self.subs:                        # (an inactive self-attention layer above advanced only the layer counter)
                if s.cur_att_layer >= s.num_att_layers:
                    s.cur_att_layer = 0
                    s.cur_step += 1
                    s.between_steps()
        return out

    def step_callback(self, x_t, transform_coords=None):
        return x_t

    # -- what graphs.py / diffusion._unet_nograd ask a controller ----------------------------------------------------------
    def harmonise(self):
        """Pad the edits' row lists (inpaint rows, rows inside the soft edit mask) to the batch's longest bucket per resolution: launch
        dimensions then depend on ONE number per list and resolution instead of B, which keeps the set of captured graphs small.  Padding
        slots are neither computed nor read (the lists carry their true length on the device)."""
        for S in list(self.subs[0].masks_cache_dict):
            cs = [s.masks_cache_dict.get(S) for s in self.subs]
            if any(c is None or "f" not in c for c in cs):
                continue
            for key, pkey in (("rows", ("rows", S, self._is_remover)), ("edit_rows", ("edit_rows", S))):
                if any(key not in c for c in cs):
                    continue
                want = max(int(c[key].numel()) for c in cs)
                for s, c in zip(self.subs, cs):
                    t = c[key]
                    if t.numel() < want and getattr(s, "persistent_tables", False) and (key != "rows" or c.get("n_rows") is not None):
                        pad = t[:1].expand(want - t.numel()) if t.numel() else torch.zeros(want, dtype=t.dtype, device=t.device)
                        c[key] = _persist(s, pkey + (want,), torch.cat([t, pad]).contiguous())

    def graph_key(self):
        for s in self.subs:
            s.cur_step = self.cur_step
            self._sync(s)
        return ("EditBatch", self.B) + self.subs[0].graph_key()[:-2] + (self.rows_identical, self.ref_stash_serial if self.use_ref_stash else None)

    def table_signature(self):
        self.harmonise()
        return tuple(s.table_signature() for s in self.subs)

    def tables_built(self, layers) -> bool:
        return all(s.tables_built(layers) for s in self.subs)

    def prebuild_tables(self, layers, q_like, transform_coords=None):
        for j, s in enumerate(self.subs):
            if not s.tables_built(layers):
                s.prebuild_tables(layers, q_like, self.coords[j])

    def after_graph_replay(self):
        self.cur_att_layer = 0
        self.cur_step += 1
        for s in self.subs:
            s.after_graph_replay()

    def sync_loss_weights(self, dev):
        for s in self.subs:
            s.sync_loss_weights(dev)

    def export_loss_state(self):
        return [(s.loss, {k: (dict(v) if isinstance(v, dict) else v) for k, v in s.loss_log_dict.items()}) for s in self.subs]

    def import_loss_state(self, state):
        for s, (loss, log) in zip(self.subs, state):
            s.loss = loss
            s.loss_log_dict = {k: (dict(v) if isinstance(v, dict) else v) for k, v in log.items()}

    def undo_step(self):
        """The driver's ``controller.cur_step -= 1`` after an optimisation pass (U/editor.py:307)."""
        self.cur_step -= 1
        for s in self.subs:
            s.cur_step -= 1


class GraphedBatchOptPass(graphs.GraphedOptPass):
    """The optimisation pass of an EditBatch (one hipGraph, reused across batches): latents [2 B] = reference rows then edit rows, context
    [2 B, 77, C] = the text rows of the same samples."""
    ctx_text_rows_only = False


# ---------------------------------------------------------------------------------------------------------------------------
_PINNED = []


def _logs_to_host_async(subs):
    """The loss logs of all edits of an optimisation pass in ONE device -> pinned-host copy behind the work queued so far.
    -> (keys per edit, host view, event)."""
    keys, vals = [], []
    for s in subs:
        k, v = E._log_keys_vals(s.loss_log_dict)
        keys.append(k)
        vals.extend(v)
    dev = next((v.device for v in vals if v.is_cuda), None)
    if dev is None:
        return None
    stacked = torch.stack([v.detach().float().to(dev).reshape(()) for v in vals])
    if len(_PINNED) < 4:
        _PINNED.append(torch.empty(1024, dtype=torch.float32).pin_memory())
    buf = _PINNED[0]
    _PINNED.append(_PINNED.pop(0))
    host = buf[:stacked.numel()]
    host.copy_(stacked, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return keys, host, ev


def _finish_logs(subs, logs, handle, i, skip_optim_steps, edit_type, use_adaptive_optimization, removal_loss_value_in, num_ddim_steps, out_logs):
    """U/editor.py:284-306 for every edit: the logged terms on the host and the adaptive weight schedule of THAT edit."""
    if handle is not None:
        keys, host, ev = handle
        ev.synchronize()
        flat = host.tolist()
    o = 0
    for j, s in enumerate(subs):
        if handle is not None:
            n = len(keys[j])
            d = E._log_from_host(logs[j], keys[j], flat[o:o + n])
            o += n
        else:
            d = E.convert_loss_log_to_numpy(logs[j])
        if use_adaptive_optimization:
            fn = adaptive_optimization_step_editing if edit_type == "geometry_editor" else adaptive_optimization_step_remover
            fn(s, i, skip_optim_steps, d, num_ddim_steps=num_ddim_steps, removal_loss_value_in=removal_loss_value_in)
        out_logs[j][i] = d


@torch.no_grad()
def text2image_ldm_stable_batch(model, prompts: Sequence[str], batch: EditBatch, num_inference_steps: int, guidance_scale: float,
                                latent: torch.Tensor, ddim_latents: Sequence[torch.Tensor], masks_obj: Sequence, uncond_embeddings=None,
                                start_time=50, return_type="latents", optimize_steps=0.2, latent_replace=0.2, lr=0.0,
                                optimize_embeddings=False, optimize_latents=False, edit_type="geometry_editor", fast_start_steps=0.0,
                                num_first_optim_steps=1, use_adaptive_optimization=True, removal_loss_value_in=-1.5, image_size=None,
                                skip_optim_steps=None, num_ddim_steps=None):
    """U/editor.py:65-423 for B edits in lockstep.  ``latent`` [B,4,h,w] = every edit's x_T, ``ddim_latents`` = the inversion trajectory
    as T + 1 tensors [B,4,h,w], ``prompts`` one per edit (the reference passes [prompt, prompt] for the two rows of one edit).
    -> (latents [2 B,4,h,w] role-major or decoded images, per-edit loss logs).  Step order, gates and arithmetic follow
    editor.text2image_ldm_stable statement by statement; what is batched is the UNet, what is looped over is per-edit glue."""
    if fast_start_steps or num_first_optim_steps != 1:
        raise NotImplementedError("the batched driver runs the batch drivers' configuration: fast_start_steps = 0, one optimisation iteration per step")
    if uncond_embeddings is not None:
        raise NotImplementedError("null-text embeddings per step are not batched (every reference driver runs with perform_inversion=False)")
    B, subs = batch.B, batch.subs
    skip_optim_steps = E.SKIP_OPTIM_STEPS if skip_optim_steps is None else skip_optim_steps
    num_ddim_steps = E.NUM_DDIM_STEPS if num_ddim_steps is None else num_ddim_steps
    logs_out: List[Dict[int, dict]] = [dict() for _ in range(B)]
    batch.persistent_tables = os.environ.get("GD_PERSISTENT_TABLES", "1") == "1"
    register_attention_control_diffusers(model, batch, None)
    for s in subs:
        s.num_att_layers = batch.num_att_layers
    dev = model.device

    tok = model.tokenizer
    embs = []
    for p in prompts:
        ti = tok([p], padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
        embs.append(encode_text(model, ti.input_ids))
    text = torch.cat(embs)                                                         # [B, 77, C]
    ui = tok([E.UNCOND_TEXT], padding="max_length", max_length=tok.model_max_length, return_tensors="pt", truncation=True)
    ctx_uncond = encode_text(model, ui.input_ids).expand(B, -1, -1).contiguous()   # the uncond_edit rows
    ctx_text = torch.cat([text, text])                                             # cond_ref rows, then cond_edit rows

    latents = torch.cat([latent, latent]).to(dev)                                  # :134-139 (expand to the two rows of every edit)
    model.scheduler.set_timesteps(num_inference_steps)
    for p in model.unet.parameters():
        p.requires_grad = False
    timesteps = model.scheduler.timesteps[-start_time:]
    T = len(timesteps)

    upd_masks = []
    for j, s in enumerate(subs):                                                   # :147-149 (512^2 mask warp, once per edit)
        t_m = reshape_transform_coords(batch.coords[j].to(dev).float(), in_mat_shape=s.image_mask.shape)
        t_m = t_m.tile(s.image_mask.shape[0], 1, 1, 1).to(E._coords_dtype(text))
        s.mask_new_warped = binarize_tensor(warp_grid_edit(s.image_mask[:, None].to(dev).float(), t_m)).type_as(text)
        # the mask of _update_latent (U/optimization.py:228: resized, NOT re-binarised), constant over the steps
        m = s.mask_new_warped[:1]
        m = reshape_attention_mask(m[None, None].to(dev).float().reshape(1, 1, *m.shape[-2:]), in_mat_shape=latents[-1:].shape)
        upd_masks.append(m[-1, 0].reshape(-1).contiguous())

    ref_from_opt = E.REF_FROM_OPT and E.ref_from_opt_supported() and MERGED

    def cfg_pass(lat, ctx_t, tt):
        assert not torch.is_grad_enabled()
        if ref_from_opt and batch.ref_stash_serial is not None and batch.ref_stash_t == int(tt):
            # a step with an optimisation pass: every edit's reference row is already there, layer by layer (ref_stash)
            E.REF_FROM_OPT_PASSES += 1
            set_attn_processor_for_edit(model, coords_base=(1, 1), coords_edit=(1, 2), use_cfg=True, n_batch=2)
            batch.use_ref_stash, batch._ref_pos = True, 0
            try:
                lat_in = torch.cat([lat[B:], lat[B:]])                                 # [uncond_edit | cond_edit] x B
                ctx2 = torch.cat([ctx_uncond.to(ctx_t.dtype), ctx_t[B:]])
                eps = _unet_nograd(model, batch, lat_in, tt, ctx2, "cfg2s", None, ctx_src=ctx_t)
            finally:
                batch.use_ref_stash, batch.ref_stash_t = False, None
            edit_out = _sched_step(model.scheduler, eps[:B], tt, lat[B:], eps[B:], guidance_scale)
            warp_utils.SPLATTER.clear_cache()
            return torch.cat([lat[:B].to(edit_out.dtype), edit_out])
        set_attn_processor_for_edit(model, coords_base=(1, 2), coords_edit=(2, 3), use_cfg=True, n_batch=3)
        lat_in = torch.cat([lat[B:], lat[:B], lat[B:]])                            # [uncond_edit | cond_ref | cond_edit] x B (U/diffusion.py:43 less uncond_ref)
        ctx3 = torch.cat([ctx_uncond.to(ctx_t.dtype), ctx_t])
        eps = _unet_nograd(model, batch, lat_in, tt, ctx3, "cfg3", None, ctx_src=ctx_t)
        edit_out = _sched_step(model.scheduler, eps[:B], tt, lat[B:], eps[2 * B:], guidance_scale)
        warp_utils.SPLATTER.clear_cache()
        return torch.cat([lat[:B].to(edit_out.dtype), edit_out])

    opt_pass = GraphedBatchOptPass(model, None, guidance_scale)
    remover = type(subs[0]).__name__ == "AttentionGeometryRemover"
    for i, t in enumerate(timesteps):
        for s in subs:
            E.clear_controller_loss(s)
        if (i < optimize_steps * T) and (i % skip_optim_steps == 0):                                    # :181
            l_eff = lr * (50 - i) * skip_optim_steps * (50 / (num_ddim_steps + 1e-8))                  # :207
            set_attn_processor_for_edit(model, coords_base=(0, 1), coords_edit=(1, 2), use_cfg=False)   # :213
            batch.collect_ref = ref_from_opt
            n0 = [ops.sumsq(latents[B + j].detach().float().contiguous()) for j in range(B)]            # orig_norm^2 (:219)
            for j, s in enumerate(subs):
                s.rows_identical = bool(E.TIE_IDENTICAL_ROWS and i == 0 and remover and torch.equal(latents[j], latents[B + j])
                                        and torch.equal(ctx_text[j], ctx_text[B + j]))
            g_lat, g_ctx, lat_in, ctx_in = opt_pass.grads(batch, latents, ctx_text, t)                  # :218-273
            new_rows = [ops.masked_latent_update(lat_in[B + j].detach().float().contiguous(), g_lat[B + j].detach().float().contiguous(),
                                                 upd_masks[j], float(l_eff)) for j in range(B)]         # U/optimization.py:230-245
            lat_upd = torch.cat([lat_in[:B].detach(), torch.stack(new_rows).to(lat_in.dtype)])
            g_c = torch.nan_to_num(g_ctx, posinf=0.0, neginf=0.0, nan=0.0)
            ctx_upd = torch.cat([ctx_in[:B].detach(), ctx_in[B:].detach() - l_eff * g_c[B:]])
            logs = [s.loss_log_dict for s in subs]
            handle = _logs_to_host_async(subs) if E.LATE_LOSS_SYNC else None
            if not E.LATE_LOSS_SYNC:
                _finish_logs(subs, logs, None, i, skip_optim_steps, edit_type, use_adaptive_optimization, removal_loss_value_in, num_ddim_steps, logs_out)
            for s in subs:
                E.clear_controller_loss(s)
                s.rows_identical = False
            batch.undo_step()                                                                           # :307
            if optimize_latents:                                                                        # :312-316
                latents = lat_upd.detach()
                rows = []
                for j in range(B):
                    last = latents[B + j].float().contiguous()
                    rows.append(ops.norm_rescale(last, n0[j], ops.sumsq(last)))
                latents = torch.cat([latents[:B], torch.stack(rows).to(latents.dtype)])
            if optimize_embeddings:                                                                     # :319-322
                ctx_text = ctx_upd.detach()
            latents = cfg_pass(latents, ctx_text, t)                                                    # :343-351
            if E.LATE_LOSS_SYNC:
                _finish_logs(subs, logs, handle, i, skip_optim_steps, edit_type, use_adaptive_optimization, removal_loss_value_in, num_ddim_steps, logs_out)
        else:
            latents = cfg_pass(latents, ctx_text, t)                                                    # :366-368

        i_n = len(ddim_latents) - 2 - i                                                                 # :375-377
        latents = torch.cat([ddim_latents[i_n].type_as(latents.detach()), latents[B:].detach()])

        if not remover and i < T * latent_replace:                                                      # :382-399 latent warp
            s_ = latents.shape[-1]
            rows = []
            for j, s in enumerate(subs):
                if masks_obj[j] is None:
                    rows.append(latents[B + j])
                    continue
                t_c = reshape_transform_coords(batch.coords[j].to(dev).float(), in_mat_shape=latents[:1].shape).to(E._coords_dtype(latents))
                i_mask = ((E._resize_mask(s.mask_new_warped[:1].detach().float(), s_) > 0.5) * 1.0).type_as(latents)
                warped = warp_grid_edit(latents[j:j + 1].detach().clone(), t_c)
                rows.append((latents[B + j:B + j + 1] * (1 - i_mask) + i_mask * warped.type_as(latents))[0])
            latents = torch.cat([latents[:B], torch.stack(rows)])

    if return_type == "image":
        return latent2image(model.vae, latents), logs_out
    return latents, logs_out


# ---------------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def ddim_inversion_batch(model, images: Sequence[np.ndarray], prompts: Sequence[str], num_ddim_steps: int, guidance_scale: float, device):
    """NullInversion.invert (U/inversion.py:131-196,261-277; perform_inversion=False) for B images at once: VAE encode at batch B, then
    the 50 inversion passes at batch B (prompt == unconditional text: the two CFG rows of an edit are one sample, inversion.py
    SINGLE_ROW) or 2 B.  -> list over time of [B,4,h,w] latents (index 0 = the encoded images, -1 = x_T)."""
    ni = NullInversion(model, num_ddim_steps=num_ddim_steps, uncond_text=E.UNCOND_TEXT, device=device, progress_bar=None, guidance_scale=guidance_scale)
    x = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(im) for im in images]))).to(device).float() / 127.5 - 1
    latent = model.vae.encode(x.permute(0, 3, 1, 2))["latent_dist"].mean * model.vae.config.scaling_factor
    B = latent.shape[0]
    tok = model.tokenizer
    un = encode_text(model, tok([E.UNCOND_TEXT], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids)
    cond = torch.cat([encode_text(model, tok([p], padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids)
                      for p in prompts])
    single = all(bool(torch.equal(cond[j], un[0])) for j in range(B))
    ctx = cond.contiguous() if single else torch.cat([un.expand(B, -1, -1), cond]).contiguous()
    model.unet.set_attn_processor(VanillaAttentionProcessor())
    inv = ni._inverse
    inv.set_timesteps(num_ddim_steps, device=device)
    traj = [latent]
    latents = latent.clone().detach()
    runner = model.__dict__.get("_graphed")
    if runner is None:
        runner = model.__dict__["_graphed"] = graphs.GraphedUNet(model.unet)
    for t in inv.timesteps:
        x_in = latents if single else torch.cat([latents] * 2)
        if graphs.ENABLED:
            eps, _ = runner(("inversion",), x_in, t, ctx, ctx_src=ctx)
        else:
            eps = model.unet(x_in, t, encoder_hidden_states=ctx, return_dict=False)[0]
        if single:
            latents = inv.step(eps, t, latents, return_dict=False)[0]
        else:
            e_u, e_c = eps.chunk(2)
            latents = inv.step(e_u, t, latents, eps_cond=e_c, guidance_scale=guidance_scale, return_dict=False)[0]
        traj.append(latents.detach())
    return traj


def perform_geometric_edit_batch(edits: Sequence[dict], ldm_stable_model=None, tokenizer_model=None, scheduler_in=None,
                                 cross_replace_steps={"default_": 0.95}, self_replace_steps=0.95, optimize_steps=0.6, lr=0.03,
                                 latent_replace=0.6, optimize_embeddings=True, optimize_latents=True, obj_edit_step=1.0,
                                 perform_inversion=False, guidance_scale=7.5, skip_optim_steps=1, num_ddim_steps=50, splatting_radius=1.3,
                                 edit_type="geometry_editor", loss_weights_dict=None, return_loss_log_dict=False, splatting_tau=1.0,
                                 splatting_points_per_pixel=15, use_adaptive_optimization=True, removal_loss_value_in=-1.5,
                                 return_latents=False, **ignored):
    """``editor.perform_geometric_edit`` (U/editor.py:428-710) for a LIST of edits that share one configuration (what the batch driver
    does: one column of settings for every folder, large_scale_editor.py:196-317).  ``edits``: dicts with image, depth, image_mask,
    transform_in and optionally prompt.  -> one result per edit, each what perform_geometric_edit returns for it (images[, loss log]
    [, final latents [2,4,h,w]]).  The keyword arguments mean what they mean there; ``loss_weights_dict`` is deep-copied per edit (every
    edit's adaptive schedule edits its own)."""
    import copy
    # the rest of perform_geometric_edit's signature: accepted where it does not change the result, refused where it would
    allowed = dict(image_stitch=None, progress=None, fast_start_steps=0.0, num_first_optim_steps=1, return_attention_maps=False, unet_path="",
                   use_optimizer=True)
    for k_, v_ in ignored.items():
        if k_ not in allowed:
            raise TypeError(f"perform_geometric_edit_batch() got an unexpected keyword argument {k_!r}")
        if k_ in ("fast_start_steps", "return_attention_maps", "image_stitch", "unet_path") and v_ != allowed[k_] and v_:
            raise NotImplementedError(f"perform_geometric_edit_batch: {k_}={v_!r} is only supported by the one-edit driver (perform_geometric_edit)")
    if perform_inversion:
        raise NotImplementedError("null-text optimisation is per edit: use perform_geometric_edit (every reference driver passes perform_inversion=False)")
    if edit_type not in ("geometry_editor", "geometry_remover"):
        raise NotImplementedError(edit_type)
    if len(edits) == 0:
        return []
    if len(edits) > MAX_EDITS:
        # a launch carries one row-list / pair segment per edit (GD_ATTN_MAX_ROWLIST_SEGS): longer lists run MAX_EDITS at a time
        kw = dict(ldm_stable_model=ldm_stable_model, tokenizer_model=tokenizer_model, scheduler_in=scheduler_in, cross_replace_steps=cross_replace_steps,
                  self_replace_steps=self_replace_steps, optimize_steps=optimize_steps, lr=lr, latent_replace=latent_replace,
                  optimize_embeddings=optimize_embeddings, optimize_latents=optimize_latents, obj_edit_step=obj_edit_step,
                  perform_inversion=perform_inversion, guidance_scale=guidance_scale, skip_optim_steps=skip_optim_steps, num_ddim_steps=num_ddim_steps,
                  splatting_radius=splatting_radius, edit_type=edit_type, loss_weights_dict=loss_weights_dict, return_loss_log_dict=return_loss_log_dict,
                  splatting_tau=splatting_tau, splatting_points_per_pixel=splatting_points_per_pixel,
                  use_adaptive_optimization=use_adaptive_optimization, removal_loss_value_in=removal_loss_value_in, return_latents=return_latents, **ignored)
        out = []
        for i in range(0, len(edits), MAX_EDITS):
            out += perform_geometric_edit_batch(edits[i:i + MAX_EDITS], **kw)
        return out
    prev_grad = torch.is_grad_enabled()
    torch.set_grad_enabled(False)
    import time
    t_last = [time.perf_counter()]

    def _tm(what):                      # GD_BATCH_TIMING=1: host-side section times (development; synchronises)
        if os.environ.get("GD_BATCH_TIMING") == "1":
            h = time.perf_counter() - t_last[0]
            torch.cuda.synchronize()
            print(f"[batch] {what}: host {1e3 * h:.1f} ms, with the device {1e3 * (time.perf_counter() - t_last[0]):.1f} ms", flush=True)
            t_last[0] = time.perf_counter()
    try:
        torch.manual_seed(E.SEED)
        torch.cuda.manual_seed_all(E.SEED)
        max_opt = max(self_replace_steps, cross_replace_steps["default_"])
        optimize_steps = min(optimize_steps, max_opt)
        warp_utils.SPLATTER.clear_cache()
        E.GUIDANCE_SCALE, E.SKIP_OPTIM_STEPS, E.NUM_DDIM_STEPS = guidance_scale, skip_optim_steps, num_ddim_steps
        if scheduler_in is not None:
            model, tokenizer = ldm_stable_model, tokenizer_model
        else:
            from .diffusion import load_model
            model, tokenizer, _ = load_model(diffusion_model=E.DIFFUSION_MODEL, device=E.DEVICE)
        dev = E.DEVICE
        B = len(edits)
        shapes = {tuple(np.asarray(e["image"]).shape) for e in edits}
        if len(shapes) != 1:
            raise ValueError(f"perform_geometric_edit_batch: the edits of a batch must share one image size (got {sorted(shapes)}); group them by size")
        prompts = [e.get("prompt", "") for e in edits]
        images = [np.asarray(e["image"]) for e in edits]
        cls = AttentionGeometryEdit if edit_type == "geometry_editor" else AttentionGeometryRemover

        def prepass():
            # the geometry pre-passes and the controllers (host work with device round trips, independent of the inversion) run beside
            # the inversion: on a worker thread and a side stream (editor.start_ahead)
            subs, coords, masks, dev_in = [], [], [], []
            for e, image in zip(edits, images):
                image_mask = torch.as_tensor(np.asarray(e["image_mask"])).float()
                H = image.shape[0]
                t_coords_depth, _, amodal = vis_utils.get_transform_coordinates(
                    image, e["depth"], image_mask.numpy(), transform_in=e["transform_in"],
                    focal_length=550 * H / 512.0 if H != 512 else 550, return_mesh=True, device=str(dev), as_torch=True, preview=False)
                c = cls([e.get("prompt", "")] * 2, num_ddim_steps, cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps,
                        equalizer=None, local_blend=None, controller=None, image_mask=image_mask.numpy(), empty_scale=0.0, use_all=False,
                        obj_edit_step=obj_edit_step, tokenizer=tokenizer, device=dev, mode=E.MODE)
                c.amodal_mask = torch_erode(amodal.float())
                if loss_weights_dict is not None:
                    lw = copy.deepcopy(loss_weights_dict)
                    c.loss_weight_dict = lw
                    c.default_loss_weights = lw
                subs.append(c); coords.append(t_coords_depth[None].detach()); masks.append(image_mask)
                dev_in.append((torch.from_numpy(np.ascontiguousarray(image)).to(dev), image_mask.to(dev)))      # for the post-process
            return subs, coords, masks, dev_in, EditBatch(subs, coords)

        ahead = E.start_ahead(prepass)
        try:
            traj = ddim_inversion_batch(model, images, prompts, num_ddim_steps, guidance_scale, dev)
        except BaseException:
            E._drain_ahead(ahead)           # the inversion's exception is the one to report; the pre-pass's, if any, is logged
            raise
        subs, coords, masks, dev_in, batch = ahead.result()
        _tm("inversion + pre-pass + controllers")
        out, logs = text2image_ldm_stable_batch(
            model, prompts, batch, num_ddim_steps, guidance_scale, latent=traj[-1], ddim_latents=traj, masks_obj=[m[None, None] for m in masks],
            optimize_steps=optimize_steps, latent_replace=latent_replace, lr=lr, optimize_embeddings=optimize_embeddings,
            optimize_latents=optimize_latents, edit_type=edit_type, use_adaptive_optimization=use_adaptive_optimization,
            removal_loss_value_in=removal_loss_value_in, return_type="latents", image_size=images[0].shape[0],
            skip_optim_steps=skip_optim_steps, num_ddim_steps=num_ddim_steps)
        _tm("edit loop")
        decoded = latent2image(model.vae, out, as_tensor=True)                     # [2 B, H, W, 3] uint8 on the device, role-major
        # every edit's post-process queued first, then ONE download per kind (a download per edit is a synchronisation per edit)
        edited = torch.stack([E.post_process(dev_in[j][0], dev_in[j][1], decoded[B + j], coords[j], subs[j].mask_new_warped, edit_type, as_numpy=False)
                              for j in range(B)]).cpu().numpy()
        recon = decoded[:B].cpu().numpy()
        results = []
        for j, e in enumerate(edits):
            imgs = [recon[j], edited[j]]
            ret = [imgs]
            if return_loss_log_dict:
                ret.append(logs[j])
            if return_latents:
                ret.append(torch.stack([out[j], out[B + j]]))
            results.append(ret[0] if len(ret) == 1 else tuple(ret))
        model.unet.set_attn_processor(VanillaAttentionProcessor())
        _tm("decode + post-process")
        return results
    finally:
        torch.set_grad_enabled(prev_grad)
