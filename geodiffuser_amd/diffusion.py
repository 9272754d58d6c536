"""One denoising step + VAE glue.  Mirror of GeoDiffuser/utils/diffusion.py:39-97 (same names / argument meaning).

The reference wraps these in ``torch.autocast("cuda", fp16)``; here the UNet holds 16-bit weights and casts its inputs,
which gives the same dtypes at the attention processors without an autocast context.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import graphs, warp_utils
from .pipeline import build_random_sd21
from .scheduler import DDIMScheduler


def _unet_nograd(model, controller, x, t, ctx, tag, transform_coords=None, ctx_src=None, learn=True):
    """UNet call of a no-grad pass, through a captured hipGraph when the controller's launch sequence is static.

    A replay runs no Python, and the captured kernels read the controller's per-resolution tables (masks, splat idx / w, inpaint
    rows) by ADDRESS from persistent buffers that outlive an edit.  So before any replay the CURRENT controller must have written
    its tables into those buffers: the (resolution, heads) pairs of the hooked layers are learnt from the first eager pass on this
    model / latent size, and a controller that has not built them yet (a new edit whose first UNet pass is a CFG pass:
    optimize_steps == 0, or fast_start_steps > 0) builds them up front.  The table shapes are part of the graph key."""
    key_fn = getattr(controller, "graph_key", None)

    def eager():
        # (per-row timesteps arrive as a tuple; an eager pass leaves no reference rows a captured pass may read: see GraphedUNet)
        tt = torch.tensor([int(v) for v in t], device=x.device, dtype=torch.long) if isinstance(t, tuple) else t
        collect = getattr(controller, "collect_ahead", False)
        if collect:
            controller._ahead = None
        out = model.unet(x, tt, encoder_hidden_states=ctx)["sample"]
        if collect:
            st = controller._ahead
            ok = st is not None and len(st) == controller.num_att_layers and all(isinstance(a, tuple) for a in st)
            controller.ref_stash, controller.ref_stash_serial, controller._ahead = (st if ok else None), None, None
        return out

    if not graphs.ENABLED or torch.is_grad_enabled() or key_fn is None or getattr(controller, "store_attention_maps", False) \
            or not getattr(controller, "persistent_tables", False):
        return eager()
    if not learn:
        # a pass that reads none of the controller's per-resolution tables (the batched reference rows: vanilla attention for every row): no
        # table has to exist before a replay and the table shapes are not part of its key
        runner = model.__dict__.get("_graphed")
        if runner is None:
            runner = model.__dict__["_graphed"] = graphs.GraphedUNet(model.unet)
        out, replayed = runner((tag,) + key_fn(), x, t, ctx, ctx_src=ctx_src, controller=controller)
        if replayed:
            controller.after_graph_replay()
        return out
    seen = model.__dict__.setdefault("_cfg_layers", {})
    # the hooked (resolution, heads, head dim) set is learnt per (latent size, controller type); a model whose hooks changed since is
    # caught by the `learnt() != layers` comparison below (the controller then holds a table the list does not know) and learns it again
    lk = tuple(x.shape[2:]) + (type(controller).__name__,)
    layers = seen.get(lk)

    def learnt():
        return sorted((S, c["f"], c["D"]) for S, c in controller.masks_cache_dict.items() if "f" in c)

    if layers is None:                                    # first hooked no-grad pass at this latent size: eager, learn the layers
        out = eager()
        seen[lk] = learnt()
        return out
    if not controller.tables_built(layers):
        q_like = torch.empty(1, device=x.device, dtype=model.unet.dtype)
        controller.prebuild_tables(layers, q_like, transform_coords)
    if learnt() != layers:
        # the controller holds a table the learnt list does not know (the first pass did not reach every hooked resolution, or the hooks
        # changed on this model object): a capture would build it lazily, with host syncs, inside torch.cuda.graph — run eagerly and relearn
        out = eager()
        seen[lk] = learnt()
        return out
    runner = model.__dict__.get("_graphed")
    if runner is None:
        runner = model.__dict__["_graphed"] = graphs.GraphedUNet(model.unet)
    out, replayed = runner((tag,) + key_fn() + (controller.table_signature(),), x, t, ctx, ctx_src=ctx_src, controller=controller)
    if replayed:
        controller.after_graph_replay()
    return out


_TEXT_CACHE = {}       # id(text_encoder) -> (weakref, weights version, {(token ids) -> embeddings})
TEXT_CACHE = os.environ.get("GD_TEXT_CACHE", "1") == "1"


@torch.no_grad()
def encode_text(model, input_ids: torch.Tensor) -> torch.Tensor:
    """``model.text_encoder(input_ids)[0]`` (U/editor.py:116-121, U/inversion.py:213-224 call it four times per edit) with the result kept per
    token-id tuple: the embedding is a pure function of the ids and the encoder's weights, the batch driver passes the same prompt ("")
    for every edit, and a 23-layer text tower at batch 1 is ~300 eagerly dispatched launches during which the GPU idles at the start of
    every edit.  The cache dies with the encoder and is dropped when any of its parameters is modified in place (version counters).
    Callers get a fresh copy (nothing downstream writes into it, but a cached tensor must not depend on that).  GD_TEXT_CACHE=0: off."""
    enc = model.text_encoder
    ids = input_ids.to(model.device)
    if not TEXT_CACHE or not isinstance(enc, torch.nn.Module):
        return enc(ids)[0]
    import weakref
    params = list(enc.parameters())
    ver = (tuple(p._version for p in params), params[0].dtype if params else None, str(model.device))
    ent = _TEXT_CACHE.get(id(enc))
    if ent is None or ent[0]() is not enc or ent[1] != ver:
        ent = _TEXT_CACHE[id(enc)] = (weakref.ref(enc), ver, {})
        for k in [k for k, e in _TEXT_CACHE.items() if e[0]() is None]:
            del _TEXT_CACHE[k]
    key = tuple(input_ids.reshape(-1).tolist()) + tuple(input_ids.shape)
    emb = ent[2].get(key)
    if emb is None:
        if len(ent[2]) >= 64:
            ent[2].clear()
        emb = ent[2][key] = enc(ids)[0].detach()
    return emb.clone()


def _sched_step(scheduler, eps_uncond, t, sample, eps_cond, guidance_scale):
    """scheduler.step with the classifier-free-guidance combine.  The repo's own schedulers fuse the combine into the DDIM kernel
    (``eps_cond`` / ``guidance_scale`` keywords of gd_ddim_step); any other scheduler object (e.g. a diffusers ``DDIMScheduler`` a
    caller passes as ``scheduler_in``) gets the reference's explicit combine (diffusion.py:47-49) and its plain ``step`` signature."""
    if isinstance(scheduler, DDIMScheduler):
        return scheduler.step(eps_uncond, t, sample, eta=0.0, eps_cond=eps_cond, guidance_scale=guidance_scale)["prev_sample"]
    eps = eps_uncond + guidance_scale * (eps_cond - eps_uncond)
    return scheduler.step(eps, t, sample, eta=0.0)["prev_sample"]


def diffusion_step(model, controller, latents, context, t, guidance_scale, low_resource=False, transform_coords=None,
                   use_cfg=True, return_noise=False, skip_uncond_ref=False, skip_scheduler=False, ref_from_stash=False, ref_ahead=None):
    """diffusion.py:39-59: UNet -> (CFG combine) -> scheduler.step(eta=0) -> controller.step_callback.
    The CFG combine is fused into the DDIM kernel (gd_ddim_step) unless the caller asks for the combined noise."""
    if use_cfg and skip_uncond_ref and ref_ahead is not None:
        # The 3-row pass below with the NEXT step's reference sample riding along as row 0 (editor.REF_AHEAD): ref_ahead = (its latent — the
        # inversion trajectory's entry for the next timestep —, that timestep).  Rows are independent inside the UNet (per-sample norms,
        # per-sample attention) and row 0 is a vanilla row for the controller, which keeps its per-layer q / k / v and attention output
        # (collect_ahead); its noise prediction is not used.  Batch [ref_next, uncond_edit, cond_ref, cond_edit] with per-row timesteps:
        # the layout of the reference's own 4-row CFG batch (coords_base (2, 3), coords_edit (3, 4)) with row 0 put to use.
        ref_next, t_next = ref_ahead
        latents_input = torch.cat([ref_next.to(latents.dtype), latents[1:2], latents[0:1], latents[1:2]])
        ctx4 = torch.cat([context[2:3], context[1:2], context[2:3], context[3:4]])
        noise_pred = _unet_nograd(model, controller, latents_input, (int(t_next), int(t), int(t), int(t)), ctx4, "cfg4n", transform_coords,
                                  ctx_src=context)
        edit_out = _sched_step(model.scheduler, noise_pred[1:2], t, latents[1:2], noise_pred[3:4], guidance_scale)
        latents_out = torch.cat([latents[0:1].to(edit_out.dtype), edit_out])
        noise_pred_out = None
    elif use_cfg and skip_uncond_ref and ref_from_stash:
        # the `cond_ref` row's per-layer q / k / v and attention outputs were left by the optimisation pass of this step (same latent,
        # timestep and text row: attention_processors.ref_stash).  Batch [uncond_edit, cond_edit].
        latents_input = torch.cat([latents[1:2], latents[1:2]])
        ctx2 = torch.cat([context[1:2], context[3:4]])
        noise_pred = _unet_nograd(model, controller, latents_input, t, ctx2, "cfg2s", transform_coords, ctx_src=context)
        edit_out = _sched_step(model.scheduler, noise_pred[0:1], t, latents[1:2], noise_pred[1:2], guidance_scale)
        latents_out = torch.cat([latents[0:1].to(edit_out.dtype), edit_out])
        noise_pred_out = None
    elif use_cfg and skip_uncond_ref:
        # The reference latent is overwritten by the inversion trajectory after every step (editor.py:375-377), so the
        # reference rows' noise prediction is never used; `cond_ref` is still needed for its per-layer q/k/v, `uncond_ref`
        # is not (vanilla attention, per-sample norms => no influence on other rows).  Batch [uncond_edit, cond_ref, cond_edit].
        latents_input = torch.cat([latents[1:2], latents[0:1], latents[1:2]])
        ctx3 = torch.cat([context[1:2], context[2:3], context[3:4]])
        noise_pred = _unet_nograd(model, controller, latents_input, t, ctx3, "cfg3", transform_coords, ctx_src=context)
        edit_out = _sched_step(model.scheduler, noise_pred[0:1], t, latents[1:2], noise_pred[2:3], guidance_scale)
        latents_out = torch.cat([latents[0:1].to(edit_out.dtype), edit_out])
        noise_pred_out = None
    elif use_cfg:
        latents_input = torch.cat([latents] * 2)
        noise_pred = _unet_nograd(model, controller, latents_input, t, context, "cfg4", transform_coords, ctx_src=context)
        noise_pred_uncond, noise_prediction_text = noise_pred.chunk(2)
        if return_noise:
            noise_pred_out = noise_pred_uncond + guidance_scale * (noise_prediction_text - noise_pred_uncond)
            latents_out = model.scheduler.step(noise_pred_out, t, latents, eta=0.0)["prev_sample"]
        else:
            noise_pred_out = None
            latents_out = _sched_step(model.scheduler, noise_pred_uncond, t, latents, noise_prediction_text, guidance_scale)
    else:
        noise_pred_out = model.unet(latents, t, encoder_hidden_states=context)["sample"]
        if skip_scheduler:      # the optimisation pass discards x_{t-1} (editor.py:253); a captured pass cannot read t on the host
            latents_out = latents.detach()
        else:
            latents_out = model.scheduler.step(noise_pred_out.detach(), t, latents.detach(), eta=0.0)["prev_sample"]
    latents_out = controller.step_callback(latents_out, transform_coords)
    warp_utils.SPLATTER.clear_cache()                                   # diffusion.py:54
    if return_noise:
        return latents_out, noise_pred_out
    return latents_out


@torch.no_grad()
def latent2image(vae, latents, as_tensor: bool = False):
    """diffusion.py:61-68.  ``as_tensor``: the uint8 images stay on the device ([n,H,W,3] tensor; same float32 arithmetic and the same
    truncation as numpy's ``astype``) for the post-process that follows."""
    latents = 1 / 0.18215 * latents
    image = vae.decode(latents)["sample"]
    image = (image.float() / 2 + 0.5).clamp(0, 1)
    if as_tensor:
        return (image.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()
    image = image.cpu().permute(0, 2, 3, 1).numpy()
    return (image * 255).astype(np.uint8)


@torch.no_grad()
def image2latent(image, model, mask=None, device="cuda:0"):
    """diffusion.py:71-97 (mask=None branch; the masked variants are only used by the stitch editors, which do not exist)."""
    if type(image) is torch.Tensor and image.dim() == 4:
        return image
    # upload the uint8 image and do the arithmetic on the device (same float32 operations; a 786k-element CPU op wakes torch's whole
    # intra-op thread pool on a many-core host, which cost up to 200 ms per edit)
    image = torch.from_numpy(np.ascontiguousarray(image)).to(device).float() / 127.5 - 1
    image = image.permute(2, 0, 1).unsqueeze(0)
    latents = model.vae.encode(image)["latent_dist"].mean
    return latents * 0.18215


def load_model(diffusion_model="stabilityai/stable-diffusion-2-1-base", unet_path="", device="cuda:0", random_init=None,
               dtype=torch.float16, tiny=False, prediction_type=None):
    """diffusion.py:99-149.  ``prediction_type``: "epsilon" / "v_prediction"; default: the checkpoint's scheduler config when a
    pipeline is loaded ("stabilityai/stable-diffusion-2-1" at 768^2 is a v-prediction model, BASELINE configs[3]), else "epsilon".  With diffusers + weights available this would wrap ``StableDiffusionPipeline.from_pretrained``;
    in this environment neither exists (no network), so a seeded random-init model of the same shape is built.
    A local directory of the diffusers layout (``unet/``, ``vae/``, ``text_encoder/`` with safetensors) is loaded into this package's own
    modules (geodiffuser_amd/checkpoint.py); GD_USE_DIFFUSERS=1 prefers diffusers' modules where diffusers is installed.
    Returns (ldm_stable, tokenizer, scheduler) like the reference."""
    if os.environ.get("GD_MIOPEN_CACHE", "1") == "1":
        # before the first convolution of the process: committed find-db + no naive solvers in MIOpen's per-shape search (miopen_cache.py)
        from . import miopen_cache
        miopen_cache.configure()
    # A checkpoint DIRECTORY of the diffusers layout on local disk: its weights into THIS package's modules (checkpoint.from_safetensors), so
    # that the edit runs on the harness the benchmark measures — NHWC convolutions, fused norms, captured passes — with real weights.
    local = unet_path or diffusion_model
    if isinstance(local, str) and os.path.isdir(os.path.join(local, "unet")) and not random_init and os.environ.get("GD_USE_DIFFUSERS", "0") != "1":
        from .checkpoint import from_safetensors
        pipe = from_safetensors(local, device=device, dtype=dtype)
        if prediction_type not in (None, pipe.scheduler.config.prediction_type):
            pipe.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                           set_alpha_to_one=False, prediction_type=prediction_type)
        return pipe, pipe.tokenizer, pipe.scheduler
    try:  # pragma: no cover
        import diffusers  # noqa: F401
        have_diffusers = True
    except Exception:  # noqa: BLE001
        have_diffusers = False
    if have_diffusers and not random_init:  # pragma: no cover - not reachable in the build image
        from diffusers import StableDiffusionPipeline as _SDP
        from .attention_processors import VanillaAttentionProcessor
        pipe = _SDP.from_pretrained(unet_path or diffusion_model, torch_dtype=dtype).to(device)
        # the repo's own scheduler (same betas / alphas / 'leading' timestep table as the diffusers object the reference builds at
        # diffusion.py:110): its step runs the fused CFG + DDIM kernel and accepts the keywords diffusion_step passes
        if prediction_type is None:
            prediction_type = getattr(getattr(pipe.scheduler, "config", None), "prediction_type", "epsilon")
        pipe.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                       set_alpha_to_one=False, prediction_type=prediction_type)
        pipe.unet.set_attn_processor(VanillaAttentionProcessor())
        pipe.unet.eval()
        return pipe, pipe.tokenizer, pipe.scheduler
    if "xl" in str(diffusion_model).lower():                       # SDXL-base shape (BASELINE configs[4]); 1024^2 micro-conditioning
        from .pipeline import build_random_sdxl
        pipe = build_random_sdxl(device=device, dtype=dtype, tiny=tiny)
    else:
        name = str(diffusion_model).lower()
        sd14 = "v1-" in name or "v1." in name or "sd14" in name            # e.g. "CompVis/stable-diffusion-v1-4", the reference's default
        pipe = build_random_sd21(device=device, dtype=dtype, tiny=tiny, sd14=sd14)
    if prediction_type not in (None, "epsilon"):
        pipe.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                       set_alpha_to_one=False, prediction_type=prediction_type)
    return pipe, pipe.tokenizer, pipe.scheduler
