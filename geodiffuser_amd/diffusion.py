"""One denoising step + VAE glue.  Mirror of GeoDiffuser/utils/diffusion.py:39-97 (same names / argument meaning).

The reference wraps these in ``torch.autocast("cuda", fp16)``; here the UNet holds 16-bit weights and casts its inputs,
which gives the same dtypes at the attention processors without an autocast context.
"""
from __future__ import annotations

import numpy as np
import torch

from . import graphs, warp_utils
from .pipeline import build_random_sd21
from .scheduler import DDIMScheduler


def _unet_nograd(model, controller, x, t, ctx, tag):
    """UNet call of a no-grad pass, through a captured hipGraph when the controller's launch sequence is static."""
    key_fn = getattr(controller, "graph_key", None)
    if not graphs.ENABLED or torch.is_grad_enabled() or key_fn is None or getattr(controller, "store_attention_maps", False) \
            or not getattr(controller, "persistent_tables", False):
        return model.unet(x, t, encoder_hidden_states=ctx)["sample"]
    runner = model.__dict__.get("_graphed")
    if runner is None:
        runner = model.__dict__["_graphed"] = graphs.GraphedUNet(model.unet)
    out, replayed = runner((tag,) + key_fn(), x, t, ctx)
    if replayed:
        controller.after_graph_replay()
    return out


def diffusion_step(model, controller, latents, context, t, guidance_scale, low_resource=False, transform_coords=None,
                   use_cfg=True, return_noise=False, skip_uncond_ref=False, skip_scheduler=False):
    """diffusion.py:39-59: UNet -> (CFG combine) -> scheduler.step(eta=0) -> controller.step_callback.
    The CFG combine is fused into the DDIM kernel (gd_ddim_step) unless the caller asks for the combined noise."""
    if use_cfg and skip_uncond_ref:
        # The reference latent is overwritten by the inversion trajectory after every step (editor.py:375-377), so the
        # reference rows' noise prediction is never used; `cond_ref` is still needed for its per-layer q/k/v, `uncond_ref`
        # is not (vanilla attention, per-sample norms => no influence on other rows).  Batch [uncond_edit, cond_ref, cond_edit].
        latents_input = torch.cat([latents[1:2], latents[0:1], latents[1:2]])
        ctx3 = torch.cat([context[1:2], context[2:3], context[3:4]])
        noise_pred = _unet_nograd(model, controller, latents_input, t, ctx3, "cfg3")
        edit_out = model.scheduler.step(noise_pred[0:1], t, latents[1:2], eta=0.0, eps_cond=noise_pred[2:3],
                                        guidance_scale=guidance_scale)["prev_sample"]
        latents_out = torch.cat([latents[0:1].to(edit_out.dtype), edit_out])
        noise_pred_out = None
    elif use_cfg:
        latents_input = torch.cat([latents] * 2)
        noise_pred = _unet_nograd(model, controller, latents_input, t, context, "cfg4")
        noise_pred_uncond, noise_prediction_text = noise_pred.chunk(2)
        if return_noise:
            noise_pred_out = noise_pred_uncond + guidance_scale * (noise_prediction_text - noise_pred_uncond)
            latents_out = model.scheduler.step(noise_pred_out, t, latents, eta=0.0)["prev_sample"]
        else:
            noise_pred_out = None
            latents_out = model.scheduler.step(noise_pred_uncond, t, latents, eta=0.0, eps_cond=noise_prediction_text,
                                               guidance_scale=guidance_scale)["prev_sample"]
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
def latent2image(vae, latents):
    """diffusion.py:61-68."""
    latents = 1 / 0.18215 * latents
    image = vae.decode(latents)["sample"]
    image = (image.float() / 2 + 0.5).clamp(0, 1)
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
               dtype=torch.float16, tiny=False):
    """diffusion.py:99-149.  With diffusers + weights available this would wrap ``StableDiffusionPipeline.from_pretrained``;
    in this environment neither exists (no network), so a seeded random-init model of the same shape is built.
    Returns (ldm_stable, tokenizer, scheduler) like the reference."""
    try:  # pragma: no cover
        import diffusers  # noqa: F401
        have_diffusers = True
    except Exception:  # noqa: BLE001
        have_diffusers = False
    if have_diffusers and not random_init:  # pragma: no cover - not reachable in the build image
        from diffusers import DDIMScheduler as _DS, StableDiffusionPipeline as _SDP
        from .attention_processors import VanillaAttentionProcessor
        pipe = _SDP.from_pretrained(unet_path or diffusion_model, torch_dtype=dtype).to(device)
        pipe.scheduler = _DS(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)
        pipe.unet.set_attn_processor(VanillaAttentionProcessor())
        pipe.unet.eval()
        return pipe, pipe.tokenizer, pipe.scheduler
    pipe = build_random_sd21(device=device, dtype=dtype, tiny=tiny)
    return pipe, pipe.tokenizer, pipe.scheduler
