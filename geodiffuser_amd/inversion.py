"""DDIM inversion.  Mirror of GeoDiffuser/utils/inversion.py (``NullInversion``: ``invert`` :261-277, ``ddim_loop``
:131-196, ``prev_step`` / ``next_step`` :47-65, ``init_prompt`` :113-128, ``null_optimization`` :213-259 — off in every reference
driver (``perform_inversion=False``), on in ``perform_geometric_edit``'s default signature)."""
from __future__ import annotations

import numpy as np
import torch

from . import diffusion, graphs
from .attention_processors import VanillaAttentionProcessor
from .scheduler import DDIMInverseScheduler, DDIMScheduler


SINGLE_ROW_WHEN_PROMPT_IS_UNCOND = True


class NullInversion:
    def prev_step(self, model_output, timestep: int, sample):
        """inversion.py:47-55 (closed form, executed by gd_ddim_step)."""
        return self.scheduler.step(model_output, timestep, sample, eta=0.0)["prev_sample"]

    def next_step(self, model_output, timestep: int, sample):
        """inversion.py:57-65."""
        return self._inverse.step(model_output, timestep, sample)["prev_sample"]

    @torch.no_grad()
    def latent2image(self, latents, return_type="np"):
        latents = latents.detach() / self.model.vae.config.scaling_factor
        image = self.model.vae.decode(latents)["sample"]
        if return_type == "np":
            image = (image.float() / 2 + 0.5).clamp(0, 1)
            image = image.cpu().permute(0, 2, 3, 1).numpy()[0]
            image = (image * 255).astype(np.uint8)
        return image

    @torch.no_grad()
    def image2latent(self, image):
        if type(image) is torch.Tensor and image.dim() == 4:
            return image
        image = torch.from_numpy(np.ascontiguousarray(image)).to(self.device).float() / 127.5 - 1      # arithmetic on the device
        image = image.permute(2, 0, 1).unsqueeze(0)
        latents = self.model.vae.encode(image)["latent_dist"].mean
        return latents * self.model.vae.config.scaling_factor

    @torch.no_grad()
    def init_prompt(self, prompt: str):
        tok = self.model.tokenizer
        uncond_input = tok([self.uncond_text], padding="max_length", max_length=tok.model_max_length, return_tensors="pt")
        uncond_embeddings = diffusion.encode_text(self.model, uncond_input.input_ids)
        text_input = tok([prompt], padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
        text_embeddings = diffusion.encode_text(self.model, text_input.input_ids)
        self.context = torch.cat([uncond_embeddings, text_embeddings])
        self.prompt = prompt

    @torch.no_grad()
    def ddim_loop(self, latent, latent_2=None):
        """inversion.py:131-196: 50 batch-2 (CFG) UNet passes with the vanilla processor; CFG uses the SAME guidance scale
        as editing (:187)."""
        self.model.unet.set_attn_processor(VanillaAttentionProcessor())
        if latent_2 is not None:
            latent = torch.cat([latent, latent_2], 0)
        all_latent = [latent]
        all_noise = [latent]
        latents = latent.clone().detach()
        inv = self._inverse
        inv.set_timesteps(self.num_ddim_steps, device=self.device)
        context_in = self.context
        if latent_2 is not None:
            uncond_e, cond_e = context_in.chunk(2)
            context_in = torch.cat([uncond_e, uncond_e, cond_e, cond_e], 0)
        # Parity-preserving saving: when the prompt equals the unconditional text (the batch driver always passes "",
        # large_scale_editor.py:196) the two CFG rows of the inversion pass are the same sample, eps_u == eps_c and the guided
        # noise is eps_u for any guidance scale: run that sample once.  (One host sync per edit for the comparison.)
        single = (SINGLE_ROW_WHEN_PROMPT_IS_UNCOND and latent_2 is None and context_in.shape[0] == 2
                  and bool(torch.equal(context_in[0], context_in[1])))
        if single:
            context_in = context_in[1:2]
        for i, t in enumerate(inv.timesteps):
            if self.progress_bar is not None:
                self.progress_bar(i / self.num_ddim_steps, desc="Performing DDIM Inversion")
            latent_model_input = latents if single else torch.cat([latents] * 2)
            if graphs.ENABLED and not torch.is_grad_enabled():
                runner = self.model.__dict__.get("_graphed")
                if runner is None:
                    runner = self.model.__dict__["_graphed"] = graphs.GraphedUNet(self.model.unet)
                noise_pred, _ = runner(("inversion",), latent_model_input, t, context_in, ctx_src=context_in)
            else:
                noise_pred = self.model.unet(latent_model_input, t, encoder_hidden_states=context_in, return_dict=False)[0]
            if single:
                noise_pred_uncond = noise_pred_cond = noise_pred
                latents = inv.step(noise_pred, t, latents, return_dict=False)[0]
            else:
                noise_pred_uncond, noise_pred_cond = noise_pred.chunk(2)
                latents = inv.step(noise_pred_uncond, t, latents, eps_cond=noise_pred_cond, guidance_scale=self.guidance_scale,
                                   return_dict=False)[0]
            all_latent.append(latents.detach())
            all_noise.append(noise_pred_cond.detach().clone())      # a graphed pass returns its static output buffer: keep a copy per step
        return all_latent, all_noise

    @property
    def scheduler(self):
        return self.model.scheduler

    @torch.no_grad()
    def ddim_inversion(self, image, image_2=None):
        latent = self.image2latent(image)
        image_rec = None                      # the reference decodes a reconstruction here and never uses it (:205)
        latent_2 = self.image2latent(image_2) if image_2 is not None else None
        ddim_latents, ddim_noise = self.ddim_loop(latent, latent_2)
        return image_rec, ddim_latents, ddim_noise

    def _prev_step_autograd(self, model_output, timestep: int, sample):
        """inversion.py:47-55 in torch ops: the null-text loss differentiates through the step (gd_ddim_step has no backward)."""
        sch = self.scheduler
        t = int(timestep)
        tp = t - sch.config.num_train_timesteps // sch.num_inference_steps
        a_t = float(sch.alphas_cumprod[t])
        a_p = float(sch.alphas_cumprod[tp]) if tp >= 0 else float(sch.final_alpha_cumprod)
        if getattr(sch, "_v", False):                         # v-prediction: recover x0 / eps from v first
            x0 = a_t ** 0.5 * sample - (1 - a_t) ** 0.5 * model_output
            model_output = a_t ** 0.5 * model_output + (1 - a_t) ** 0.5 * sample
        else:
            x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * model_output

    def get_noise_pred_single(self, latents, t, context):
        """inversion.py:67-70."""
        return self.model.unet(latents, t, encoder_hidden_states=context)["sample"]

    def get_noise_pred(self, latents, t, is_forward=True, context=None):
        """inversion.py:72-85."""
        latents_input = torch.cat([latents] * 2)
        if context is None:
            context = self.context
        guidance_scale = 1 if is_forward else self.guidance_scale
        noise_pred = self.model.unet(latents_input, t, encoder_hidden_states=context)["sample"]
        noise_pred_uncond, noise_prediction_text = noise_pred.chunk(2)
        if is_forward:
            return self.next_step(noise_pred_uncond + guidance_scale * (noise_prediction_text - noise_pred_uncond), t, latents)
        return self.scheduler.step(noise_pred_uncond, t, latents, eta=0.0, eps_cond=noise_prediction_text,
                                   guidance_scale=guidance_scale)["prev_sample"]

    def null_optimization(self, latents, num_inner_steps, epsilon, t_coords=None):
        """inversion.py:213-259 — null-text inversion: per DDIM step, up to ``num_inner_steps`` Adam steps (lr 1e-2 (1 - i/100)) on the
        unconditional embedding so that the guided step from x_t reproduces the inversion trajectory's x_{t-1}; early stop at
        ``epsilon + 2e-5 i``.  The UNet passes run with the vanilla processor; their attention goes through gd_attn_fwd and the full
        backward (dQ, dK, dV: gd_attn_bwd / gd_attn_bwd_dkv).  Off in every reference driver ("not required for GeoDiffuser",
        inversion.py:269), on by default in ``perform_geometric_edit``'s signature (editor.py:437)."""
        unet = self.model.unet
        # the reference leaves the processors alone (its vanilla path is diffusers' default); here the vanilla processor is attached for
        # the optimisation and whatever the caller had registered — and the parameters' requires_grad flags — come back afterwards
        saved_procs, saved_rg = dict(unet.attn_processors), [p.requires_grad for p in unet.parameters()]
        unet.set_attn_processor(VanillaAttentionProcessor())
        try:
            return self._null_optimization(latents, num_inner_steps, epsilon)
        finally:
            unet.set_attn_processor(saved_procs)
            for p, rg in zip(unet.parameters(), saved_rg):
                p.requires_grad = rg

    def _null_optimization(self, latents, num_inner_steps, epsilon):
        uncond_embeddings, cond_embeddings = self.context.chunk(2)
        uncond_embeddings_list = []
        latent_cur = latents[-1]
        with torch.enable_grad():
            for p in self.model.unet.parameters():
                p.requires_grad = False
            for i in range(self.num_ddim_steps):
                uncond_embeddings = uncond_embeddings.clone().detach().float()
                uncond_embeddings.requires_grad = True
                optimizer = torch.optim.Adam([uncond_embeddings], lr=1e-2 * (1.0 - i / 100.0))
                latent_prev = latents[len(latents) - i - 2].detach().float()
                t = self.model.scheduler.timesteps[i]
                with torch.no_grad():
                    noise_pred_cond = self.get_noise_pred_single(latent_cur, t, cond_embeddings).float()
                for j in range(num_inner_steps):
                    if self.progress_bar is not None:
                        self.progress_bar((i * num_inner_steps + j) / (self.num_ddim_steps * num_inner_steps), desc="Null-text optimization")
                    noise_pred_uncond = self.get_noise_pred_single(latent_cur, t, uncond_embeddings).float()
                    noise_pred = noise_pred_uncond + self.guidance_scale * (noise_pred_cond - noise_pred_uncond)
                    latents_prev_rec = self._prev_step_autograd(noise_pred, t, latent_cur.float())
                    loss = torch.nn.functional.mse_loss(latents_prev_rec, latent_prev)
                    optimizer.zero_grad()
                    loss.backward()
                    optimizer.step()
                    if loss.item() < epsilon + i * 2e-5:
                        break
                uncond_embeddings_list.append(uncond_embeddings[:1].detach())
                with torch.no_grad():
                    context = torch.cat([uncond_embeddings.detach().to(cond_embeddings.dtype), cond_embeddings])
                    latent_cur = self.get_noise_pred(latent_cur, t, False, context)
        return uncond_embeddings_list

    def invert(self, image_gt, prompt: str, offsets=(0, 0, 0, 0), num_inner_steps=10, early_stop_epsilon=1e-5, verbose=False,
               t_coords=None, perform_inversion=True, image_2=None):
        """inversion.py:261-277."""
        self.init_prompt(prompt)
        image_rec, ddim_latents, ddim_noise = self.ddim_inversion(image_gt, image_2)
        if perform_inversion:
            uncond_embeddings = self.null_optimization(ddim_latents, num_inner_steps, early_stop_epsilon, t_coords=t_coords)
        else:
            uncond_embeddings = None
        return (image_gt, image_rec), ddim_latents[-1], uncond_embeddings, ddim_latents, ddim_noise

    def __init__(self, model, num_ddim_steps=50, uncond_text="", device="cuda:0", progress_bar=None, guidance_scale=3.0):
        self.guidance_scale = guidance_scale
        self.progress_bar = progress_bar
        self.device = device
        self.num_ddim_steps = num_ddim_steps
        self.uncond_text = uncond_text
        self.model = model
        self.tokenizer = self.model.tokenizer
        self.model.scheduler.set_timesteps(self.num_ddim_steps)
        self._inverse = DDIMInverseScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                             clip_sample=False, set_alpha_to_one=False)
        self._inverse.set_timesteps(self.num_ddim_steps)
        self.prompt = None
        self.context = None
        self.controller = None
