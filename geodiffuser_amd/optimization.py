"""Latent / embedding gradient step and the adaptive loss-weight schedule.

Mirror of GeoDiffuser/utils/optimization.py:7-105,165-253 (the live ``optimizer is None`` branch; ``use_optimizer``
is never forwarded by the reference driver).  Same function names and argument meaning.
"""
from __future__ import annotations

import torch

from . import ops
from .generic_torch import reshape_attention_mask


def _adaptive(controller, i, skip_optim_steps, out_loss_log_dict, num_ddim_steps, removal_loss_value_in, down: float):
    cur = out_loss_log_dict["self"]["removal"]
    frac = i / num_ddim_steps
    if frac < 0.4:
        remaining_steps = int((0.4 - frac) * num_ddim_steps / skip_optim_steps)
        expected = removal_loss_value_in / (1.25) ** remaining_steps
        if expected < cur:
            controller.loss_weight_dict["self"]["removal"] *= 1.3
        elif 2.5 * expected > cur:
            controller.loss_weight_dict["self"]["removal"] /= down
    elif 0.4 < frac < 0.8:
        if (removal_loss_value_in - 0.3) < cur:
            controller.loss_weight_dict["self"]["removal"] *= 2.0
        else:
            controller.initialize_default_loss_weights()
    else:
        controller.initialize_default_loss_weights()


def adaptive_optimization_step_editing(controller, i, skip_optim_steps, out_loss_log_dict, num_ddim_steps, removal_loss_value_in=-1.5):
    """optimization.py:7-56."""
    _adaptive(controller, i, skip_optim_steps, out_loss_log_dict, num_ddim_steps, removal_loss_value_in, 2.0)


def adaptive_optimization_step_remover(controller, i, skip_optim_steps, out_loss_log_dict, num_ddim_steps, removal_loss_value_in=-1.5):
    """optimization.py:58-105."""
    _adaptive(controller, i, skip_optim_steps, out_loss_log_dict, num_ddim_steps, removal_loss_value_in, 2.5)


def _update_latent(latents: torch.Tensor, loss: torch.Tensor, step_size: float, mask=None, context=None, scaler=None,
                   optimizer=None):
    """optimization.py:165-253.  ``grads = autograd.grad(loss, [latents, context])`` then
    x1 <- x1 - step*(1+m)*nan_to_num(g1)  (two chained masked updates, gd_masked_latent_update) and
    ctx[-1] <- ctx[-1] - step*nan_to_num(g_ctx[-1]).  Only the last batch entry is ever updated."""
    if optimizer is not None or scaler is not None:
        raise NotImplementedError("the optimizer / GradScaler branches are dead in the reference (SURVEY.md F4)")
    if context is None:
        raise ValueError("context is required (the reference always passes it, editor.py:273)")
    # In the reference every leaf is graph-reachable through the batched tensors even where its gradient is identically
    # zero (e.g. the text embedding for the remover, whose replace path uses only detached keys/values); the fused layer
    # returns no gradient there, hence allow_unused + explicit zeros.
    grad_cond, context_grad = _latent_grads(latents, loss, context)
    return _apply_latent_update(latents, grad_cond, context, context_grad, step_size, mask)


def _latent_grads(latents: torch.Tensor, loss: torch.Tensor, context: torch.Tensor):
    """The autograd half of ``_update_latent`` (optimization.py:190-196): d loss / d latents, d loss / d context."""
    grads = torch.autograd.grad(loss, [latents, context], retain_graph=False, allow_unused=True)
    grad_cond = grads[0] if grads[0] is not None else torch.zeros_like(latents)
    context_grad = grads[1] if grads[1] is not None else torch.zeros_like(context)
    return grad_cond, context_grad


def _apply_latent_update(latents, grad_cond, context, context_grad, step_size: float, mask=None):
    """The arithmetic half of ``_update_latent`` (optimization.py:197-253)."""
    context_grad = torch.nan_to_num(context_grad, posinf=0.0, neginf=0.0, nan=0.0)
    x1 = latents[-1].detach().float().contiguous()
    g1 = grad_cond[-1].detach().float().contiguous()
    if mask is not None:
        m = reshape_attention_mask(mask[None, None].to(latents.device).float().reshape(1, 1, *mask.shape[-2:]),
                                   in_mat_shape=latents[-1:].shape)
        m = m[-1, 0].reshape(-1).contiguous()
    else:
        m = torch.zeros(latents.shape[-1] * latents.shape[-2], dtype=torch.float32, device=latents.device)
        g1 = g1 * 1.0
    if mask is not None:
        new_last = ops.masked_latent_update(x1, g1, m, float(step_size))
    else:
        new_last = x1 - step_size * torch.nan_to_num(g1, posinf=0.0, neginf=0.0, nan=0.0)
    latents_out = torch.cat([latents[:-1].detach(), new_last[None].to(latents.dtype)], 0)
    context_new = torch.cat([context[:-1].detach(), context[-1:].detach() - step_size * context_grad[-1:]], 0)
    return latents_out, context_new
