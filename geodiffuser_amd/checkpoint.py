"""Real weights through the fast harness: the public Stable Diffusion checkpoints (diffusers directory layout, safetensors) into the
modules of this package — the UNet whose hooked layers, NHWC convolutions, fused norms and captured passes the edit path runs on
(unet_sd21.py), the VAE and the text tower (pipeline.py).

The reference loads ``StableDiffusionPipeline.from_pretrained(...)`` (GeoDiffuser/utils/diffusion.py:99-140) and hooks diffusers' own UNet;
with diffusers importable this package does the same (diffusion.load_model), but then none of the harness's kernels apply.  This module
is the other route: same files, our modules.

    pipe = checkpoint.from_safetensors("/path/to/stable-diffusion-2-1-base", device="cuda:0", dtype=torch.float16)

* UNet: the module tree of unet_sd21.UNet2DConditionModel carries diffusers' parameter names one to one (tests/test_checkpoint.py holds
  its ``state_dict()`` to the public key / shape list, tests/golden/sd21_base_keys.json); SD1.x checkpoints store the transformers'
  ``proj_in`` / ``proj_out`` as 1x1 convolutions: reshaped to the Linear this harness uses.
* VAE (AutoencoderKL) and text encoder (transformers CLIPTextModel): explicit name maps onto pipeline.AutoencoderKL / TextEncoder
  (``vae_key_map`` / ``text_key_map``); the text tower's q / k / v projections are stacked into nn.MultiheadAttention's ``in_proj``.
  Checkpoints that predate diffusers' attention rename (``query`` / ``key`` / ``value`` / ``proj_attn`` in the VAE's mid-block layer) are
  accepted.
* Every loader is strict: a missing key, an unexpected key or a shape mismatch raises ``CheckpointError`` naming the first few offenders —
  nothing is silently left at its random initialisation.  No checkpoint exists in the build environment (no network): the maps are held
  to the public key lists on the CPU and exercised end to end on synthetic checkpoint directories of the same layout.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import torch


class CheckpointError(RuntimeError):
    pass


def _check(want: Dict[str, Tuple[int, ...]], have: Dict[str, Tuple[int, ...]], what: str):
    missing = [k for k in want if k not in have]
    extra = [k for k in have if k not in want]
    bad = [(k, have[k], want[k]) for k in want if k in have and tuple(have[k]) != tuple(want[k])]
    if missing or extra or bad:
        raise CheckpointError(f"{what}: {len(missing)} missing, {len(extra)} unexpected, {len(bad)} of another shape — missing {missing[:4]}, "
                              f"unexpected {extra[:4]}, shapes (checkpoint, module) {bad[:4]}")


# ---------------------------------------------------------------------------------------------------------- UNet
@torch.no_grad()
def load_unet_state_dict(unet, sd: Dict[str, torch.Tensor], validate_only: bool = False) -> int:
    """diffusers ``UNet2DConditionModel`` state dict -> unet_sd21.UNet2DConditionModel (same names).  Returns the number of parameters
    loaded.  ``validate_only``: names and shapes only (tensors may live on the meta device)."""
    own = unet.state_dict()
    fixed = {}
    for k, t in sd.items():
        if k in own and t.dim() == 4 and own[k].dim() == 2 and t.shape[2:] == (1, 1) and k.rsplit(".", 2)[-2] in ("proj_in", "proj_out"):
            t = t.reshape(t.shape[0], t.shape[1])                # SD1.x: use_linear_projection False
        fixed[k] = t
    _check({k: tuple(v.shape) for k, v in own.items()}, {k: tuple(v.shape) for k, v in fixed.items()}, "UNet checkpoint")
    if not validate_only:
        for k, v in own.items():
            v.copy_(fixed[k].to(device=v.device, dtype=v.dtype))
    return sum(v.numel() for v in own.values())


# ---------------------------------------------------------------------------------------------------------- VAE
def vae_key_map(vae) -> Dict[str, torch.Tensor]:
    """public AutoencoderKL parameter name -> the tensor of pipeline.AutoencoderKL it lands in."""
    from .pipeline import _NormSiLU, _VAttn, _VDown, _VRes
    import torch.nn as nn
    m: Dict[str, torch.Tensor] = {}

    def put(prefix, mod):
        for n, p in mod.named_parameters(recurse=False):
            m[f"{prefix}.{n}"] = p

    def res(prefix, r):
        put(prefix + ".norm1", r.n1); put(prefix + ".conv1", r.c1); put(prefix + ".norm2", r.n2); put(prefix + ".conv2", r.c2)
        if r.sc is not None:
            put(prefix + ".conv_shortcut", r.sc)

    def attn(prefix, a):
        put(prefix + ".group_norm", a.norm); put(prefix + ".to_q", a.q); put(prefix + ".to_k", a.k); put(prefix + ".to_v", a.v)
        put(prefix + ".to_out.0", a.o)

    def walk(seq, side, block_name, sampler_name):
        mods = list(seq)
        assert isinstance(mods[0], nn.Conv2d)
        put(f"{side}.conv_in", mods[0])
        i = 1
        if side == "decoder":                                    # the decoder's mid block comes first
            res("decoder.mid_block.resnets.0", mods[1]); attn("decoder.mid_block.attentions.0", mods[2]); res("decoder.mid_block.resnets.1", mods[3])
            i = 4
        blk = j = 0
        while i < len(mods):
            mod = mods[i]
            if isinstance(mod, _VRes) and isinstance(mods[i + 1] if i + 1 < len(mods) else None, _VAttn) and side == "encoder":
                res("encoder.mid_block.resnets.0", mod); attn("encoder.mid_block.attentions.0", mods[i + 1]); res("encoder.mid_block.resnets.1", mods[i + 2])
                i += 3
                continue
            if isinstance(mod, _VRes):
                res(f"{side}.{block_name}.{blk}.resnets.{j}", mod)
                j += 1
            elif isinstance(mod, _VDown):
                put(f"encoder.down_blocks.{blk}.downsamplers.0.conv", mod.conv)
                blk, j = blk + 1, 0
            elif isinstance(mod, nn.Upsample):
                put(f"decoder.up_blocks.{blk}.upsamplers.0.conv", mods[i + 1])
                blk, j = blk + 1, 0
                i += 1
            else:
                raise CheckpointError(f"vae_key_map: unexpected module {type(mod).__name__} in the {side}")
            i += 1

    walk(vae.encoder, "encoder", "down_blocks", "downsamplers")
    put("encoder.conv_norm_out", vae.enc_out[0].norm); put("encoder.conv_out", vae.enc_out[1])
    put("quant_conv", vae.quant_conv); put("post_quant_conv", vae.post_quant_conv)
    walk(vae.decoder, "decoder", "up_blocks", "upsamplers")
    put("decoder.conv_norm_out", vae.dec_out[0].norm); put("decoder.conv_out", vae.dec_out[1])
    assert isinstance(vae.enc_out[0], _NormSiLU)
    return m


_VAE_OLD_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


@torch.no_grad()
def load_vae_state_dict(vae, sd: Dict[str, torch.Tensor], validate_only: bool = False) -> int:
    m = vae_key_map(vae)
    fixed = {}
    for k, t in sd.items():
        parts = k.split(".")
        if len(parts) >= 2 and parts[-2] in _VAE_OLD_ATTN and "attentions" in parts:            # pre-rename checkpoints
            k = ".".join(parts[:-2] + [_VAE_OLD_ATTN[parts[-2]], parts[-1]])
        if k in m and t.dim() == 4 and m[k].dim() == 2 and t.shape[2:] == (1, 1):                  # ... which also stored the layer as 1x1 convolutions
            t = t.reshape(t.shape[0], t.shape[1])
        fixed[k] = t
    _check({k: tuple(v.shape) for k, v in m.items()}, {k: tuple(v.shape) for k, v in fixed.items()}, "VAE checkpoint")
    if not validate_only:
        for k, v in m.items():
            v.copy_(fixed[k].to(device=v.device, dtype=v.dtype))
    return sum(v.numel() for v in m.values())


# ---------------------------------------------------------------------------------------------------------- text encoder
def text_key_shapes(te) -> Dict[str, Tuple[int, ...]]:
    """transformers ``CLIPTextModel`` parameter name -> shape, for a pipeline.TextEncoder of this width / depth."""
    d = te.tok.weight.shape[1]
    out = {"text_model.embeddings.token_embedding.weight": tuple(te.tok.weight.shape),
           "text_model.embeddings.position_embedding.weight": tuple(te.pos.shape)}
    for i, l in enumerate(te.layers):
        p = f"text_model.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[f"{p}.self_attn.{n}.weight"], out[f"{p}.self_attn.{n}.bias"] = (d, d), (d,)
        for n in ("layer_norm1", "layer_norm2"):
            out[f"{p}.{n}.weight"], out[f"{p}.{n}.bias"] = (d,), (d,)
        out[f"{p}.mlp.fc1.weight"], out[f"{p}.mlp.fc1.bias"] = tuple(l.fc1.weight.shape), tuple(l.fc1.bias.shape)
        out[f"{p}.mlp.fc2.weight"], out[f"{p}.mlp.fc2.bias"] = tuple(l.fc2.weight.shape), tuple(l.fc2.bias.shape)
    out["text_model.final_layer_norm.weight"], out["text_model.final_layer_norm.bias"] = (d,), (d,)
    return out


@torch.no_grad()
def load_text_state_dict(te, sd: Dict[str, torch.Tensor], validate_only: bool = False) -> int:
    """``CLIPTextModel`` state dict -> pipeline.TextEncoder: q / k / v projections stacked into nn.MultiheadAttention's in_proj (its own
    order: q, k, v), everything else one to one.  ``text_model.embeddings.position_ids`` (a buffer some exports carry) is ignored."""
    sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}
    want = text_key_shapes(te)
    _check(want, {k: tuple(v.shape) for k, v in sd.items()}, "text encoder checkpoint")
    if validate_only:
        return sum(p.numel() for p in te.parameters())

    def cp(dst, src):
        dst.copy_(src.to(device=dst.device, dtype=dst.dtype))

    cp(te.tok.weight, sd["text_model.embeddings.token_embedding.weight"]); cp(te.pos, sd["text_model.embeddings.position_embedding.weight"])
    for i, l in enumerate(te.layers):
        p = f"text_model.encoder.layers.{i}"
        cp(l.attn.in_proj_weight, torch.cat([sd[f"{p}.self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0))
        cp(l.attn.in_proj_bias, torch.cat([sd[f"{p}.self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0))
        cp(l.attn.out_proj.weight, sd[f"{p}.self_attn.out_proj.weight"]); cp(l.attn.out_proj.bias, sd[f"{p}.self_attn.out_proj.bias"])
        for dst, n in ((l.ln1, "layer_norm1"), (l.ln2, "layer_norm2"), (l.fc1, "mlp.fc1"), (l.fc2, "mlp.fc2")):
            cp(dst.weight, sd[f"{p}.{n}.weight"]); cp(dst.bias, sd[f"{p}.{n}.bias"])
    cp(te.ln_f.weight, sd["text_model.final_layer_norm.weight"]); cp(te.ln_f.bias, sd["text_model.final_layer_norm.bias"])
    return sum(p.numel() for p in te.parameters())


# ---------------------------------------------------------------------------------------------------------- the directory
def _read_tensors(folder: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    names = sorted(f for f in os.listdir(folder) if f.endswith(".safetensors") and "fp16" not in f and "non_ema" not in f) or \
        sorted(f for f in os.listdir(folder) if f.endswith(".safetensors"))
    if not names:
        raise CheckpointError(f"{folder}: no .safetensors file (convert .bin checkpoints with the safetensors tools first)")
    out: Dict[str, torch.Tensor] = {}
    shards = [n for n in names if "-of-" in n] or names[:1]                   # a sharded export, or the one file
    for n in shards:
        out.update(load_file(os.path.join(folder, n)))
    return out


def _config(folder: str) -> dict:
    p = os.path.join(folder, "config.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def unet_from_config(cfg: dict):
    """The harness's UNet for a diffusers ``unet/config.json``.  Supported: the SD family with one transformer block per attention
    (SD1.x, SD2.x) and the SDXL-base layout; anything else is refused by name."""
    from .unet_sd21 import UNet2DConditionModel
    ch = tuple(cfg.get("block_out_channels", (320, 640, 1280, 1280)))
    ahd = cfg.get("attention_head_dim", 8)
    heads = tuple(ahd) if isinstance(ahd, (list, tuple)) else (int(ahd),) * len(ch)       # diffusers' historical naming: this IS the head count
    down = cfg.get("down_block_types") or ["CrossAttnDownBlock2D"] * (len(ch) - 1) + ["DownBlock2D"]
    attn_levels = tuple(t.startswith("CrossAttn") for t in down)
    tdepth = cfg.get("transformer_layers_per_block", 1)
    tdepth = tuple(tdepth) if isinstance(tdepth, (list, tuple)) else (int(tdepth),) * len(ch)
    kw = {}
    if cfg.get("addition_embed_type") == "text_time":
        kw = dict(addition_time_embed_dim=int(cfg["addition_time_embed_dim"]),
                  addition_text_embed_dim=int(cfg["projection_class_embeddings_input_dim"]) - 6 * int(cfg["addition_time_embed_dim"]))
    elif cfg.get("addition_embed_type") or cfg.get("class_embed_type"):
        raise CheckpointError(f"unet/config.json: addition_embed_type {cfg.get('addition_embed_type')!r} / class_embed_type "
                              f"{cfg.get('class_embed_type')!r} not supported by this harness")
    return UNet2DConditionModel(in_channels=int(cfg.get("in_channels", 4)), out_channels=int(cfg.get("out_channels", 4)), block_out_channels=ch,
                                heads=heads, cross_attention_dim=int(cfg.get("cross_attention_dim", 1024)),
                                layers_per_block=int(cfg.get("layers_per_block", 2)), attn_levels=attn_levels, transformer_depth=tdepth, **kw)


def from_safetensors(model_dir: str, device="cuda:0", dtype=torch.float16):
    """A diffusers Stable Diffusion directory (``unet/``, ``vae/``, ``text_encoder/``, optionally ``tokenizer/`` and ``scheduler/``) ->
    pipeline.StableDiffusionPipeline on this package's modules, ready for ``run_geodiffuser(..., ldm_stable_model=pipe)``.
    The CLIP tokenizer needs ``transformers`` and the directory's vocabulary files; without them the stand-in tokenizer (exact for the
    empty prompt only) is used and says so."""
    from .pipeline import AutoencoderKL, SimpleTokenizer, StableDiffusionPipeline, TextEncoder
    from .scheduler import DDIMScheduler
    ucfg, vcfg, tcfg = (_config(os.path.join(model_dir, d)) for d in ("unet", "vae", "text_encoder"))
    if os.path.isdir(os.path.join(model_dir, "text_encoder_2")):
        raise CheckpointError("SDXL checkpoints (two text towers) are not wired into from_safetensors: the reference has no SDXL path either "
                              "(GeoDiffuser/utils/diffusion.py:106 is commented out)")
    unet = unet_from_config(ucfg)
    vae = AutoencoderKL(ch=tuple(vcfg.get("block_out_channels", (128, 256, 512, 512))), latent=int(vcfg.get("latent_channels", 4)),
                        scaling_factor=float(vcfg.get("scaling_factor", 0.18215)))
    if tcfg.get("hidden_act", "gelu") not in ("gelu",):
        raise CheckpointError(f"text_encoder/config.json: hidden_act {tcfg.get('hidden_act')!r} (SD1.x's quick_gelu tower) is not what "
                              f"pipeline.TextEncoder computes (gelu: the OpenCLIP tower of SD2.x)")
    te = TextEncoder(width=int(tcfg.get("hidden_size", 1024)), layers=int(tcfg.get("num_hidden_layers", 23)),
                     heads=int(tcfg.get("num_attention_heads", 16)), vocab=int(tcfg.get("vocab_size", 49408)),
                     max_len=int(tcfg.get("max_position_embeddings", 77)))
    n = load_unet_state_dict(unet, _read_tensors(os.path.join(model_dir, "unet")))
    n += load_vae_state_dict(vae, _read_tensors(os.path.join(model_dir, "vae")))
    n += load_text_state_dict(te, _read_tensors(os.path.join(model_dir, "text_encoder")))
    cl = os.environ.get("GD_CHANNELS_LAST", "1") == "1"
    for m in (unet, vae, te):
        m.to(device=device, dtype=dtype).eval()
        if cl and m is not te:
            m.to(memory_format=torch.channels_last)
        for p in m.parameters():
            p.requires_grad_(False)
    tok = None
    tdir = os.path.join(model_dir, "tokenizer")
    if os.path.isdir(tdir):
        try:
            from transformers import CLIPTokenizer
            tok = CLIPTokenizer.from_pretrained(tdir)
        except Exception as e:  # noqa: BLE001
            print(f"[geodiffuser_amd.checkpoint] CLIP tokenizer not loaded from {tdir} ({e!r}): stand-in tokenizer, exact for the empty prompt only")
    if tok is None:
        tok = SimpleTokenizer()
    scfg = _config(os.path.join(model_dir, "scheduler")) if os.path.isdir(os.path.join(model_dir, "scheduler")) else {}
    sp = os.path.join(model_dir, "scheduler", "scheduler_config.json")
    if os.path.exists(sp):
        scfg = json.load(open(sp))
    sched = DDIMScheduler(beta_start=float(scfg.get("beta_start", 0.00085)), beta_end=float(scfg.get("beta_end", 0.012)),
                          beta_schedule=scfg.get("beta_schedule", "scaled_linear"), clip_sample=False, set_alpha_to_one=False,
                          prediction_type=scfg.get("prediction_type", "epsilon"))
    pipe = StableDiffusionPipeline(unet, vae, te, tok, sched, device)
    pipe.loaded_parameters = n
    return pipe
