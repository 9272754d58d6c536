"""Model container with the members the reference reads from a diffusers ``StableDiffusionPipeline``
(``unet``, ``vae``, ``text_encoder``, ``tokenizer``, ``scheduler``, ``device``, ``progress_bar``): enough of
SD2.1-base's *shape* to make edits/sec measurable when diffusers and the weights are not available (no network in the
build / benchmark environment).  VAE and text encoder are stock PyTorch modules (plumbing around the hot path, SURVEY.md
8f N2); all UNet attention runs through the HIP processors.

Random-init weights are seeded; throughput does not depend on weight values.
"""
from __future__ import annotations

import contextlib
import math
import zlib
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .scheduler import DDIMScheduler
from .unet_sd21 import GroupNormAct, UNet2DConditionModel, sd14_unet, sdxl_unet, tiny_unet


# ------------------------------------------------------------------------------------------------ tokenizer
class SimpleTokenizer:
    """Stand-in for the CLIP BPE tokenizer (vocabulary files are not available offline): BOS, hashed word ids, EOS,
    pad 0 — exact for the empty prompt every reference driver uses for the unconditional branch."""

    model_max_length = 77
    bos, eos, pad = 49406, 49407, 0

    def __call__(self, texts, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        if isinstance(texts, str):
            texts = [texts]
        L = max_length or self.model_max_length
        ids = torch.full((len(texts), L), self.pad, dtype=torch.long)
        for i, t in enumerate(texts):
            toks = [self.bos] + [1000 + (zlib.crc32(w.encode()) % 40000) for w in t.lower().split()][: L - 2] + [self.eos]
            ids[i, : len(toks)] = torch.tensor(toks)
        return SimpleNamespace(input_ids=ids)


class _TextLayer(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.ln1, self.ln2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads, batch_first=True)
        self.fc1, self.fc2 = nn.Linear(d, 4 * d), nn.Linear(4 * d, d)

    def forward(self, x, mask):
        h = self.ln1(x)
        x = x + self.attn(h, h, h, attn_mask=mask, need_weights=False)[0]
        return x + self.fc2(F.gelu(self.fc1(self.ln2(x))))


class TextEncoder(nn.Module):
    """OpenCLIP ViT-H text tower shape: width 1024, 23 layers, 16 heads, 77 positions (causal)."""

    def __init__(self, width=1024, layers=23, heads=16, vocab=49408, max_len=77):
        super().__init__()
        self.tok = nn.Embedding(vocab, width)
        self.pos = nn.Parameter(torch.randn(max_len, width) * 0.01)
        self.layers = nn.ModuleList([_TextLayer(width, heads) for _ in range(layers)])
        self.ln_f = nn.LayerNorm(width)

    def forward(self, input_ids):
        x = self.tok(input_ids) + self.pos[: input_ids.shape[1]].to(self.tok.weight.dtype)
        L = x.shape[1]
        mask = torch.full((L, L), float("-inf"), device=x.device, dtype=x.dtype).triu(1)
        for l in self.layers:
            x = l(x, mask)
        return (self.ln_f(x),)


class DualTextEncoder(nn.Module):
    """SDXL's two text towers behind the one ``text_encoder(ids)[0]`` call the reference's drivers make (U/editor.py:156-163): the
    context is the two towers' hidden states concatenated on the feature axis (768 + 1280 = 2048); the second tower's pooled output
    (its EOS position, projected) is kept in ``last_pooled`` for the UNet's text_time embedding."""

    def __init__(self, width_1=768, layers_1=12, heads_1=12, width_2=1280, layers_2=32, heads_2=20):
        super().__init__()
        self.encoder_1 = TextEncoder(width_1, layers_1, heads_1)
        self.encoder_2 = TextEncoder(width_2, layers_2, heads_2)
        self.text_projection = nn.Linear(width_2, width_2, bias=False)
        self.last_pooled = None

    def forward(self, input_ids):
        h1, h2 = self.encoder_1(input_ids)[0], self.encoder_2(input_ids)[0]
        eos = (input_ids == SimpleTokenizer.eos).float().argmax(dim=1)
        self.last_pooled = self.text_projection(h2[torch.arange(h2.shape[0], device=h2.device), eos])
        return (torch.cat([h1, h2], dim=-1), self.last_pooled)


# ------------------------------------------------------------------------------------------------ VAE
class _VRes(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.n1, self.c1 = GroupNormAct(32, cin, eps=1e-6), nn.Conv2d(cin, cout, 3, padding=1)
        self.n2, self.c2 = GroupNormAct(32, cout, eps=1e-6), nn.Conv2d(cout, cout, 3, padding=1)
        self.sc = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.c1(self.n1(x, silu=True))              # fused channels-last GroupNorm + SiLU (HIP) on 16-bit GPU tensors
        h = self.c2(self.n2(h, silu=True))
        return (self.sc(x) if self.sc is not None else x) + h


class _NormSiLU(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.norm = GroupNormAct(32, ch, eps=1e-6)

    def forward(self, x):
        return self.norm(x, silu=True)


class _VAttn(nn.Module):
    """Single-head 512-dim spatial attention of the VAE mid block (once per encode/decode; not the hot path)."""

    def __init__(self, ch):
        super().__init__()
        self.norm = GroupNormAct(32, ch, eps=1e-6)
        self.q, self.k, self.v, self.o = (nn.Linear(ch, ch) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.norm(x).reshape(b, c, h * w).transpose(1, 2)
        a = F.scaled_dot_product_attention(self.q(t)[:, None], self.k(t)[:, None], self.v(t)[:, None])[:, 0]
        return x + self.o(a).transpose(1, 2).reshape(b, c, h, w)


class _VDown(nn.Module):
    """The VAE encoder's stride-2 convolution as the public AutoencoderKL computes it: zero padding on the right / bottom only
    (``F.pad(x, (0, 1, 0, 1))``, then a 3x3 stride-2 convolution without padding) — with symmetric padding real weights would see every
    feature map shifted by half a pixel."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class AutoencoderKL(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), latent=4, scaling_factor=0.18215):
        super().__init__()
        self.config = SimpleNamespace(scaling_factor=scaling_factor)
        enc = [nn.Conv2d(3, ch[0], 3, padding=1)]
        cin = ch[0]
        for i, c in enumerate(ch):
            enc += [_VRes(cin, c), _VRes(c, c)]
            if i < len(ch) - 1:
                enc.append(_VDown(c))
            cin = c
        enc += [_VRes(cin, cin), _VAttn(cin), _VRes(cin, cin)]
        self.encoder = nn.Sequential(*enc)
        self.enc_out = nn.Sequential(_NormSiLU(cin), nn.Conv2d(cin, 2 * latent, 3, padding=1))
        self.quant_conv = nn.Conv2d(2 * latent, 2 * latent, 1)
        self.post_quant_conv = nn.Conv2d(latent, latent, 1)
        dec = [nn.Conv2d(latent, cin, 3, padding=1), _VRes(cin, cin), _VAttn(cin), _VRes(cin, cin)]
        for i, c in enumerate(reversed(ch)):
            dec += [_VRes(cin, c), _VRes(c, c), _VRes(c, c)]
            if i < len(ch) - 1:
                dec += [nn.Upsample(scale_factor=2.0, mode="nearest"), nn.Conv2d(c, c, 3, padding=1)]
            cin = c
        self.decoder = nn.Sequential(*dec)
        self.dec_out = nn.Sequential(_NormSiLU(cin), nn.Conv2d(cin, 3, 3, padding=1))

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def encode(self, x):
        moments = self.quant_conv(self.enc_out(self.encoder(x.to(self.dtype))))
        mean = moments.chunk(2, dim=1)[0]
        return {"latent_dist": SimpleNamespace(mean=mean)}

    def decode(self, z):
        return {"sample": self.dec_out(self.decoder(self.post_quant_conv(z.to(self.dtype))))}


# ------------------------------------------------------------------------------------------------ pipeline
class StableDiffusionPipeline:
    def __init__(self, unet, vae, text_encoder, tokenizer, scheduler, device):
        self.unet, self.vae, self.text_encoder, self.tokenizer, self.scheduler = unet, vae, text_encoder, tokenizer, scheduler
        self.device = torch.device(device)

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        yield SimpleNamespace(update=lambda *a, **k: None)

    def parameters(self):
        for m in (self.unet, self.vae, self.text_encoder):
            yield from m.parameters()


def build_random_sd21(device="cuda:0", dtype=torch.float16, seed=1234, tiny=False, sd14=False) -> StableDiffusionPipeline:
    """Seeded random-init SD2.1-base-shaped model.  ``tiny``: same topology, narrow (for smoke tests).  ``sd14``: the SD1.x head layout
    (8 heads per level: head dims 40 / 80 / 160) and text width (768) of the reference's default model."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        if sd14:
            unet = sd14_unet(tiny=tiny, ctx_dim=64 if tiny else 768)
            vae = AutoencoderKL(ch=(32, 32, 64, 64)) if tiny else AutoencoderKL()
            te = TextEncoder(width=64, layers=2, heads=2) if tiny else TextEncoder(width=768, layers=12, heads=12)
        elif tiny:
            unet = tiny_unet(ctx_dim=64)
            vae = AutoencoderKL(ch=(32, 32, 64, 64))
            te = TextEncoder(width=64, layers=2, heads=2)
        else:
            unet = UNet2DConditionModel()
            vae = AutoencoderKL()
            te = TextEncoder()
    finally:
        torch.random.set_rng_state(g)
    import os
    cl = os.environ.get("GD_CHANNELS_LAST", "1") == "1"
    for m in (unet, vae, te):
        m.to(device=device, dtype=dtype).eval()
        if cl and m is not te:
            m.to(memory_format=torch.channels_last)      # MIOpen's igemm kernels are NHWC: avoids per-conv layout transposes
        for p in m.parameters():
            p.requires_grad_(False)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)
    return StableDiffusionPipeline(unet, vae, te, SimpleTokenizer(), sched, device)


def build_random_sdxl(device="cuda:0", dtype=torch.float16, seed=4321, tiny=False, image_size=1024) -> StableDiffusionPipeline:
    """Seeded random-init SDXL-base-shaped model (BASELINE configs[4]; the reference's own SDXL line is commented out,
    U/diffusion.py:106): UNet ``sdxl_unet``, two text towers behind one ``text_encoder`` call, the VAE of the SD family with SDXL's scaling
    factor.  The UNet's text_time conditioning defaults to the pooled embedding of the empty prompt and the micro-conditioning ids
    (original size, crop 0, target size) of an un-cropped ``image_size`` square — every driver of the reference edits with the empty
    prompt; pass ``added_cond_kwargs`` to the UNet for anything else."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        if tiny:
            te = DualTextEncoder(32, 2, 2, 64, 2, 2)
            unet = sdxl_unet(tiny=True, ctx_dim=96, text_embed_dim=64)
            vae = AutoencoderKL(ch=(32, 32, 64, 64), scaling_factor=0.13025)
        else:
            te = DualTextEncoder()
            unet = sdxl_unet()
            vae = AutoencoderKL(scaling_factor=0.13025)
    finally:
        torch.random.set_rng_state(g)
    import os
    cl = os.environ.get("GD_CHANNELS_LAST", "1") == "1"
    for m in (unet, vae, te):
        m.to(device=device, dtype=dtype).eval()
        if cl and m is not te:
            m.to(memory_format=torch.channels_last)
        for p in m.parameters():
            p.requires_grad_(False)
    tok = SimpleTokenizer()
    with torch.no_grad():
        te(tok([""]).input_ids.to(device))
    unet.default_added_cond = (te.last_pooled.detach().clone(),
                               torch.tensor([[image_size, image_size, 0, 0, image_size, image_size]], dtype=torch.float32, device=device))
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)
    return StableDiffusionPipeline(unet, vae, te, tok, sched, device)
