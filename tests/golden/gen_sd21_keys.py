"""Parameter names and shapes of the PUBLIC Stable Diffusion 2.1-base checkpoint files, generated from the public model configs —
independently of geodiffuser_amd (nothing of the package is imported): what `unet/diffusion_pytorch_model.safetensors`,
`vae/diffusion_pytorch_model.safetensors` and `text_encoder/model.safetensors` of `stabilityai/stable-diffusion-2-1-base` contain.

    python tests/golden/gen_sd21_keys.py            # rewrites tests/golden/sd21_base_keys.json

Sources (public; none of it is in /root/reference, which only calls `StableDiffusionPipeline.from_pretrained`,
GeoDiffuser/utils/diffusion.py:99-140): the repository's `unet/config.json`, `vae/config.json` (AutoencoderKL:
block_out_channels 128/256/512/512, layers_per_block 2, latent_channels 4, 32 norm groups), `text_encoder/config.json` (CLIPTextModel:
hidden 1024, 23 layers, 16 heads, intermediate 4096, vocab 49408, 77 positions) and the naming rules of diffusers' UNet2DConditionModel /
AutoencoderKL and transformers' CLIPTextModel.  UNPINNED against those libraries (neither is importable with weights here): the check this
file supports is "the harness's modules expose exactly these names and shapes, and the loader consumes exactly these names" — plus the
public parameter counts asserted below (UNet 865,910,724; VAE 83,653,863; text encoder 340,387,840).
"""
import json
import os


def _lin(p, cin, cout, bias=True):
    d = {p + ".weight": [cout, cin]}
    if bias:
        d[p + ".bias"] = [cout]
    return d


def _conv(p, cin, cout, k):
    return {p + ".weight": [cout, cin, k, k], p + ".bias": [cout]}


def _norm(p, c):
    return {p + ".weight": [c], p + ".bias": [c]}


def _resnet(p, cin, cout, temb=None):
    d = {}
    d.update(_norm(p + ".norm1", cin)); d.update(_conv(p + ".conv1", cin, cout, 3))
    if temb:
        d.update(_lin(p + ".time_emb_proj", temb, cout))
    d.update(_norm(p + ".norm2", cout)); d.update(_conv(p + ".conv2", cout, cout, 3))
    if cin != cout:
        d.update(_conv(p + ".conv_shortcut", cin, cout, 1))
    return d


def _transformer(p, c, ctx, depth=1, linear_proj=True):
    d = {}
    d.update(_norm(p + ".norm", c))
    d.update(_lin(p + ".proj_in", c, c) if linear_proj else _conv(p + ".proj_in", c, c, 1))
    for b in range(depth):
        q = f"{p}.transformer_blocks.{b}"
        for name, kv in (("attn1", c), ("attn2", ctx)):
            d.update(_lin(f"{q}.{name}.to_q", c, c, bias=False)); d.update(_lin(f"{q}.{name}.to_k", kv, c, bias=False))
            d.update(_lin(f"{q}.{name}.to_v", kv, c, bias=False)); d.update(_lin(f"{q}.{name}.to_out.0", c, c))
        for n in ("norm1", "norm2", "norm3"):
            d.update(_norm(f"{q}.{n}", c))
        d.update(_lin(f"{q}.ff.net.0.proj", c, 8 * c)); d.update(_lin(f"{q}.ff.net.2", 4 * c, c))
    d.update(_lin(p + ".proj_out", c, c) if linear_proj else _conv(p + ".proj_out", c, c, 1))
    return d


def unet_keys(ch=(320, 640, 1280, 1280), ctx=1024, layers=2, attn=(True, True, True, False), in_ch=4, out_ch=4, linear_proj=True):
    """UNet2DConditionModel: CrossAttnDownBlock2D x3 + DownBlock2D, UNetMidBlock2DCrossAttn, UpBlock2D + CrossAttnUpBlock2D x3."""
    temb = 4 * ch[0]
    d = {}
    d.update(_conv("conv_in", in_ch, ch[0], 3))
    d.update(_lin("time_embedding.linear_1", ch[0], temb)); d.update(_lin("time_embedding.linear_2", temb, temb))
    cin = ch[0]
    skips = [ch[0]]
    for i, c in enumerate(ch):
        for j in range(layers):
            d.update(_resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else c, c, temb))
            if attn[i]:
                d.update(_transformer(f"down_blocks.{i}.attentions.{j}", c, ctx, linear_proj=linear_proj))
            skips.append(c)
        if i < len(ch) - 1:
            d.update(_conv(f"down_blocks.{i}.downsamplers.0.conv", c, c, 3))
            skips.append(c)
        cin = c
    d.update(_resnet("mid_block.resnets.0", ch[-1], ch[-1], temb)); d.update(_transformer("mid_block.attentions.0", ch[-1], ctx, linear_proj=linear_proj))
    d.update(_resnet("mid_block.resnets.1", ch[-1], ch[-1], temb))
    prev = ch[-1]
    for i, c in enumerate(reversed(ch)):
        lvl = len(ch) - 1 - i
        for j in range(layers + 1):
            d.update(_resnet(f"up_blocks.{i}.resnets.{j}", prev + skips.pop(), c, temb))
            if attn[lvl]:
                d.update(_transformer(f"up_blocks.{i}.attentions.{j}", c, ctx, linear_proj=linear_proj))
            prev = c
        if i < len(ch) - 1:
            d.update(_conv(f"up_blocks.{i}.upsamplers.0.conv", c, c, 3))
    d.update(_norm("conv_norm_out", ch[0])); d.update(_conv("conv_out", ch[0], out_ch, 3))
    return d


def _vae_attn(p, c):
    d = _norm(p + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        d.update(_lin(f"{p}.{n}", c, c))
    return d


def vae_keys(ch=(128, 256, 512, 512), latent=4, layers=2):
    d = {}
    d.update(_conv("encoder.conv_in", 3, ch[0], 3))
    cin = ch[0]
    for i, c in enumerate(ch):
        for j in range(layers):
            d.update(_resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else c, c))
        if i < len(ch) - 1:
            d.update(_conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3))
        cin = c
    for side, c in (("encoder", ch[-1]), ("decoder", ch[-1])):
        d.update(_resnet(f"{side}.mid_block.resnets.0", c, c)); d.update(_vae_attn(f"{side}.mid_block.attentions.0", c))
        d.update(_resnet(f"{side}.mid_block.resnets.1", c, c))
    d.update(_norm("encoder.conv_norm_out", ch[-1])); d.update(_conv("encoder.conv_out", ch[-1], 2 * latent, 3))
    d.update(_conv("quant_conv", 2 * latent, 2 * latent, 1)); d.update(_conv("post_quant_conv", latent, latent, 1))
    d.update(_conv("decoder.conv_in", latent, ch[-1], 3))
    cin = ch[-1]
    for i, c in enumerate(reversed(ch)):
        for j in range(layers + 1):
            d.update(_resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else c, c))
        if i < len(ch) - 1:
            d.update(_conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3))
        cin = c
    d.update(_norm("decoder.conv_norm_out", ch[0])); d.update(_conv("decoder.conv_out", ch[0], 3, 3))
    return d


def clip_text_keys(width=1024, layers=23, inter=4096, vocab=49408, positions=77):
    d = {"text_model.embeddings.token_embedding.weight": [vocab, width], "text_model.embeddings.position_embedding.weight": [positions, width]}
    for i in range(layers):
        p = f"text_model.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            d.update(_lin(f"{p}.self_attn.{n}", width, width))
        d.update(_norm(p + ".layer_norm1", width)); d.update(_norm(p + ".layer_norm2", width))
        d.update(_lin(p + ".mlp.fc1", width, inter)); d.update(_lin(p + ".mlp.fc2", inter, width))
    d.update(_norm("text_model.final_layer_norm", width))
    return d


def count(d):
    n = 0
    for s in d.values():
        k = 1
        for x in s:
            k *= x
        n += k
    return n


if __name__ == "__main__":
    out = {"unet": unet_keys(), "vae": vae_keys(), "text_encoder": clip_text_keys()}
    counts = {k: count(v) for k, v in out.items()}
    assert counts == {"unet": 865910724, "vae": 83653863, "text_encoder": 340387840}, counts       # the public parameter counts
    out["_counts"] = counts
    out["_source"] = "tests/golden/gen_sd21_keys.py (public configs of stabilityai/stable-diffusion-2-1-base; unpinned against diffusers / transformers)"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sd21_base_keys.json")
    json.dump(out, open(path, "w"), sort_keys=True, separators=(",", ":"))
    print(path, counts, {k: len(v) for k, v in out.items() if not k.startswith("_")})
