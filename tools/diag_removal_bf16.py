"""Why is the self-attention removal term 0 in bf16 at 512^2 / full width?  Prints what every removal_fwd call of the first optimisation
pass sees (development aid)."""
import os, sys
os.environ["GD_GRAPHS"] = "0"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import cases
from geodiffuser_amd import editor, ops
from geodiffuser_amd.attention_processors import AttentionGeometryEdit
from geodiffuser_amd.generic_torch import torch_erode
from geodiffuser_amd.diffusion import load_model
dtype = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float16
c = cases.LOOP_CFG1; inp = cases.loop_inputs(c)
p, tok, sched = load_model(device="cuda:0", tiny=False, dtype=dtype)
orig = ops.removal_fwd
calls = []
def spy(Pe, Pb, m_inp, m_wo, rows, S, n_valid=None):
    aux, rm = orig(Pe, Pb, m_inp, m_wo, rows, S, n_valid=n_valid)
    nv = int(n_valid.item()) if n_valid is not None else rows.numel()
    pi, pw, w = aux["p_in"][:, :nv].float(), aux["p_wo"][:, :nv].float(), aux["wgt"][:, :nv].float()
    if nv == 0:
        print(f'S={S} M={Pe.shape[2]} R={rows.numel()} n_valid=0 rm={float(rm):.5f}')
    elif len(calls) < 24:
        print(f"S={S} M={Pe.shape[2]} R={rows.numel()} n_valid={nv} rm={float(rm):.5f} p_in[min,mean,max]=({float(pi.min()):.3e},{float(pi.mean()):.3e},{float(pi.max()):.3e}) "
              f"p_wo=({float(pw.min()):.3e},{float(pw.mean()):.3e},{float(pw.max()):.3e}) wgt mean={float(w.mean()):.3e} Pe[sum over keys, mean]={float(Pe[:, :nv].float().sum(-1).mean()):.4f} nan={bool(torch.isnan(pi).any())}")
    calls.append(float(rm))
    return aux, rm
ops.removal_fwd = spy
import geodiffuser_amd.attention_processors as AP
lw = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30.0, "amodal": 80.5}, "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15.0, "amodal": 3.5}}
ctrl = AttentionGeometryEdit(["", ""], c["steps"], {"default_": c["cross_replace"]}, c["self_replace"], image_mask=inp["mask"], obj_edit_step=c["obj_edit_step"], device="cuda:0")
ctrl.amodal_mask = torch_erode(torch.from_numpy(cases.amodal_input(inp["mask"], *c.get("amodal_shift", (32, -12)))))
ctrl.default_loss_weights = lw; ctrl.initialize_default_loss_weights()
editor.NUM_DDIM_STEPS, editor.GUIDANCE_SCALE, editor.SKIP_OPTIM_STEPS = c["steps"], c["guidance"], c["skip_optim"]
ddim = [torch.from_numpy(a).to("cuda").to(dtype) for a in inp["ddim_latents"]]
lat, _, log = editor.text2image_ldm_stable(p, ["", ""], ctrl, latent=torch.from_numpy(inp["x_T"]).to("cuda").to(dtype), num_inference_steps=c["steps"], guidance_scale=c["guidance"],
    uncond_embeddings=None, transform_coordinates=torch.from_numpy(inp["coords"]), mask_obj=torch.from_numpy(inp["mask"]), optimize_steps=c["optimize_steps"], latent_replace=c["latent_replace"],
    lr=c["lr"], optimize_embeddings=True, optimize_latents=True, ddim_latents=ddim, ddim_noise=None, edit_type="geometry_editor", fast_start_steps=0.0, num_first_optim_steps=1,
    use_adaptive_optimization=True, return_type="latents", image_size=c["size"])
first = sorted(log)[0]
print("first pass log:", log[first])
