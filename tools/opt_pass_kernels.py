"""Which kernels does ONE optimisation pass (forward with losses + backward to latent and text embedding) launch?  (torch.profiler on the
eager path; development aid)"""
import os, sys
os.environ["GD_GRAPHS"] = "0"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases
from geodiffuser_amd import editor, graphs
from geodiffuser_amd.attention_processors import AttentionGeometryEdit, register_attention_control_diffusers, set_attn_processor_for_edit
from geodiffuser_amd.generic_torch import torch_erode
from geodiffuser_amd.diffusion import load_model
from torch.profiler import profile, ProfilerActivity
dtype = torch.bfloat16
p, tok, sched = load_model(device="cuda:0", dtype=dtype)
mask = cases.ellipse_mask(); coords = torch.from_numpy(cases.make_coords("rotate", mask))
ctrl = AttentionGeometryEdit(["", ""], 50, {"default_": 0.95}, 0.95, image_mask=mask, obj_edit_step=0.9, device="cuda:0")
ctrl.amodal_mask = torch_erode(torch.from_numpy(cases.amodal_input(mask)))
lw = {"self": {"sim": 55, "movement": 30.5, "removal": 2.6, "smoothness": 30.0, "amodal": 80.5}, "cross": {"sim": 45, "movement": 30.34, "removal": 2.6, "smoothness": 15.0, "amodal": 3.5}}
ctrl.default_loss_weights = lw; ctrl.initialize_default_loss_weights()
register_attention_control_diffusers(p, ctrl, transform_coords=coords)
set_attn_processor_for_edit(p, coords_base=(0, 1), coords_edit=(1, 2), use_cfg=False)
from geodiffuser_amd.generic_torch import binarize_tensor, reshape_transform_coords
from geodiffuser_amd.warp_utils import warp_grid_edit
t_m = reshape_transform_coords(coords.to("cuda").float(), in_mat_shape=ctrl.image_mask.shape).tile(2, 1, 1, 1).half()
ctrl.mask_new_warped = binarize_tensor(warp_grid_edit(ctrl.image_mask[:, None].to("cuda").float(), t_m))
ids = tok(["", ""]).input_ids.to("cuda")
ctx0 = p.text_encoder(ids)[0].float()
lat0 = torch.randn(2, 4, 64, 64, device="cuda")
def one():
    ctrl.loss = 0.0; ctrl.initialize_loss_log_dict()
    lat = lat0.clone().requires_grad_(True); ctx = ctx0.clone().requires_grad_(True)
    with torch.enable_grad():
        p.unet(lat.to(dtype), 500, encoder_hidden_states=ctx.to(dtype))
        g = torch.autograd.grad(ctrl.loss, [lat, ctx])
    ctrl.cur_step -= 1
    return g
for _ in range(3): one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    one(); torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
ev.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in ev); n = sum(e.count for e in ev)
print(f"optimisation pass: {n} kernels, {tot / 1e3:.2f} ms of kernel time")
for e in ev[:60]:
    print(f"{e.count:5d} x {e.device_time_total / e.count:7.1f} us = {e.device_time_total / 1e3:7.3f} ms  {e.key[:120]}")

print("\n== aten::copy_ / contiguous / clone / to by input shape (where do the copy kernels come from) ==")
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::zeros", "aten::zero_", "aten::fill_") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print(f"{e.count:4d} x {e.device_time_total / max(1, e.count):6.1f} us  {e.key:18s} {str(e.input_shapes)[:110]}")

print("\n== elementwise / copy / fill / cat launches by the innermost geodiffuser_amd source line that issued them ==")
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof2:
    one(); torch.cuda.synchronize()
import collections
by_line = collections.defaultdict(lambda: [0, 0.0])
OPS = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::zero_", "aten::fill_", "aten::cat", "aten::sub", "aten::div", "aten::neg",
       "aten::silu", "aten::sum", "aten::index_put_", "aten::_to_copy", "aten::silu_backward", "aten::masked_fill_", "aten::where", "aten::sigmoid", "aten::exp",
       "aten::cos", "aten::sin", "aten::mean", "aten::native_dropout", "aten::gelu", "aten::slice_backward", "aten::select_backward", "aten::sum_to_size")
for e in prof2.events():
    if e.name in OPS and e.device_time_total > 0 and not any(c.name in OPS and c.device_time_total > 0 for c in e.cpu_children):
        where = "(autograd engine / no package frame)"
        for fr in e.stack:
            if "geodiffuser_amd" in fr:
                where = fr.split("geodiffuser_amd/")[-1]; break
        a = by_line[(where, e.name)]
        a[0] += len(e.kernels) or 1; a[1] += e.device_time_total
tot = sum(a[1] for a in by_line.values()); n = sum(a[0] for a in by_line.values())
print(f"{n} launches, {tot / 1e3:.2f} ms")
for (where, name), a in sorted(by_line.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{a[0]:4d} x {a[1] / a[0]:6.1f} us = {a[1] / 1e3:6.3f} ms  {name:22s} {where[:110]}")
