"""Which kernels make up one optimisation pass (forward with losses + autograd), full-size model (development aid)."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ["GD_GRAPHS"] = "0"
torch.backends.cudnn.benchmark = True
import cases
from geodiffuser_amd import graphs
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import AttentionGeometryEdit, register_attention_control_diffusers, set_attn_processor_for_edit
from geodiffuser_amd.editor import clear_controller_loss
from geodiffuser_amd.generic_torch import torch_erode
from geodiffuser_amd.synthetic import make_edit
from geodiffuser_amd import vis_utils
from torch.profiler import profile, ProfilerActivity
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
image, depth, mask, T = make_edit(0, kind="rotate")
coords, _, amodal = vis_utils.get_transform_coordinates(image, depth, mask, transform_in=T, return_mesh=True, as_torch=True)
c = AttentionGeometryEdit(["", ""], 50, {"default_": 0.95}, 0.95, image_mask=mask, obj_edit_step=0.9, device="cuda:0")
c.amodal_mask = torch_erode(amodal.float().cpu()) if torch.is_tensor(amodal) else torch_erode(torch.from_numpy(amodal))
coords = coords[None] if coords.dim() == 3 else coords
register_attention_control_diffusers(pipe, c, coords)
sched.set_timesteps(50)
lat = torch.randn(2, 4, 64, 64, device="cuda", dtype=torch.bfloat16)
ctx = torch.randn(4, 77, 1024, device="cuda", dtype=torch.bfloat16)
op = graphs.GraphedOptPass(pipe, coords, 3.0)
def one():
    clear_controller_loss(c)
    set_attn_processor_for_edit(pipe, coords_base=(0, 1), coords_edit=(1, 2), use_cfg=False)
    op.grads(c, lat, ctx, 981)
    c.cur_step -= 1
for _ in range(3):
    one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA or getattr(e, "self_device_time_total", 0) > 0]
rows = sorted(((e.self_device_time_total, e.count, e.key) for e in prof.key_averages() if e.self_device_time_total > 0), reverse=True)
tot = sum(r[0] for r in rows)
print(f"total device time {tot/1e3:.2f} ms, {sum(r[1] for r in rows)} kernels/ops")
for t, n, k in rows[:70]:
    print(f"{t/1e3:8.3f} ms {n:5d}  {k[:130]}")
