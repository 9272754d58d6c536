"""Kernels and host time of the geometry pre-pass (vis_utils.get_transform_coordinates) at 512^2 (development aid)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import vis_utils
from geodiffuser_amd.synthetic import make_edit
from torch.profiler import profile, ProfilerActivity
image, depth, mask, T = make_edit(3, size=512, kind="rotate")
def one():
    r = vis_utils.get_transform_coordinates(image / 255.0, depth, mask.numpy() if hasattr(mask, "numpy") else mask, transform_in=T)
    torch.cuda.synchronize(); return r
for _ in range(3): one()
t0 = time.perf_counter(); one(); print(f"wall {1e3 * (time.perf_counter() - t0):.2f} ms")
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    one()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
ev.sort(key=lambda e: -e.device_time_total)
print(f"{sum(e.count for e in ev)} kernels, {sum(e.device_time_total for e in ev) / 1e3:.2f} ms of kernel time")
for e in ev[:14]:
    print(f"{e.count:4d} x {e.device_time_total / e.count:8.1f} us = {e.device_time_total / 1e3:7.3f} ms  {e.key[:100]}")
cpu = sorted([e for e in prof.key_averages() if e.device_type.name == "CPU"], key=lambda e: -e.self_cpu_time_total)[:8]
for e in cpu:
    print(f"  host {e.self_cpu_time_total / 1e3:7.2f} ms  x{e.count:4d} {e.key[:80]}")
