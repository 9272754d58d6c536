"""What does MIOpen do on the first call of a convolution shape when the find-db already has it?  (development aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "1") == "1"
from geodiffuser_amd import miopen_cache
print("db:", miopen_cache.configure(), "find mode:", os.environ.get("MIOPEN_FIND_MODE"), "benchmark:", torch.backends.cudnn.benchmark)
dev = "cuda"
shapes = [(1, 320, 64, 320), (1, 640, 32, 640), (3, 320, 64, 320), (1, 1280, 16, 1280), (3, 960, 64, 320), (1, 2560, 8, 1280)]
for (b, cin, hw, cout) in shapes:
    x = torch.randn(b, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = torch.nn.functional.conv2d(x, w, None, padding=1); torch.cuda.synchronize(); t1 = time.perf_counter()
    y = torch.nn.functional.conv2d(x, w, None, padding=1); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"conv {b}x{cin}x{hw}x{hw} -> {cout}: first {1e3 * (t1 - t0):8.1f} ms, second {1e3 * (t2 - t1):6.2f} ms", flush=True)
