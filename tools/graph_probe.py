"""Feasibility probe: capture one UNet pass (vanilla HIP attention processors) in a hipGraph via torch.cuda.graph."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd.diffusion import load_model
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
unet = pipe.unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.bfloat16)
ctx = torch.randn(B, 77, 1024, device="cuda", dtype=torch.bfloat16)
t = torch.tensor([500], device="cuda")
with torch.no_grad():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            y = unet(x, t, encoder_hidden_states=ctx)["sample"]
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = unet(x, t, encoder_hidden_states=ctx)["sample"]
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    ref = y.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yg = unet(x, t, encoder_hidden_states=ctx)["sample"]
    g.replay(); torch.cuda.synchronize()
    print("max diff graph vs eager", float((yg.float() - ref.float()).abs().max()))
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 10
    # new input
    x.copy_(torch.randn_like(x)); t.fill_(480)
    g.replay(); torch.cuda.synchronize()
    y2 = unet(x, t, encoder_hidden_states=ctx)["sample"]
    print("max diff after input change", float((yg.float() - y2.float()).abs().max()))
print(f"B={B} eager {eager*1e3:.2f} ms  graph {graph*1e3:.2f} ms")
