"""Where does a process's FIRST edit go?  Times the first and second edit of a fresh process and, inside them, every hipGraph capture
(torch.cuda.graph enter -> exit incl. instantiation), the MIOpen-bound first convolution calls and everything else.
    python tools/warmup_probe.py            (uses the committed find-db through geodiffuser_amd.miopen_cache)"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.backends.cudnn.benchmark = True
from geodiffuser_amd import miopen_cache
print("miopen db:", miopen_cache.configure())
from geodiffuser_amd import _lib, editor
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.synthetic import editor_kwargs, make_edit

acc = collections.Counter(); cnt = collections.Counter()
_enter, _exit = torch.cuda.graph.__enter__, torch.cuda.graph.__exit__
def enter(self):
    torch.cuda.synchronize(); self._t0 = time.perf_counter(); return _enter(self)
def exit_(self, *a):
    r = _exit(self, *a); torch.cuda.synchronize(); acc["graph capture+instantiate"] += time.perf_counter() - self._t0; cnt["graph capture+instantiate"] += 1; return r
torch.cuda.graph.__enter__, torch.cuda.graph.__exit__ = enter, exit_
_replay = torch.cuda.CUDAGraph.replay
def replay(self):
    first = not getattr(self, "_seen", False)
    if first:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    r = _replay(self)
    if first:
        torch.cuda.synchronize(); acc["first replay of a graph"] += time.perf_counter() - t0; cnt["first replay of a graph"] += 1; self._seen = True
    return r
torch.cuda.CUDAGraph.replay = replay
_conv = torch.nn.functional.conv2d
seen = set()
def conv2d(x, w, *a, **k):
    if torch.cuda.is_current_stream_capturing():
        return _conv(x, w, *a, **k)
    key = (tuple(x.shape), tuple(w.shape), x.requires_grad, tuple(v for v in a if not torch.is_tensor(v)), tuple(sorted((kk, vv) for kk, vv in k.items() if not torch.is_tensor(vv))))
    if key in seen:
        return _conv(x, w, *a, **k)
    seen.add(key); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = _conv(x, w, *a, **k); torch.cuda.synchronize(); acc["first conv2d of a shape (fwd)"] += time.perf_counter() - t0; cnt["first conv2d of a shape (fwd)"] += 1
    return r
torch.nn.functional.conv2d = conv2d
_lin = torch.nn.functional.linear
seen_l = set()
def linear(x, w, b=None):
    if torch.cuda.is_current_stream_capturing():
        return _lin(x, w, b)
    key = (tuple(x.shape), tuple(w.shape))
    if key in seen_l:
        return _lin(x, w, b)
    seen_l.add(key); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = _lin(x, w, b); torch.cuda.synchronize(); acc["first linear of a shape"] += time.perf_counter() - t0; cnt["first linear of a shape"] += 1
    return r
torch.nn.functional.linear = linear

_lib.load()
t0 = time.perf_counter()
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
torch.cuda.synchronize(); print(f"load_model (random init of the SD2.1-shaped weights): {time.perf_counter() - t0:.1f} s")
for j in range(3):
    image, depth, mask, T = make_edit(1000 + j, size=512, kind="rotate")
    kw = editor_kwargs(); kw.update(num_ddim_steps=50, ldm_stable_model=pipe, tokenizer_model=tok, scheduler_in=sched)
    acc.clear(); cnt.clear()
    t0 = time.perf_counter(); editor.run_geodiffuser(image, depth, mask, T, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"edit {j}: {dt:.2f} s")
    for k, v in acc.most_common():
        print(f"    {k}: {v:.2f} s over {cnt[k]} events")
    print(f"    everything else: {dt - sum(acc.values()):.2f} s")
