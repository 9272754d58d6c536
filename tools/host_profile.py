"""cProfile of one steady-state edit: where does the HOST spend its time?  (development aid)"""
import os, sys, cProfile, pstats, torch
torch.backends.cudnn.benchmark = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd import editor
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.synthetic import editor_kwargs, make_edit
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
def one(j):
    image, depth, mask, T = make_edit(j, kind="rotate")
    kw = editor_kwargs(); kw.update(ldm_stable_model=pipe, tokenizer_model=tok, scheduler_in=sched)
    editor.run_geodiffuser(image, depth, mask, T, **kw); torch.cuda.synchronize()
for j in range(3): one(j)
pr = cProfile.Profile(); pr.enable(); one(5); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io; buf = io.StringIO(); st.stream = buf; st.print_stats(70); out = buf.getvalue()
for l in out.split("\n"):
    if "geodiffuser_amd" in l or "bench" in l or "synthetic" in l or "tottime" in l or "{method" in l or "built-in" in l:
        print(l[:200])
