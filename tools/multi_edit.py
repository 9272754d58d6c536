"""Run several edits back to back in one process (graph reuse across edits, allocator state): python tools/multi_edit.py 0 1 2 3"""
import torch, time, sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
torch.backends.cudnn.benchmark = True
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd import editor, graphs
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.synthetic import editor_kwargs, make_edit
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
for it in [int(a) for a in sys.argv[1:]]:
    kw = editor_kwargs(); kw.update(num_ddim_steps=int(os.environ.get('DBG_STEPS', '50')), ldm_stable_model=pipe, tokenizer_model=tok, scheduler_in=sched)
    image, depth, mask, T = make_edit(it, kind=os.environ.get("KIND", "rotate"))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, log = editor.run_geodiffuser(image, depth, mask, T, return_loss_log_dict=True, **kw)
    torch.cuda.synchronize()
    if os.environ.get('DBG_GC') == '1':
        import gc; gc.collect(); torch.cuda.empty_cache()
    print(it, f"{time.perf_counter()-t0:.3f}s", f"reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", "opt graphs:", len(graphs._OPT_GRAPHS), [k[2] for k in graphs._OPT_GRAPHS], flush=True)
print("done", flush=True)
