"""Wall-clock breakdown of one edit by phase (development aid; adds synchronisations)."""
import os, sys, time, torch, collections
torch.backends.cudnn.benchmark = os.environ.get('GD_MIOPEN_FIND', '1') == '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import miopen_cache; miopen_cache.configure()
from geodiffuser_amd import editor, inversion, diffusion, optimization, vis_utils
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.synthetic import editor_kwargs, make_edit
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(mod, name, key=None):
    orig = getattr(mod, name)
    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize(); kk = key(a, k) if callable(key) else (key or name)
        acc[kk] += time.perf_counter() - t0; cnt[kk] += 1
        return r
    setattr(mod, name, f)
wrap(inversion.NullInversion, "ddim_loop")
wrap(inversion.NullInversion, "image2latent")
wrap(editor, "diffusion_step", key=lambda a, k: "diffusion_step_opt_fwd" if k.get("use_cfg", True) is False else "diffusion_step_cfg")
from geodiffuser_amd import graphs
wrap(graphs.GraphedOptPass, "grads")
wrap(editor, "_apply_latent_update")
wrap(editor, "latent2image")
wrap(editor.vis_utils, "get_transform_coordinates")
wrap(editor, "masked_histogram_matching")
wrap(editor, "convert_loss_log_to_numpy")
kw = editor_kwargs(); kw.update(ldm_stable_model=pipe, tokenizer_model=tok, scheduler_in=sched)
for it in range(3):
    acc.clear(); cnt.clear()
    image, depth, mask, T = make_edit(it, kind="rotate")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    editor.run_geodiffuser(image, depth, mask, T, **kw)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"total {tot:.3f} s")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {v:7.3f} s  x{cnt[k]:3d}  {1e3*v/cnt[k]:8.2f} ms each")
print(f"  other {tot - sum(acc.values()):.3f}")
