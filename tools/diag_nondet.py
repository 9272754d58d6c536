"""Which modules of the UNet give different bits on two identical passes / two identical batch rows?  (development aid)"""
import os, sys, torch, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
tiny = "--tiny" in sys.argv
dt = torch.bfloat16
p, _, _ = load_model(device="cuda:0", tiny=tiny, dtype=dt)
p.unet.set_attn_processor(VanillaAttentionProcessor())
torch.manual_seed(0)
S = 32 if tiny else 64
x1 = torch.randn(1, 4, S, S, device="cuda").to(dt); c1 = torch.randn(1, 77, 64 if tiny else 1024, device="cuda").to(dt)
x = x1.expand(2, -1, -1, -1).contiguous(); c = c1.expand(2, -1, -1).contiguous()
rec = []
def hook(name):
    def f(m, i, o):
        if torch.is_tensor(o): rec[-1][name] = (type(m).__name__, o.detach().clone(), [t.detach().clone() for t in i if torch.is_tensor(t)])
    return f
for n, m in p.unet.named_modules():
    if n: m.register_forward_hook(hook(n))
for rep in range(2):
    rec.append(collections.OrderedDict())
    with torch.no_grad(): p.unet(x, 500, encoder_hidden_states=c)
a, b = rec
bad_pass = collections.Counter(); bad_row = collections.Counter(); tot = collections.Counter()
first = None
for n in a:
    ty, oa, ia = a[n]; _, ob, ib = b[n]
    tot[ty] += 1
    same_in = all(torch.equal(u, v) for u, v in zip(ia, ib))
    if same_in and not torch.equal(oa, ob):
        bad_pass[ty] += 1
        if first is None: first = (n, ty)
    rows_in_same = all(u.shape[0] != 2 or torch.equal(u[0], u[1]) for u in ia)
    if rows_in_same and oa.shape[0] == 2 and not torch.equal(oa[0], oa[1]): bad_row[ty] += 1
shown = 0
for n in a:
    ty, oa, ia = a[n]; _, ob, ib = b[n]
    same_in = all(torch.equal(u, v) for u, v in zip(ia, ib))
    if same_in and not torch.equal(oa, ob) and shown < 8:
        print("  same input, different output:", n, ty, f"rel {float((oa.float()-ob.float()).norm()/ob.float().norm()):.1e}"); shown += 1
print("module types:", dict(tot))
print("same input, different output across two passes:", dict(bad_pass), "first:", first)
print("identical input rows, different output rows:", dict(bad_row))
