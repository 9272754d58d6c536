import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/oracle", ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import numpy as np, torch
import cases, ref_cpu as O
from _util import case_inputs, case_gout, rel_err, rel_l2, load
from test_controller_parity import _make_hip_controller, _make_oracle_controller, _run_hip, _pad64, _scale_weights
name = "rem_self_opt_32"; case = cases.CONTROLLER_CASES[name]
q, k, v, mask, coords = case_inputs(case)
D, f = case["D"], case["f"]
co = _make_oracle_controller(case, mask)
qo = q.clone().requires_grad_(True); ko = k.clone().requires_grad_(True)
out_ref = co(qo, ko, v, case["cross"], "up", transform_coords=coords, scale=D ** -0.5)
aux = co.aux
c = _make_hip_controller(case, mask); _scale_weights(c, 64.0 / D)
import geodiffuser_amd.ops as ops
orig = ops.removal_fwd
cap = {}
def wrap(*a, **kw):
    r = orig(*a, **kw); cap["aux"] = r[0]; cap["rm"] = r[1]; return r
ops.removal_fwd = wrap
gout = case_gout(case, out_ref.shape)
res = _run_hip(c, case, _pad64(q), _pad64(k), _pad64(v), coords, D ** -0.5, _pad64(gout))
h = cap["aux"]
print("n_inp", aux["j_in"].shape, "rm hip", float(cap["rm"]), " log removal hip", float(c.loss_log_dict["self"]["removal"]), "ref", float(co.loss_log_dict["self"]["removal"]))
print("j_in same frac", float((h["j_in"].cpu().long() == aux["j_in"]).float().mean()), "j_wo same frac", float((h["j_wo"].cpu().long() == aux["j_wo"]).float().mean()))
print("p_in rel", rel_err(h["p_in"].cpu(), aux["p_in"]), "p_wo rel", rel_err(h["p_wo"].cpu(), aux["p_wo"]))
print("p_in ref min/max", float(aux["p_in"].min()), float(aux["p_in"].max()), "p_wo ref min/max", float(aux["p_wo"].min()), float(aux["p_wo"].max()))
d = (h["p_wo"].cpu() - aux["p_wo"]).abs() / aux["p_wo"]
print("p_wo relerr per-elem max", float(d.max()), "p_in", float(((h["p_in"].cpu() - aux["p_in"]).abs() / aux["p_in"].clamp_min(1e-9)).max()))
