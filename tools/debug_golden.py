import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/oracle", ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import numpy as np, torch
import cases
from _util import case_inputs, case_gout, rel_err, rel_l2, load
from test_controller_parity import _make_hip_controller, _run_hip, _pad64, _scale_weights
for name, case in cases.CONTROLLER_CASES.items():
    if case["cfg"] or case["kind"] != "edit": continue
    g = load("G6_" + name)
    q, k, v, mask, coords = case_inputs(case)
    D, f = case["D"], case["f"]
    c = _make_hip_controller(case, mask); _scale_weights(c, 64.0 / D)
    gout = case_gout(case, g["out"].shape)
    res = _run_hip(c, case, _pad64(q), _pad64(k), _pad64(v), coords, D ** -0.5, _pad64(gout))
    a, b = res["dq"][f:, :, :D], torch.from_numpy(g["dq"])[f:]
    err = (a - b).abs()
    rows = err.amax(-1)
    top = torch.topk(rows.reshape(-1), 5)
    print(f"{name}: dq rel_max={rel_err(a,b):.3e} rel_l2={rel_l2(a,b):.3e} ref_max={float(b.abs().max()):.3e}; worst rows {[(int(i)//rows.shape[1], int(i)%rows.shape[1]) for i in top.indices]} errs {[round(float(x),5) for x in top.values]}", flush=True)
    m_inp = c.masks_cache_dict[case['S']]["m_inp"].cpu()
    print("   worst rows in inpaint mask:", [int(m_inp[int(i) % rows.shape[1]]) for i in top.indices], " n_inp", int(m_inp.sum()))
