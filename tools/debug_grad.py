import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/oracle", ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import numpy as np, torch
import cases, ref_cpu as O
from _util import case_inputs, case_gout, rel_err, rel_l2, warped_mask
from test_controller_parity import _make_hip_controller, _make_oracle_controller, _run_hip, ORACLE_CASES

for name in sys.argv[1:]:
    case = ORACLE_CASES[name]
    q, k, v, mask, coords = case_inputs(case)
    f, D = case["f"], case["D"]; scale = D ** -0.5
    co = _make_oracle_controller(case, mask)
    qo, ko = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
    out_ref = co(qo, ko, v, case["cross"], "up", transform_coords=coords, scale=scale)
    gout = case_gout(case, out_ref.shape)
    e0 = 1
    # per-term gradients in the oracle
    lw = co.loss_weight_dict["cross" if case["cross"] else "self"]
    parts = {}
    parts["gout"] = torch.autograd.grad((out_ref[e0 * f:] * gout[e0 * f:]).sum(), qo, retain_graph=True)[0][e0 * f:]
    parts["loss"] = torch.autograd.grad(co.loss, qo, retain_graph=True)[0][e0 * f:]
    for wzero in (None, "removal", "sim", "movement", "smoothness", "amodal"):
        ch = _make_hip_controller(case, mask)
        if wzero is not None:
            for kk in ch.loss_weight_dict["self"]:
                if kk != wzero: ch.loss_weight_dict["self"][kk] = 0.0; ch.loss_weight_dict["cross"][kk] = 0.0
        co2 = _make_oracle_controller(case, mask)
        if wzero is not None:
            for kk in co2.loss_weight_dict["self"]:
                if kk != wzero: co2.loss_weight_dict["self"][kk] = 0.0; co2.loss_weight_dict["cross"][kk] = 0.0
        qo2, ko2 = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
        o2 = co2(qo2, ko2, v, case["cross"], "up", transform_coords=coords, scale=scale)
        ref = torch.autograd.grad(co2.loss, [qo2, ko2], allow_unused=True)
        qd, kd, vd = (t.half().cuda().contiguous() for t in (q, k, v))
        qd.requires_grad_(True); kd.requires_grad_(True)
        with torch.enable_grad():
            out = ch(qd, kd, vd, is_cross=case["cross"], place_in_unet="up", transform_coords=coords, scale=scale)
        dq, dk = torch.autograd.grad(ch.loss, [qd, kd], allow_unused=True)
        a, b = dq.float().cpu()[e0 * f:], ref[0][e0 * f:]
        print(f"{name} only={wzero}: loss hip={float(ch.loss):.6f} ref={float(co2.loss):.6f}  dq rel_max={rel_err(a,b):.3e} rel_l2={rel_l2(a,b):.3e} |ref|max={float(b.abs().max()):.3e}", flush=True)
        if wzero == "removal":
            aux = co2.aux
            print("   oracle j_in[:8]", aux["j_in"][0, :8].tolist(), "j_wo[:8]", aux["j_wo"][0, :8].tolist())
