import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import test_controller_parity as T
from _util import case_gout, case_inputs, rel_err, rel_l2, removal_consistency
name = sys.argv[1] if len(sys.argv) > 1 else "edit_cross_opt_32_d64"
case = T.ORACLE_CASES.get(name) or T.SDXL_CASES[name]
dtype = torch.bfloat16
for pre in (False, True):
    q, k, v, mask, coords = case_inputs(case)
    q, k, v = (t.to(dtype).float() for t in (q, k, v))
    D = case["D"]; scale = D ** -0.5; cfac = scale * 1.4426950408889634
    q_dev = q
    if pre:
        q_dev = (q * cfac).to(dtype).float(); q = q_dev / cfac
    co, qo, ko, out_ref = T._oracle_run(case, q, k, v, mask, coords, scale, None, nn_ties="index")
    gout = case_gout(case, out_ref.shape)
    ch = T._make_hip_controller(case, mask)
    T._prebuild_tables(ch, case, q, coords, dtype, inject_topk=False)
    if pre: ch.q_scaled_hm = True
    res = T._run_hip(ch, case, q_dev, k, v, coords, scale, gout, dtype)
    kind = "cross" if case["cross"] else "self"
    S, f = case["S"], case["f"]
    tab = ch.masks_cache_dict[S]
    same, rm_exp = removal_consistency(ch._last_removal_aux, co.aux, S, f, tab["s_inp"], tie_tol=1.6e-2)
    print("pre", pre, "out err", rel_err(res["out"], out_ref.detach()), "same", same, "rm_expected", rm_exp, "oracle", {k2: float(v2) for k2, v2 in co.loss_log_dict[kind].items()},
          "hip", {k2: float(v2) for k2, v2 in ch.loss_log_dict[kind].items()})
