import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import test_controller_parity as T
from geodiffuser_amd import _lib
lib = _lib.load()
case = T.SDXL_CASES["edit_self_opt_32_f20"]
dtype = torch.bfloat16
q, k, v, mask, coords = T.case_inputs(case)
q, k, v = (t.to(dtype).float() for t in (q, k, v))
f, D = case["f"], case["D"]
co, qo, ko, out_ref = T._oracle_run(case, q, k, v, mask, coords, D ** -0.5, None)
gout = T.case_gout(case, out_ref.shape)
print("oracle sim", float(co.loss_log_dict["self"]["sim"]))
for cfg in ((4, 1), (8, 1), (4, 1), (8, 1)):
    lib.gd_attn_fwd_set_config(*cfg)
    ch = T._make_hip_controller(case, mask)
    T._prebuild_tables(ch, case, q, coords, dtype)
    res = T._run_hip(ch, case, q, k, v, coords, D ** -0.5, gout, dtype)
    e = (res["out"] - out_ref.detach()).abs()
    print(cfg, "out rel err", float(e.max() / out_ref.abs().max()), "per block of f rows:", [round(float(e[i * f:(i + 1) * f].max()), 5) for i in range(e.shape[0] // f)],
          {kk: round(float(vv), 6) for kk, vv in ch.loss_log_dict["self"].items()})
