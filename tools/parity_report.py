"""Controller-level parity report: HIP controller vs the CPU oracle at head dim 64 for every regime, in fp16 and bf16, printing the measured
errors (development aid; the bounds asserted in tests/test_controller_parity.py are set from this table)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import test_controller_parity as T
from _util import case_gout, case_inputs, rel_err, rel_l2

print("| case | dtype | out rel-max | loss rel | dq rel-L2 | dq rel-max | dk rel-L2 |")
print("|---|---|---|---|---|---|---|")
for name, case in T.ORACLE_CASES.items():
    for dtype in (torch.float16, torch.bfloat16):
        q, k, v, mask, coords = case_inputs(case)
        q, k, v = (t.to(dtype).float() for t in (q, k, v))
        f, D = case["f"], case["D"]
        co, qo, ko, out_ref = T._oracle_run(case, q, k, v, mask, coords, D ** -0.5, None)
        gout = case_gout(case, out_ref.shape)
        ch = T._make_hip_controller(case, mask)
        T._prebuild_tables(ch, case, q, coords, dtype)
        res = T._run_hip(ch, case, q, k, v, coords, D ** -0.5, gout, dtype)
        row = [name, str(dtype).split(".")[1], f"{rel_err(res['out'], out_ref.detach()):.2e}"]
        if not case["cfg"]:
            e0 = co.coords_edit[0]
            total = (out_ref[e0 * f:] * gout[e0 * f:]).sum()
            if torch.is_tensor(co.loss):
                total = total + co.loss
                row.append(f"{abs(res['loss'] - float(co.loss)) / max(abs(float(co.loss)), 1e-9):.2e}")
            else:
                row.append("-")
            dq, dk = torch.autograd.grad(total, [qo, ko], allow_unused=True)
            row += [f"{rel_l2(res['dq'][e0 * f:], dq[e0 * f:]):.2e}", f"{rel_err(res['dq'][e0 * f:], dq[e0 * f:]):.2e}"]
            row.append(f"{rel_l2(res['dk'][e0 * f:], dk[e0 * f:]):.2e}" if (dk is not None and case["cross"] and case["kind"] == "edit") else "-")
        else:
            row += ["-", "-", "-", "-"]
        print("| " + " | ".join(row) + " |", flush=True)

print("\n## reference-recorded fixtures (G6: fp32 inputs on the reference side, fp16-rounded inputs on the device; bounds TOLS_GOLDEN)\n")
print("| fixture | dq rel-L2 | dq rel-max | arg-max identical |")
print("|---|---|---|---|")
import cases
for name, case in cases.CONTROLLER_CASES.items():
    if case["cfg"]:
        continue
    n0 = len(T.MEASURED)
    try:
        T.test_controller_vs_golden(name)
        st = ""
    except AssertionError as e:
        st = " (ASSERTION FAILED)"
    if len(T.MEASURED) > n0:
        m = T.MEASURED[-1]
        print(f"| G6_{name}{st} | {m['dq_l2']:.2e} | {m['dq_max']:.2e} | {m['same_argmax']} |", flush=True)
