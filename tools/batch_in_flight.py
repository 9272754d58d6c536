"""The reference's batch driver (large_scale_editor) over a synthetic dataset of experiment folders, one process against several edits in flight
per GPU: wall time of the whole job including every rank's start-up.  Usage (GPU box): python tools/batch_in_flight.py [n_experiments] [P ...]"""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ps = [int(a) for a in sys.argv[2:]] or [1, 4]
from geodiffuser_amd.synthetic import make_edit
from geodiffuser_amd.ui_utils import save_exp
src = "/tmp/gd_ds_src"
shutil.rmtree(src, ignore_errors=True)
os.makedirs(src)
for j in range(n):
    image, depth, mask, T = make_edit(j, size=512, kind="rotate")
    save_exp(src, image, depth, depth / depth.max(), mask, T.numpy(), h=512, w=512, exp_transform_type="Mix")
print(f"{n} experiment folders written", flush=True)
for P in ps:
    dst = f"/tmp/gd_ds_p{P}"
    shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "geodiffuser_amd.large_scale_editor", "--root", dst, "--gpus", "1", "--edits-in-flight", str(P)],
                       cwd=ROOT, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    done = sum(1 for d, _, f in os.walk(dst) if "result_ls.png" in f)
    print(f"P = {P}: exit {r.returncode}, {done} of {n} results, {dt:.1f} s wall for the whole job (model build and first-edit warm-up of every rank included) "
          f"= {60 * done / dt:.1f} experiments/min", flush=True)
    if r.returncode:
        print("\n".join(l for l in r.stderr.splitlines() if "rank0" in l or "Error" in l)[-3000:])
