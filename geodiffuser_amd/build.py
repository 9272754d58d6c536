"""Build libgeodiff_hip.so (hand-written HIP kernels, gfx950) in-tree with hipcc.

    python -m geodiffuser_amd.build            # or: from geodiffuser_amd.build import build; build()

hipcc cross-compiles without a GPU; the built .so travels with the source tree (git-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libgeodiff_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wno-unused-result"]
# per-file extras.  attn_fwd: -O3's SLP vectoriser packs the softmax's neighbouring f32 adds / multiplies into v_pk_*_f32, which cost
# more issue time beside MFMAs than the single instructions they replace (MI355X_MICROARCH.md, per-instruction constants)
EXTRA_FLAGS = {"attn_fwd.hip": ["-fno-slp-vectorize"], "attn_fwd_mp.hip": ["-fno-slp-vectorize"], "attn_bwd.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libgeodiff_hip.so")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "geodiff_hip.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, os.path.abspath(__file__)] + headers):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
