#!/bin/bash
# Timing-only variants of the pipelined attention forward (attn_fwd_mp.hip, GD_MP_DBG bits): one shared library per variant under
# tools/ub/build/ (git-ignored, travels with gpurun).  Results of these libraries are WRONG by construction; tools/bench_dbg.py times them.
#   tools/build_dbg.sh 1 3 7 15      -> tools/ub/build/libgd_dbg<N>.so
cd "$(dirname "$0")/.."
python -m geodiffuser_amd.build >/dev/null || exit 1
mkdir -p tools/ub/build
OBJS=$(ls geodiffuser_amd/csrc/build/*.o | grep -v attn_fwd_mp.o)
for d in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize -DGD_MP_DBG=$d $GD_DBG_EXTRA \
      -c geodiffuser_amd/csrc/attn_fwd_mp.hip -o tools/ub/build/attn_fwd_mp_dbg$d.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ub/build/libgd_dbg$d.so $OBJS tools/ub/build/attn_fwd_mp_dbg$d.o &&
    echo "built tools/ub/build/libgd_dbg$d.so" ) &
done
wait
