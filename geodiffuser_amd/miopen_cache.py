"""MIOpen find-db persistence for the UNet's convolutions (plumbing around the hot path).

With ``torch.backends.cudnn.benchmark = True`` MIOpen times its convolution solvers the first time it meets a problem shape (a UNet
pass at batch 1 / 2 / 3 / 4, forward and backward: ~150 shapes) — minutes on a fresh process, 100x the work of the edit itself, and every
rank of an 8-edit job (BASELINE configs[2]) would pay it.  MIOpen keeps the results in a per-user find-db; this module points that db
(and the compiled-kernel cache) at a directory INSIDE the package, ``geodiffuser_amd/miopen_db/``, which is committed with the entries
of the benchmark's shapes (recorded on an MI355X by tools/record_miopen_db.sh).  A process that finds its shapes there skips the
solver search.  Must run before the first convolution of the process.
"""
from __future__ import annotations

import os
import shutil
import tempfile

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def configure(path: str = None, per_rank_copy: bool = True) -> str:
    """Returns the directory MIOpen will use.  ``GD_MIOPEN_DB`` overrides the location; an explicit ``MIOPEN_USER_DB_PATH`` in the
    environment wins over both.  With several ranks on a node each rank works on its own copy of the committed db (sqlite / text
    files are not safe under concurrent writers) under the system temp dir."""
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = path or os.environ.get("GD_MIOPEN_DB") or _DIR
    os.makedirs(src, exist_ok=True)
    use = src
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if per_rank_copy and world > 1:
        use = os.path.join(tempfile.gettempdir(), f"gd_miopen_db_rank{os.environ.get('RANK', '0')}_{os.getpid()}")
        shutil.copytree(src, use, dirs_exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = use
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(use, "cache"))
    # Keep MIOpen's reference ("naive") direct convolutions out of the search: they are 1000x slower than the implicit-GEMM solvers on
    # these shapes (34 ms vs 33 us), each costs the search seconds, and — decisive for persistence — a find-db record that lists a solver
    # whose compiled kernel is not in the kernel cache is thrown away and regenerated ("Kernel cache entry not found for solver:
    # ConvDirectNaiveConvFwd ... Find-db regenerating", MIOPEN_LOG_LEVEL=6), and the naive kernels are never in the cache.
    for k in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
    return use
