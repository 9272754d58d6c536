"""MIOpen find-db persistence for the UNet's convolutions (plumbing around the hot path).

With ``torch.backends.cudnn.benchmark = True`` MIOpen times its convolution solvers the first time it meets a problem shape (a UNet
pass at batch 1 / 2 / 3 / 4, forward and backward: ~150 shapes) — minutes on a fresh process, 100x the work of the edit itself, and every
rank of an 8-edit job (BASELINE configs[2]) would pay it.  MIOpen keeps the results in a per-user find-db; this module points that db
(and the compiled-kernel cache) at a directory INSIDE the package, ``geodiffuser_amd/miopen_db/``, which is committed with the entries
of the benchmark's shapes (recorded on an MI355X by tools/record_miopen_db.sh).  A process that finds its shapes there skips the
solver search.  Must run before the first convolution of the process.
"""
from __future__ import annotations

import atexit
import glob
import os
import shutil
import tempfile
import warnings

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
_STATE = {"dir": None, "seed_names": (), "warned": False, "scratch": False}


def _db_names(d):
    return sorted(os.path.basename(f) for f in glob.glob(os.path.join(d, "*.ufdb.txt")))


def configure(path: str = None, per_rank_copy: bool = True) -> str:
    """Returns the directory MIOpen will use.  The committed db is a READ-ONLY SEED: every process works on a scratch copy under the
    system temp dir (removed at exit), so a run never dirties the package directory, a read-only install works, and two processes on a
    node never write one sqlite / text db concurrently.  ``GD_MIOPEN_DB`` names another seed; ``GD_MIOPEN_DB_RECORD=1`` works IN that
    directory instead of a copy (tools/record_miopen_db.sh: how the committed db is produced); an explicit ``MIOPEN_USER_DB_PATH`` in
    the environment wins over everything; ``GD_MIOPEN_CACHE=0`` leaves MIOpen's configuration alone."""
    if os.environ.get("GD_MIOPEN_CACHE", "1") != "1":
        return os.environ.get("MIOPEN_USER_DB_PATH", "")
    if _STATE["dir"]:
        return _STATE["dir"]
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = path or os.environ.get("GD_MIOPEN_DB") or _DIR
    record = os.environ.get("GD_MIOPEN_DB_RECORD", "0") == "1"
    if record or not per_rank_copy:
        os.makedirs(src, exist_ok=True)
        use = src
    else:
        use = tempfile.mkdtemp(prefix=f"gd_miopen_db_rank{os.environ.get('RANK', '0')}_")
        if os.path.isdir(src):
            shutil.copytree(src, use, dirs_exist_ok=True)
        _STATE["scratch"] = True
        atexit.register(_cleanup, use)
    _STATE["dir"] = use
    _STATE["seed"] = src
    _STATE["seed_names"] = tuple(_db_names(use))
    os.environ["MIOPEN_USER_DB_PATH"] = use
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(use, "cache"))
    # Keep MIOpen's reference ("naive") direct convolutions out of the search: they are 1000x slower than the implicit-GEMM solvers on
    # these shapes (34 ms vs 33 us), each costs the search seconds, and — decisive for persistence — a find-db record that lists a solver
    # whose compiled kernel is not in the kernel cache is thrown away and regenerated ("Kernel cache entry not found for solver:
    # ConvDirectNaiveConvFwd ... Find-db regenerating", MIOPEN_LOG_LEVEL=6), and the naive kernels are never in the cache.
    for k in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
    if not _STATE["seed_names"]:
        warnings.warn(f"geodiffuser_amd.miopen_cache: no find-db records under {src}; the first edit of this process runs MIOpen's "
                      "solver search for every convolution shape (tens of seconds)", RuntimeWarning, stacklevel=2)
    return use


def seed_dir() -> str:
    """The find-db seed in use: ``GD_MIOPEN_DB`` / the committed directory — or, when ``MIOPEN_USER_DB_PATH`` / ``GD_MIOPEN_CACHE=0`` overrode
    the mechanism, whatever MIOpen was pointed at."""
    if os.environ.get("GD_MIOPEN_CACHE", "1") != "1" or (not _STATE["dir"] and os.environ.get("MIOPEN_USER_DB_PATH")):
        return os.environ.get("MIOPEN_USER_DB_PATH", "")
    src = _STATE.get("seed") or os.environ.get("GD_MIOPEN_DB") or _DIR
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.relpath(src, here) if os.path.abspath(src).startswith(here) else src


def check_db_used() -> bool:
    """The find-db files are keyed to ONE MIOpen build string (``gfx950100.HIP.<version>_<git hash>.ufdb.txt``); a different build
    silently ignores them and searches again (48 s instead of 2.7 s for the first edit).  MIOpen names its db after its own build the
    first time it stores a record, so a file the seed did not have means the seed was not used: warn once, naming both.  Call after the
    first convolutions of the process (load_model's callers do after their first edit; also runs at exit).  True = the seed matched."""
    d = _STATE["dir"]
    if not d or not os.path.isdir(d):
        return True
    new = [n for n in _db_names(d) if n not in _STATE["seed_names"]]
    if new and _STATE["seed_names"] and not _STATE["warned"]:
        _STATE["warned"] = True
        warnings.warn("geodiffuser_amd.miopen_cache: the committed find-db (" + ", ".join(_STATE["seed_names"]) + ") has no record for "
                      "the running MIOpen build (it wrote " + ", ".join(new) + "): every process repeats the solver search. Re-record "
                      "with tools/record_miopen_db.sh on this image.", RuntimeWarning, stacklevel=2)
    return not new


def _cleanup(d):
    try:
        check_db_used()
    finally:
        shutil.rmtree(d, ignore_errors=True)
