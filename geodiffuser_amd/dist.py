"""Multi-GPU: one process per GPU, independent edits sharded round-robin, no collective in the step.

The reference is single-process / single-GPU (README.md:88); an edit is a self-contained ~117-UNet-pass job on a batch of
two latents, so the path shards naturally by edit.  The only collectives are one broadcast of the model
weights from rank 0 at start-up (RCCL over xGMI on GPUs, gloo on CPU for tests) and the barrier / max-reduce of the
benchmark's timing.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence, TypeVar

import torch
import torch.distributed as dist

T = TypeVar("T")


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def procs_per_gpu() -> int:
    """GD_EDITS_IN_FLIGHT = P > 1: P ranks share every GPU (rank r drives device LOCAL_RANK // P).  One edit is ~117 UNet passes at batch
    1-3 — launch-bound kernels that leave most of the chip idle — and edits are independent, so P processes per GPU overlap on the
    device: 66 -> 96 / 112 / 123 edits/min at P = 2 / 3 / 4 on one MI355X.  The control plane then runs over gloo (RCCL does
    not take two ranks on one device); the one-off weight broadcast is staged through the host."""
    try:
        return max(1, int(os.environ.get("GD_EDITS_IN_FLIGHT", "1")))
    except ValueError:
        return 1


def local_device_index(local_rank: int) -> int:
    return local_rank // procs_per_gpu()


PINNED_CORES = None      # the host cores this rank was pinned to (None: not pinned)


def pin_rank_to_cores(local_rank: int = None, local_world: int = None, max_threads: int = 16):
    """Give every rank of a node its own slice of the host cores — BEFORE the first GPU call, so that the runtime's helper threads inherit
    the mask — and size torch's intra-op pool to it.  A rank is a launcher thread that must keep ~1,500-kernel graph launches and the
    eager glue between them ahead of its GPU, plus the pre-pass worker (editor.start_ahead) and the folder driver's IO threads; left
    unpinned, N ranks x (those + a torch pool as wide as the whole host each) migrate over each other's cores, and one edit at a time
    already loses time to host gaps on a quiet box.  Contiguous slices of the cores the process may use (a two-socket
    node numbers a socket's cores contiguously and hangs GPUs 0..N/2-1 off socket 0); no-op for a single rank per node, where
    sched_setaffinity does not exist, or with GD_PIN_CORES=0.  -> the rank's cores (list) or None."""
    global PINNED_CORES
    if os.environ.get("GD_PIN_CORES", "1") != "1" or not hasattr(os, "sched_setaffinity"):
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1")) if local_world is None else local_world      # (set by torch.distributed.run)
    if local_world <= 1:
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(max_threads, per - 2 if per > 3 else per)))      # two cores stay with the launcher + the pre-pass worker
    PINNED_CORES = mine
    return mine


def init(backend: str = None, force: bool = None) -> tuple:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process, unless ``force`` / GD_DIST_FORCE=1 and a
    rendezvous is configured: a one-rank group, so that RCCL's initialisation and the broadcast path can run on a one-GPU box)."""
    rank, world, local = env_rank_world()
    if procs_per_gpu() > 1 and backend is None:
        backend = "gloo"
    if force is None:
        force = os.environ.get("GD_DIST_FORCE", "0") == "1"
    if (world > 1 or (force and "MASTER_PORT" in os.environ)) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # every rank must name the SAME port, so a rank cannot pick (or retry) one on its own: the launcher does
            # (torchrun --master-port, `bench.py --gpus N` takes a free one).  A silent default collides with whatever else runs on the node.
            raise RuntimeError("geodiffuser_amd.dist.init: WORLD_SIZE > 1 but MASTER_PORT is not set — start the ranks through "
                               "torchrun / `bench.py --gpus N` (they pass a port to every rank)")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        pin_rank_to_cores()              # (before the first GPU call of the rank: torch.cuda.is_available() below is one)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def visible_gpu_count() -> int:
    """GPUs this process tree may use, WITHOUT touching the HIP runtime (a launcher that only forks rank processes should not
    initialise the GPU; on ROCm ``torch.cuda.device_count()`` may fall through to ``hipGetDeviceCount``).  The visibility masks win
    (``HIP_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES`` index into what ``ROCR_VISIBLE_DEVICES`` leaves); otherwise the kernel driver's
    topology: every KFD node with ``simd_count > 0`` is a GPU.  -1 when neither source exists (no amdgpu driver: the caller decides)."""
    n_mask = None
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            k = len([x for x in v.split(",") if x.strip() not in ("", "-1")])
            n_mask = k if n_mask is None else min(n_mask, k)
    root = "/sys/class/kfd/kfd/topology/nodes"
    n_kfd = None
    if os.path.isdir(root):
        n_kfd = 0
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n_kfd += 1
            except (OSError, ValueError):
                continue
    if n_mask is not None:
        return n_mask if n_kfd is None else min(n_mask, n_kfd)
    return -1 if n_kfd is None else n_kfd


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n_gpus: int, per_gpu: int, target: Sequence[str], argv: Sequence[str], run=None) -> int:
    """Start ``n_gpus * per_gpu`` ranks of ``target`` (["-m", module] or [script path]) on this node through ``torch.distributed.run`` as a
    CHILD process — call it before anything touches the GPU — with a free rendezvous port on 127.0.0.1, and return the child's exit code.
    ``per_gpu`` > 1 exports GD_EDITS_IN_FLIGHT (see procs_per_gpu).  Refuses when fewer than ``n_gpus`` devices are visible (counted
    without HIP).  ``run``: injection point of the CPU tests."""
    import subprocess
    import sys
    have = visible_gpu_count()
    if 0 <= have < n_gpus:
        print(f"launch_ranks: {n_gpus} GPU(s) asked for but only {have} visible on this node", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if per_gpu > 1:
        env["GD_EDITS_IN_FLIGHT"] = str(per_gpu)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus * per_gpu}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), *target, *argv]
    return (run or subprocess.call)(cmd, env=env)


def shard(items: Sequence[T], rank: int, world: int) -> List[T]:
    """edit j -> rank j mod world."""
    return [x for j, x in enumerate(items) if j % world == rank]


@torch.no_grad()
def broadcast_model(modules: Iterable[torch.nn.Module], src: int = 0, bucket_bytes: int = 256 << 20) -> int:
    """One-off weight broadcast from ``src`` in large flat buckets (xGMI is point-to-point: few, large transfers).
    Returns the number of bytes sent."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    if dist.get_world_size() == 1 and os.environ.get("GD_DIST_FORCE", "0") != "1":        # (forced: a one-rank group really broadcasts, to itself)
        return 0
    total = 0
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size, total
        if not bucket:
            return
        flat = torch.cat([p.reshape(-1) for p in bucket])
        if flat.is_cuda and dist.get_backend() == "gloo":       # several ranks per GPU: control plane on gloo, payload staged through the host
            host = flat.cpu()
            dist.broadcast(host, src=src)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src)
        off = 0
        for p in bucket:
            n = p.numel()
            p.copy_(flat[off:off + n].reshape(p.shape))
            off += n
        total += flat.numel() * flat.element_size()
        bucket, size = [], 0

    by_dtype = {}
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            by_dtype.setdefault(t.dtype, []).append(t.data)
    for dt, tensors in by_dtype.items():
        for t in tensors:
            bucket.append(t)
            size += t.numel() * t.element_size()
            if size >= bucket_bytes:
                flush()
        flush()
    return total


def _ctl_device(device):
    """Where the few-byte control tensors live: the rank's GPU under RCCL, the host under gloo."""
    if dist.get_backend() != "nccl":
        return "cpu"
    return device or "cuda"


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=_ctl_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=_ctl_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(x: float, device=None) -> List[float]:
    """[x of rank 0, x of rank 1, ...] on every rank (reporting only: the per-rank seconds of the benchmark line)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [x]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=_ctl_device(device))
    t[dist.get_rank()] = x
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
