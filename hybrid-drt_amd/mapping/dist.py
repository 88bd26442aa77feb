"""One-process-per-GPU plumbing for sharded batches: spectra are independent, so the only
collectives are one broadcast of the shared lookup tables (rank 0 builds them) and one gather of the results.
``torch.distributed`` is used for exactly that (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests); the fit itself never communicates."""
import os

import numpy as np
import torch
import torch.distributed as dist


# A world of ONE rank normally skips every collective.  `force` (init_from_env(force=True), or HIPDRT_FORCE_DIST=1 in the
# environment) keeps them: a one-rank process group is created and broadcast / gather / all-reduce really go through the
# backend -- on a one-GPU box that is the only way to run RCCL load, the device binding and the host <-> device staging below
# before a multi-GPU node exists (tests/test_gpu_mapping.py::test_world_one_nccl_group_runs_every_collective, bench.py --force-dist).
_FORCED = os.environ.get("HIPDRT_FORCE_DIST", "") not in ("", "0")


def forced():
    return _FORCED


def active(world=None):
    """do the collectives of a `world`-rank job go through torch.distributed? (several ranks, or one with `force`)"""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return dist.is_initialized() and (world > 1 or _FORCED)


def init_from_env(backend=None, device=None, force=False):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets
    them).  Returns (rank, world_size, local_rank).  Single-process runs skip initialisation unless `force`.
    A backend that cannot be initialised raises here, before any fit has touched the GPU: the caller exits non-zero
    (bench.py does), nothing is retried or restarted."""
    global _FORCED
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _FORCED = _FORCED or bool(force)
    if (world > 1 or _FORCED) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = torch.device("cuda", device)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def _dev():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_arrays(arrays, src=0):
    """Broadcast a list of float64 numpy arrays of rank-identical shapes from `src` in ONE collective
    (they are packed into one buffer: the lookup tables are 2 x 16 kB, latency bound)."""
    if not active():
        return [np.asarray(a, dtype=np.float64) for a in arrays]
    shapes = [np.shape(a) for a in arrays]
    flat = np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in arrays])
    t = torch.from_numpy(flat).to(_dev())
    dist.broadcast(t, src=src)
    flat = t.cpu().numpy()
    out, pos = [], 0
    for shp in shapes:
        size = int(np.prod(shp))
        out.append(flat[pos:pos + size].reshape(shp).copy())
        pos += size
    return out


def gather_rows(local, counts, dst=0):
    """Gather per-rank result blocks (arrays whose first axis is the local spectrum count) to `dst` with ONE collective
    (a gather: only `dst` receives; blocks padded to the largest count); counts[r] = rows owned by rank r.  Returns the
    concatenated array on dst, None elsewhere."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if not active():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    tail = local.shape[1:]
    width = int(np.prod(tail)) if tail else 1
    pad = max(counts)
    buf = torch.zeros(pad * width, dtype=torch.float64, device=_dev())
    buf[:local.size] = torch.from_numpy(local.ravel()).to(_dev())
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)     # B*(n+8)*8 bytes in all (tens of MB at 10k spectra), to one rank only
    if rank != dst:
        return None
    parts = [g.cpu().numpy()[:counts[r] * width].reshape((counts[r],) + tail) for r, g in enumerate(gathered)]
    return np.concatenate(parts, axis=0)


def max_over_ranks(value):
    if not active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if active():
        dist.barrier()
