"""One-process-per-GPU plumbing for sharded batches: spectra are independent, so the only
collectives are one broadcast of the shared lookup tables (rank 0 builds them) and one gather of the results;
the fit itself never communicates.

Backends (``init_from_env(backend=...)``, or HIPDRT_DIST_BACKEND in the environment):
  'rccl'  RCCL behind the C ABI (include/hipdrt.h: hipdrt_comm_*, csrc/comm.hip) -- ctypes only, no torch import; the default
          whenever this process can open a gfx950 device.  Rank 0's ncclUniqueId travels through a file in /tmp whose name is
          derived from MASTER_PORT and the launcher process (every rank of one launch has the same parent: torch.distributed.run's
          agent, or bench.py's own spawner), so nothing listens on MASTER_PORT, which the launcher's own store may hold.
  'nccl'  torch.distributed's RCCL backend (rounds 1-5), 'gloo' its CPU backend (tests/test_dist_gloo.py: two ranks without
          a GPU).  torch is imported only when one of these is asked for."""
import os
import time

import numpy as np

# A world of ONE rank normally skips every collective.  `force` (init_from_env(force=True), or HIPDRT_FORCE_DIST=1 in the
# environment) keeps them: a one-rank communicator is created and broadcast / gather / all-reduce really go through the
# backend -- on a one-GPU box that is the only way to run RCCL load, the device binding and the host <-> device staging below
# before a multi-GPU node exists (tests/test_gpu_mapping.py::test_world_one_nccl_group_runs_every_collective, bench.py --force-dist).
_FORCED = os.environ.get("HIPDRT_FORCE_DIST", "") not in ("", "0")
_STATE = dict(backend=None, rank=0, world=1, comm=None, inits=0)


def forced():
    return _FORCED


def backend():
    """'rccl' | 'nccl' | 'gloo' once a group exists, else None"""
    return _STATE["backend"]


def _torch_dist():
    import torch.distributed as dist
    return dist


def is_initialized():
    if _STATE["backend"] == "rccl":
        return _STATE["comm"] is not None
    if _STATE["backend"] in ("nccl", "gloo"):
        return _torch_dist().is_initialized()
    return False


def get_rank():
    return _STATE["rank"] if is_initialized() else 0


def get_world_size():
    return _STATE["world"] if is_initialized() else 1


def active(world=None):
    """do the collectives of a `world`-rank job go through a backend? (several ranks, or one with `force`)"""
    if world is None:
        world = get_world_size()
    return is_initialized() and (world > 1 or _FORCED)


def _launcher_key():
    """what every rank of ONE launch shares and no other launch does: MASTER_PORT, the parent process and its start time"""
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]          # field 22: starttime (the name may hold spaces / parentheses)
    except OSError:
        pass
    run = os.environ.get("TORCHELASTIC_RUN_ID", "")
    return f"{os.environ.get('MASTER_PORT', '29500')}_{ppid}_{start}_{run}_{_STATE['inits']}"


def _exchange_unique_id(ffi, rank, world, timeout=180.0):
    if world == 1:
        return ffi.comm_unique_id(), None
    path = os.environ.get("HIPDRT_RCCL_ID_FILE") or os.path.join("/tmp", f"hipdrt_rccl_{_launcher_key()}.id")
    if rank == 0:
        uid = ffi.comm_unique_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        # /tmp is shared: never follow a link somebody else planted under the predictable name, never reuse an existing file
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)                                       # atomic: a reader never sees a partial file
        return uid, path
    t0 = time.time()
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
            with os.fdopen(fd, "rb") as f:
                mine = os.fstat(f.fileno()).st_uid == os.getuid()   # (only this user's rank 0 can have written it)
                uid = f.read() if mine else b""
            if len(uid) == 128:
                return uid, path
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError(f"rank {rank}: no RCCL unique id from rank 0 after {timeout:.0f} s ({path})")
        time.sleep(0.02)


def init_from_env(backend=None, device=None, force=False):
    """Initialise the process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world_size, local_rank).  Single-process runs skip initialisation unless `force`.
    A backend that cannot be initialised raises here, before any fit has touched the GPU: the caller exits non-zero
    (bench.py does), nothing is retried or restarted."""
    global _FORCED
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _FORCED = _FORCED or bool(force)
    if (world > 1 or _FORCED) and not is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("HIPDRT_DIST_BACKEND") or None
        dev = local if device is None else int(device)
        if backend is None:
            from .. import _ffi
            backend = "rccl" if _ffi.device_usable(dev) else "gloo"
        if backend == "rccl":
            from .. import _ffi
            uid, id_file = _exchange_unique_id(_ffi, rank, world)
            comm = _ffi.Comm(dev, rank, world, uid)                 # collective: returns when every rank has joined
            _STATE.update(backend="rccl", rank=rank, world=world, comm=comm)
            _STATE["inits"] += 1                                    # (a later group of the same launch gets another file name)
            comm.barrier()
            if rank == 0 and id_file is not None:                   # everybody has read it
                try:
                    os.remove(id_file)
                except OSError:
                    pass
        elif backend in ("nccl", "gloo"):
            import torch
            dist = _torch_dist()
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = torch.device("cuda", device)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
            _STATE.update(backend=backend, rank=rank, world=world, comm=None)
        else:
            raise ValueError(f"unknown distributed backend {backend!r} (rccl, nccl, gloo)")
    return rank, world, local


def destroy():
    """leave the group (tests, the end of bench.py)"""
    if _STATE["backend"] == "rccl" and _STATE["comm"] is not None:
        _STATE["comm"].close()
    elif _STATE["backend"] in ("nccl", "gloo"):
        dist = _torch_dist()
        if dist.is_initialized():
            dist.destroy_process_group()
    _STATE.update(backend=None, rank=0, world=1, comm=None)


def _torch_dev():
    import torch
    dist = _torch_dist()
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_arrays(arrays, src=0):
    """Broadcast a list of float64 numpy arrays of rank-identical shapes from `src` in ONE collective
    (they are packed into one buffer: the lookup tables are 2 x 16 kB, latency bound)."""
    if not active():
        return [np.asarray(a, dtype=np.float64) for a in arrays]
    shapes = [np.shape(a) for a in arrays]
    flat = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in arrays]))
    if _STATE["backend"] == "rccl":
        _STATE["comm"].broadcast(flat, src)                         # in place: numpy -> device -> RCCL -> numpy
    else:
        import torch
        t = torch.from_numpy(flat).to(_torch_dev())
        _torch_dist().broadcast(t, src=src)
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
    world, rank = get_world_size(), get_rank()
    tail = local.shape[1:]
    width = int(np.prod(tail)) if tail else 1
    pad = max(counts)
    if _STATE["backend"] == "rccl":
        buf = np.zeros(pad * width)
        buf[:local.size] = local.ravel()
        got = _STATE["comm"].gather(buf, dst)                       # B*(n+8)*8 bytes in all (tens of MB at 10k spectra), to one rank only
        if rank != dst:
            return None
        parts = [got[r][:counts[r] * width].reshape((counts[r],) + tail) for r in range(world)]
        return np.concatenate(parts, axis=0)
    import torch
    dist = _torch_dist()
    buf = torch.zeros(pad * width, dtype=torch.float64, device=_torch_dev())
    buf[:local.size] = torch.from_numpy(local.ravel()).to(_torch_dev())
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)
    if rank != dst:
        return None
    parts = [g.cpu().numpy()[:counts[r] * width].reshape((counts[r],) + tail) for r, g in enumerate(gathered)]
    return np.concatenate(parts, axis=0)


def max_over_ranks(value):
    if not active():
        return float(value)
    if _STATE["backend"] == "rccl":
        return _STATE["comm"].allreduce_max(float(value))
    import torch
    dist = _torch_dist()
    t = torch.tensor([float(value)], dtype=torch.float64, device=_torch_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if not active():
        return
    if _STATE["backend"] == "rccl":
        _STATE["comm"].barrier()
    else:
        _torch_dist().barrier()
