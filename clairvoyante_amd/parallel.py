"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

Inference shards candidates: contiguous index ranges per rank, replicated weights, NO
collective on the data path (candidates are independent: clairvoyante_v3.py:54-138 has no
cross-candidate state).  Training is data parallel: the loss is a SUM over the batch
(v3.py:140-151), so each rank back-propagates its shard and ONE all-reduce(SUM) of the flat
1 631 496-float gradient per optimizer step restores the global-batch gradient; the
lambda*w term is added once, identically on every rank, inside the Adam kernel.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _active():
    """collectives run when there is more than one rank (or when forced, to test them at N=1)"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("CV_FORCE_DIST")))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 and not os.environ.get("CV_FORCE_DIST"):     # CV_FORCE_DIST: exercise the collective path at N=1
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    # the ranks of a node share its cores: split the host threads of the native data plane between them
    try:
        from . import _lib
        lws = max(int(os.environ.get("LOCAL_WORLD_SIZE", str(ws))), 1)
        _lib.load().cv_set_host_threads(max(1, min(_lib.usable_cores() // lws, 16)))
    except Exception:
        pass
    return rank, ws, local


def shard_range(total, rank, world_size):
    """Contiguous shard [lo, hi) of `total` items for `rank`."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(flat, losses=None):
    """In-place SUM all-reduce of a flat tensor (+ optional list of python floats)."""
    if not _active():
        return losses
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if losses is not None:
        t = torch.tensor(losses, dtype=torch.float64, device=flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        losses = t.tolist()
    return losses


def allreduce_gradients(model, losses):
    """Gradient exchange of one optimizer step.  losses = [l1,l2,l3,l4,lL2,total] of this
    rank's shard; returns the global-batch values (lL2 is identical on all ranks)."""
    if not _active():
        return losses
    from . import _lib
    n = model.numParameters
    if getattr(model, "_grad_bucket", None) is None:
        model._grad_bucket = torch.empty(n, dtype=torch.float32, device=model.device)
    b = model._grad_bucket
    st = model._stream()
    _lib.check(model._lib.cv_flat_copy(model._h, 1, ctypes.c_void_p(b.data_ptr()), 0, st))
    data = allreduce_sum_(b, losses[0:4])
    _lib.check(model._lib.cv_flat_copy(model._h, 1, ctypes.c_void_p(b.data_ptr()), 1, st))
    l2 = losses[4]
    return data + [l2, sum(data) + l2]


def broadcast_parameters(model, src=0):
    """Replicate rank `src`'s weights and optimizer slots (done once after init/restore)."""
    if not _active():
        return
    from . import _lib
    n = model.numParameters
    b = torch.empty(n, dtype=torch.float32, device=model.device)
    st = model._stream()
    for which in (0, 2, 3):
        _lib.check(model._lib.cv_flat_copy(model._h, which, ctypes.c_void_p(b.data_ptr()), 0, st))
        dist.broadcast(b, src=src)
        _lib.check(model._lib.cv_flat_copy(model._h, which, ctypes.c_void_p(b.data_ptr()), 1, st))


def allreduce_scalar(value, model=None):
    """SUM of a python float over the ranks (validation-loss bookkeeping of train.py:118-122)."""
    if not _active():
        return value
    dev = model.device if model is not None else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
