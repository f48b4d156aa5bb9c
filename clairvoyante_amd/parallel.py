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


_suspended = [False]


class exchange_suspended(object):
    """`with parallel.exchange_suspended():` -- steps inside run WITHOUT the gradient exchange (every rank applies its own
    shard's gradient: the replicas diverge).  Measurement only: bench.py times the compute part of a data-parallel step
    with it, to say how much of the exchange the backward pass hides."""

    def __enter__(self):
        _suspended[0] = True

    def __exit__(self, *exc):
        _suspended[0] = False


def _active():
    """collectives run when there is more than one rank (or when forced, to test them at N=1)"""
    if _suspended[0]:
        return False
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
        backend = os.environ.get("CV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


_host_staged = [None]      # gloo without device support: collectives of device tensors go through the host


def _all_reduce(t, async_op=False):
    """all-reduce(SUM) in place; RCCL ("nccl") reduces device tensors directly.  The gloo backend (CPU tests, and
    the two-ranks-on-one-GPU tests: RCCL refuses two ranks on one device) may lack device support, in which case
    the tensor is staged through the host -- decided once, identically on every rank."""
    if t.is_cuda and dist.get_backend() == "gloo":
        if _host_staged[0] is None:
            try:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                _host_staged[0] = False
                return None
            except RuntimeError:
                _host_staged[0] = True
        if _host_staged[0]:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def _broadcast(t, src):
    if t.is_cuda and dist.get_backend() == "gloo" and _host_staged[0] is not False:
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
        return
    dist.broadcast(t, src=src)


def comm_stream(model):
    """The stream the early part of the gradient exchange is enqueued on (one per model), or None when there
    is nothing to exchange.  cv_grad_async makes it wait for the event "fc4 / fc5 / head gradients are final"."""
    if not _active() or torch.device(model.device).type != "cuda":
        return None
    st = getattr(model, "_comm_stream", None)
    if st is None:
        st = model._comm_stream = torch.cuda.Stream(device=model.device)
    return st


def exchange_bucket(model, comm=None):
    """Gradient + loss exchange of one optimizer step, in place on the model's bucket (no staging copies, no
    host synchronisation): all-reduce(SUM) of the dense part [dense_begin, end) -- 95 % of the bytes, final before
    the convolution backward pass -- on `comm`, so that it runs under the rest of the backward pass, then
    all-reduce(SUM) of [0, dense_begin) = loss header + convolution gradients in stream order.  The current
    stream continues behind both (Adam follows).  The loss header sums to the global-batch losses; its L2 slot is
    identical on every rank and is divided by the rank count when read (cv_loss_accumulate)."""
    if not _active():
        return
    b = model._bucket
    d = model._bucket_dense
    if comm is not None:
        with torch.cuda.stream(comm):
            w1 = _all_reduce(b[d:], async_op=True)
    else:
        w1 = _all_reduce(b[d:], async_op=True)
    w2 = _all_reduce(b[:d], async_op=True)
    for w in (w1, w2):
        if w is not None:
            w.wait()
    if comm is not None:
        torch.cuda.current_stream(model.device).wait_stream(comm)


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
        _broadcast(b, src)
        _lib.check(model._lib.cv_flat_copy(model._h, which, ctypes.c_void_p(b.data_ptr()), 1, st))


def allreduce_scalar(value, model=None):
    """SUM of a python float over the ranks (validation-loss bookkeeping of train.py:118-122)."""
    if not _active():
        return value
    dev = model.device if model is not None else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
