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
    """all-reduce(SUM) in place; RCCL ("nccl") reduces device tensors directly.  A synchronous call (async_op False) is
    enqueued by torch >= 2.8 on the CURRENT stream itself -- no trampoline stream, no event pair around the collective
    -- which is what the step wants: the collective is one more kernel of the stream it belongs to.  The gloo backend
    (CPU tests, and the several-ranks-on-one-GPU tests: RCCL refuses two ranks on one device) may lack device support,
    in which case the tensor is staged through the host -- decided once, identically on every rank."""
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


# A rank's share of the batch at or below which the step exchanges its bucket in ONE collective (see plan_exchange):
# 160 groups of 16 candidates (its own bound, from the exchange's fixed cost against the time there is to hide the first
# piece under -- not the library's small-batch kernel threshold, cv_model::tiny_g, which is 400 groups).
TINY_SHARE = 2560


def plan_exchange(model, global_batch):
    """Choose how the optimizer steps of `model` exchange their gradient bucket, from the GLOBAL batch size (the same
    number on every rank, so every rank chooses the same):
      "split"  the dense 95 % of the bucket (fc4 / fc5 / heads: final before the convolution backward pass) on the
               communication stream under the rest of the backward pass, the convolution part + loss header behind
               it -- two collectives, one stream hand-over each way; pays when there is a backward pass long enough to
               hide 6 MB of all-reduce (a rank's share of thousands of candidates);
      "one"    the whole bucket in one collective on the step's own stream after the backward pass -- no second
               stream, no events: at a share of ~1 000 candidates (train.py's batch of 10 000 over 8 GPUs) the
               convolution backward pass is ~150 us, less than the fixed cost of the hand-overs it would hide under
               (measured with one rank, where no byte moves: profiles/r05/exchange_fixed_cost.txt).
    CV_EXCHANGE=one|split overrides.  Returns the mode."""
    _rank, ws = world()
    share = -(-int(global_batch) // max(ws, 1))
    mode = os.environ.get("CV_EXCHANGE") or ("one" if share <= TINY_SHARE else "split")
    if mode not in ("one", "split"):
        raise ValueError("CV_EXCHANGE must be 'one' or 'split'")
    model._exchange_mode = mode
    return mode


def exchange_mode(model):
    """the plan of plan_exchange; a model nobody planned for exchanges in two pieces (correct at any size -- and not
    re-planned here from the batch a step is handed: a rank only sees its own share, shares differ by one candidate
    between ranks, and ranks that chose differently around the line would issue different collectives)"""
    mode = getattr(model, "_exchange_mode", None) or os.environ.get("CV_EXCHANGE") or "split"
    if mode not in ("one", "split"):
        raise ValueError("CV_EXCHANGE must be 'one' or 'split', not %r" % (mode,))
    return mode


def comm_stream(model):
    """The stream the early part of a "split" exchange is enqueued on (one per model), or None when there is nothing
    to exchange or the plan is "one".  cv_grad_async makes it wait for the event "fc4 / fc5 / head gradients are final"."""
    if not _active() or torch.device(model.device).type != "cuda" or exchange_mode(model) == "one":
        return None
    st = getattr(model, "_comm_stream", None)
    if st is None:
        st = model._comm_stream = torch.cuda.Stream(device=model.device)
    return st


def exchange_bucket(model, comm=None):
    """Gradient + loss exchange of one optimizer step, in place on the model's bucket (no staging copies, no host
    synchronisation), as planned by plan_exchange.  The loss header sums to the global-batch losses; its L2 slot is
    identical on every rank and is divided by the rank count when read (cv_loss_accumulate).  Adam follows on the
    current stream.

    "one": all-reduce(SUM) of the whole bucket on the current stream.
    "split": all-reduce(SUM) of the dense part [dense_begin, end) on `comm` (which cv_grad_async made wait for "dense
    gradients final"), so that it runs under the rest of the backward pass; then [0, dense_begin) = loss header +
    convolution gradients, also on `comm` once it has picked up the end of the backward pass; the current stream continues
    behind both.  The collectives are synchronous calls: enqueued on the stream that is current (see _all_reduce), so the
    cross-stream edges of a step are the event into `comm`, the pick-up and the join back.  CV_EXCHANGE_ASYNC=1 restores the round-4 form (async_op on torch's own
    communication stream, three waits) for A/B runs."""
    if not _active():
        return
    b = model._bucket
    if comm is None and exchange_mode(model) == "one":
        _all_reduce(b)
        return
    d = model._bucket_dense
    if os.environ.get("CV_EXCHANGE_ASYNC"):
        if comm is not None:
            with torch.cuda.stream(comm):
                w1 = _all_reduce(b[d:], async_op=True)
        else:
            w1 = _all_reduce(b[d:], async_op=True)
        w2 = _all_reduce(b[:d], async_op=True)
        for w in (w1, w2):
            if w is not None:
                w.wait()
    elif comm is not None:
        # BOTH collectives on `comm`, in this order on every rank: one communicator is only ever driven from one stream at
        # a time (two streams issuing on the same communicator would lean on RCCL's internal serialisation -- never
        # exercised here with more than one rank).  comm already waits for "dense gradients final"; it picks up the end of
        # the backward pass before the second piece.
        cur = torch.cuda.current_stream(model.device)
        with torch.cuda.stream(comm):
            _all_reduce(b[d:])
        comm.wait_stream(cur)
        with torch.cuda.stream(comm):
            _all_reduce(b[:d])
    else:
        _all_reduce(b[d:])
        _all_reduce(b[:d])
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
