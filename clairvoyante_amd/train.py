"""Training driver: same command line, epoch / batch schedule, learning-rate schedule,
checkpoint naming and log lines as /root/reference/clairvoyante/train.py (Run :13-34,
TrainAll :37-218, main :221-267).

    python -m clairvoyante_amd.train --bin_fn TENSORS.bin --ochk_prefix OUT/model

One process per GPU; under torchrun every rank walks the SAME schedule and trains on its
contiguous slice of each global batch, the gradient is all-reduced (RCCL) inside
`m.trainNoRT` (clairvoyante_amd/parallel.py).

Schedule facts kept from the reference:
  * trainingTotal = int(total*0.9); validationStart = trainingTotal + 1;
    numValItems = total - validationStart (train.py:67-69);
  * a batch [p, p+n) is TRAINED iff p + n < validationStart -- the test uses the pointer
    after it was advanced (train.py:88-91) -- so the batch that ends exactly at
    validationStart is evaluated, never trained; batches are 10 000 (clipped at
    validationStart) in the training part and aligned to multiples of 1 000 in the
    validation part (train.py:95-102);
  * the last batch of the data set is evaluated synchronously with getLoss (train.py:122);
  * losses are batch SUMS divided by trainingTotal / numValItems (train.py:123);
  * after >= 6 epochs a strictly alternating (zig-zag) or flat validation loss multiplies
    learning rate and lambda by their decay; the third such event stops (train.py:131-154).
"""
import argparse
import logging
import os
import pickle
import sys
import time
from threading import Thread

import numpy as np

if __package__ in (None, ""):      # run as `python <dir>/train.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param

logging.basicConfig(format='%(message)s', level=logging.INFO)


def Run(args):
    """train.py:13-34"""
    logging.info("Initializing model ...")
    from . import utils_v2 as utils
    if args.v2:
        sys.exit("Clairvoyante v2 topologies are not part of this build (v3 / v3 slim only)")
    if args.slim:
        from . import clairvoyante_v3_slim as cv
    else:
        from . import clairvoyante_v3 as cv
    utils.SetupEnv()
    from . import parallel
    parallel.init_from_env()
    m = cv.Clairvoyante()
    m.init()
    if args.chkpnt_fn is not None:
        m.restoreParameters(os.path.abspath(args.chkpnt_fn))
    parallel.broadcast_parameters(m)
    TrainAll(args, m, utils)


def _next_batch_size(ptr, validationStart, vbatch=None):
    """train.py:95-102; vbatch = size of a validation batch (the reference: predictBatchSize)"""
    if vbatch is None:
        vbatch = param.predictBatchSize
    if ptr < validationStart:
        left = validationStart - ptr
        return left if left < param.trainBatchSize else param.trainBatchSize
    if ptr % vbatch != 0:
        return vbatch - (ptr % vbatch)
    return vbatch


def _zigzag(v):
    """train.py:134-149 on the last six validation losses: strictly alternating signs of the
    successive differences (either phase), or a flat first step, request a decay."""
    d = [v[i][0] - v[i + 1][0] for i in range(-6, -1)]
    if d[0] > 0:
        return d[1] < 0 and d[2] > 0 and d[3] < 0 and d[4] > 0
    if d[0] < 0:
        return d[1] > 0 and d[2] < 0 and d[3] > 0 and d[4] < 0
    return True


def _shard(a, rank, ws):
    """this rank's contiguous slice of a global batch (data parallel)"""
    if ws == 1:
        return a
    from . import parallel
    lo, hi = parallel.shard_range(len(a), rank, ws)
    return a[lo:hi]


class _BatchStream(object):
    """The batch sequence of one epoch (train.py:83-107): 10 000-candidate batches up to
    validationStart (the last one clipped to end exactly there), then batches aligned to multiples of
    1 000; `last` marks the batch that reaches the end of the data set.

    `prefetch(size_fn, device)` starts a producer thread that walks the same sequence ahead of the consumer
    (the sizes are a function of the pointer alone) and, when `device` is given, stages every batch in HBM
    through a side stream -- decompression, host->device copy and the training step then overlap.  `fetch`
    hands out the prefetched batch if it is the one asked for, and falls back to fetching in place otherwise."""

    DEPTH = 3

    def __init__(self, utils, XC, YC, total, validationStart, rank=0, ws=1):
        self.utils, self.XC, self.YC, self.total, self.vstart = utils, XC, YC, total, validationStart
        self.rank, self.ws = rank, ws              # data parallel: this rank only decompresses its slice of a batch
        self.vbatch = None                         # validation batch size (None: the reference's predictBatchSize)
        self.ptr = 0
        self._q = None
        self._stop = None

    def rewind(self):
        self._cancel()
        self.ptr = 0

    def _fetch_at(self, ptr, size):
        """-> (X, Y, start, count of the WHOLE batch, last); with ws > 1 X / Y hold this rank's slice only"""
        if self.ws > 1:
            from . import parallel
            n = max(min(size, self.total - ptr), 0) if ptr + size >= self.total else size
            last = ptr + size >= self.total
            lo, hi = parallel.shard_range(n, self.rank, self.ws)
            X, xn, _ = self.utils.DecompressArray(self.XC, ptr + lo, hi - lo, self.total)
            Y, yn, _ = self.utils.DecompressArray(self.YC, ptr + lo, hi - lo, self.total)
            if xn != yn:
                sys.exit("Inconsistency between decompressed arrays: %d/%d" % (xn, yn))
            return X, Y, ptr, n, last
        X, xn, xe = self.utils.DecompressArray(self.XC, ptr, size, self.total)
        Y, yn, ye = self.utils.DecompressArray(self.YC, ptr, size, self.total)
        if xn != yn or xe != ye:
            sys.exit("Inconsistency between decompressed arrays: %d/%d" % (xn, yn))
        return X, Y, ptr, xn, xe != 0

    def fetch(self, size):
        if self._q is not None:
            item = self._q.get()
            if isinstance(item, BaseException):
                self._cancel()
                raise item
            if item is not None and item[0] == (self.ptr, size):
                _key, out, ready = item
                if ready is not None:
                    ready()                        # the consumer's stream waits for the staged copy
                self.ptr += out[3]
                if out[4]:
                    self._cancel()
                return out
            self._cancel()                         # a different request than predicted: fetch in place
        out = self._fetch_at(self.ptr, size)
        self.ptr += out[3]
        return out

    def next_size(self):
        return _next_batch_size(self.ptr, self.vstart, self.vbatch)

    # ---- producer side
    def prefetch(self, first_size, size_fn, device=None):
        from queue import Queue
        from threading import Event
        self._cancel()
        q = self._q = Queue(maxsize=self.DEPTH)
        stop = self._stop = Event()
        start_ptr = self.ptr

        def stage(X, Y):
            import torch
            with torch.cuda.device(device):
                side = torch.cuda.Stream(device=device)
                with torch.cuda.stream(side):
                    xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(device, non_blocking=True)
                    yd = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
            ev.synchronize()                       # producer thread: the host arrays may go once the copy is done

            def ready():
                # consumer side: its stream waits for the copy, and the blocks -- allocated on the staging stream --
                # are marked as in use by the consumer stream, so that the caching allocator does not hand them to a
                # later stage() while a queued (trainDeferred: never host-synchronised) step still reads them
                cs = torch.cuda.current_stream(device)
                cs.wait_event(ev)
                xd.record_stream(cs); yd.record_stream(cs)
            return xd, yd, ready

        def put(queue, item):
            while not stop.is_set():
                try:
                    queue.put(item, timeout=0.05)
                    return True
                except Exception:
                    continue
            return False

        def produce(out_q):
            """decompression: walks the batch sequence; items ((ptr, size), (X, Y, start, n, last), None)"""
            try:
                ptr, size = start_ptr, first_size
                while not stop.is_set():
                    X, Y, st, n, last = self._fetch_at(ptr, size)
                    if not put(out_q, ((ptr, size), (X, Y, st, n, last), None)) or last:
                        break
                    ptr += n
                    size = size_fn(ptr)
            except BaseException as e:             # surfaced by fetch()
                put(out_q, e)

        def stage_all(in_q):
            """host -> device copies, one batch behind the decompression (its own thread: the two overlap)"""
            try:
                while not stop.is_set():
                    try:
                        item = in_q.get(timeout=0.05)
                    except Exception:
                        continue
                    if isinstance(item, BaseException):
                        put(q, item)
                        break
                    key, (X, Y, st, n, last), _ = item
                    ready = None
                    if n > 0:
                        X, Y, ready = stage(X, Y)
                    if not put(q, (key, (X, Y, st, n, last), ready)) or last:
                        break
            except BaseException as e:
                put(q, e)

        if device is None:
            self._threads = [Thread(target=produce, args=(q,), daemon=True)]
        else:
            mid = Queue(maxsize=2)
            self._threads = [Thread(target=produce, args=(mid,), daemon=True),
                             Thread(target=stage_all, args=(mid,), daemon=True)]
        for t in self._threads:
            t.start()

    def _cancel(self):
        if self._q is not None:
            self._stop.set()
            try:
                while True:
                    self._q.get_nowait()
            except Exception:
                pass
            self._q = None


class _Job(Thread):
    """Worker thread of the in-flight batch (train.py:87-93,109); unlike the reference an exception
    raised inside it is re-raised by the caller."""

    def __init__(self, fn, X, Y):
        Thread.__init__(self)
        self.fn, self.X, self.Y, self.err = fn, X, Y, None

    def run(self):
        try:
            self.fn(self.X, self.Y)
        except BaseException as e:
            self.err = e

    def finish(self):
        self.join()
        if self.err is not None:
            raise self.err


def load_dataset(args, utils):
    """(total, XC, YC, posC) from --bin_fn or from --tensor_fn/--var_fn/--bed_fn (train.py:39-49)"""
    if args.bin_fn is not None:
        if hasattr(utils, "LoadBin"):
            try:
                return utils.LoadBin(args.bin_fn, lazy=True)     # memory-mapped: one page-cache copy for all ranks
            except TypeError:                                    # a foreign utils module with the plain signature
                return utils.LoadBin(args.bin_fn)
        return _load_bin(args.bin_fn)
    return utils.GetTrainingArray(args.tensor_fn, args.var_fn, args.bed_fn)


def run_epoch(stream, m, rank, ws, writer, epoch, validationStart):
    """One pass over the data set (train.py:83-123): every batch that ends strictly before
    validationStart is trained, the others are evaluated; the next batch is decompressed while the
    GPU works; the batch that reaches the end of the data set is evaluated synchronously.
    Returns (sum of training losses, sum of validation losses)."""
    from . import parallel

    def mine(a):                                    # the stream hands out this rank's slice already
        return a

    def reduced(v):
        return parallel.allreduce_scalar(float(v), m) if ws > 1 else float(v)

    train_sum = 0
    val_sum = 0
    stream.rewind()
    if ws > 1 and hasattr(m, "_enqueue_step"):
        parallel.plan_exchange(m, param.trainBatchSize)      # one collective per step when a rank's share is tiny
    # real models take batches that are already in HBM; mock / foreign model objects get the numpy arrays
    device = getattr(m, "device", None) if getattr(m, "accepts_device_batches", False) else None
    # the validation loss is a sum over candidates: a real model takes it in passes of 16 000 instead of the
    # reference's 1 000 (a pass over 1 000 candidates is launch-bound on the GPU) -- same sum
    stream.vbatch = param.predictBatchSize * 16 if device is not None else None
    vbatch = stream.vbatch
    stream.prefetch(param.trainBatchSize, lambda p: _next_batch_size(p, validationStart, vbatch), device)
    # a real model without a summary writer enqueues its steps and keeps the losses on the device (trainDeferred):
    # the epoch's sum is read once at the end instead of one host round trip per step (train.py:113-114 only sums)
    deferred = device is not None and writer is None and hasattr(m, "trainDeferred")
    if deferred:
        m.readLosses(reset=True)
    X, Y, start, count, last = stream.fetch(param.trainBatchSize)
    while True:
        training = start + count < validationStart
        step = (m.trainDeferred if deferred else m.trainNoRT) if training else m.getLossNoRT
        job = _Job(step, mine(X), mine(Y))
        job.start()
        nxt = stream.fetch(stream.next_size())
        job.finish()
        if training:
            if not deferred:
                train_sum += m.trainLossRTVal
                if writer is not None:
                    writer.add_summary(m.trainSummaryRTVal, epoch)
        else:
            val_sum += reduced(m.getLossLossRTVal)
        X, Y, start, count, last = nxt
        if last:
            break
    val_sum += reduced(m.getLoss(mine(X), mine(Y)))
    if deferred:
        train_sum = m.readLosses(reset=True)[0][5]
    return train_sum, val_sum


def TrainAll(args, m, utils):
    """train.py:37-218"""
    from . import parallel
    rank, ws = parallel.world()
    logging.info("Loading the training dataset ...")
    total, XC, YC, _posC = load_dataset(args, utils)
    logging.info("The size of training dataset: {}".format(total))

    writer = m.summaryFileWriter(args.olog_dir) if (args.olog_dir is not None and rank == 0) else None

    logging.info("Start training ...")
    logging.info("Learning rate: %.2e" % m.setLearningRate(args.learning_rate))
    logging.info("L2 regularization lambda: %.2e" % m.setL2RegularizationLambda(args.lambd))

    t_begin = time.time()
    trainingTotal = int(total * param.trainingDatasetPercentage)
    validationStart = trainingTotal + 1
    numValItems = total - validationStart
    switches_left = param.maxLearningRateSwitch
    history = []                     # (validation loss sum, epoch)
    since_switch = 0
    epoch = 1 if args.chkpnt_fn is None else int(args.chkpnt_fn[-param.parameterOutputPlaceHolder:]) + 1
    stream = _BatchStream(utils, XC, YC, total, validationStart, rank, ws)

    while epoch < param.maxEpoch:
        t_epoch = time.time()
        train_sum, val_sum = run_epoch(stream, m, rank, ws, writer, epoch, validationStart)
        logging.info(" ".join([str(epoch), "Training loss:", str(train_sum / trainingTotal), "Validation loss: ",
                               str(val_sum / numValItems)]))
        logging.info("Epoch time elapsed: %.2f s" % (time.time() - t_epoch))
        history.append((val_sum, epoch))
        if args.ochk_prefix is not None and rank == 0:
            name = "%s-%0*d" % (args.ochk_prefix, param.parameterOutputPlaceHolder, epoch)
            m.saveParameters(os.path.abspath(name))
        since_switch += 1
        if since_switch >= 6 and _zigzag(history):
            switches_left -= 1
            if switches_left == 0:
                break
            logging.info("New learning rate: %.2e" % m.setLearningRate())
            logging.info("New L2 regularization lambda: %.2e" % m.setL2RegularizationLambda())
            since_switch = 0
        epoch += 1

    logging.info("Training time elapsed: %.2f s" % (time.time() - t_begin))
    history.sort()
    logging.info("Best validation loss at batch: %d" % history[0][1])

    logging.info("Testing on the training and validation dataset ...")
    PredictAndReport(m, utils, total, XC, YC)


def PredictAndReport(m, utils, total, XC, YC):
    """train.py:163-218 / evaluate.py:51-107: predict the whole set in batches of 1 000 (the first
    batch's end flag is not looked at), then the accuracy / confusion-matrix report"""
    t_pred = time.time()
    step = param.predictBatchSize
    if getattr(m, "accepts_device_batches", False):
        step *= 16          # same concatenated result; a call on 1 000 candidates is latency-bound on the GPU
    outs = [[], [], [], []]
    ptr = 0
    while True:
        Xb, _, endFlag = utils.DecompressArray(XC, ptr, step, total)
        for acc, o in zip(outs, m.predict(Xb)):
            acc.append(o)
        ptr += step
        if ptr >= total or (endFlag != 0 and ptr > step):
            break
    bases, zs, ts, ls = [np.concatenate(o) for o in outs]
    logging.info("Prediciton time elapsed: %.2f s" % (time.time() - t_pred))
    YArray, _, _ = utils.DecompressArray(YC, 0, total, total)
    EvaluateReport(bases, zs, ts, ls, YArray)


def EvaluateReport(bases, zs, ts, ls, YArray):
    """train.py:190-218: top-1/top-2 base accuracy and the three confusion matrices"""
    logging.info("Version 2 model, evaluation on base change:")
    n = len(bases)
    truth = np.argmax(YArray[:n, 0:4], axis=1)
    order = np.argsort(bases, axis=1, kind="stable")[:, ::-1]
    top1 = int(np.sum(order[:, 0] == truth))
    top2 = top1 + int(np.sum((order[:, 0] != truth) & (order[:, 1] == truth)))
    logging.info("all/top1/top2/top1p/top2p: %d/%d/%d/%.2f/%.2f" %
                 (n, top1, top2, float(top1) / n * 100, float(top2) / n * 100))
    for title, pred, lo, hi in (("Zygosity", zs, 4, 6), ("variant type", ts, 6, 10), ("indel length", ls, 10, 16)):
        logging.info("Version 2 model, evaluation on %s:" % title)
        k = hi - lo
        ed = np.zeros((k, k), dtype=np.int64)
        np.add.at(ed, (np.argmax(YArray[:n, lo:hi], axis=1), np.argmax(pred, axis=1)), 1)
        for r in range(k):
            logging.info("\t".join([str(ed[r][j]) for j in range(k)]))


def _load_bin(fn):
    with open(fn, "rb") as fh:
        return pickle.load(fh), pickle.load(fh), pickle.load(fh), pickle.load(fh)


_CLI = (   # flag, type, default, help  -- the reference's options and defaults (train.py:221-262)
    ("--bin_fn", str, None, "Binary tensor input generated by tensor2Bin.py, tensor_fn, var_fn and bed_fn will be ignored"),
    ("--tensor_fn", str, "vartensors", "Tensor input"),
    ("--var_fn", str, "truthvars", "Truth variants list input"),
    ("--bed_fn", str, None, "High confident genome regions input in the BED format"),
    ("--chkpnt_fn", str, None, "Input a checkpoint for testing or continue training"),
    ("--learning_rate", float, param.initialLearningRate, "Set the initial learning rate, default: %(default)s"),
    ("--lambd", float, param.l2RegularizationLambda, "Set the l2 regularization lambda, default: %(default)s"),
    ("--ochk_prefix", str, None, "Prefix for checkpoint outputs at each learning rate change, optional"),
    ("--olog_dir", str, None, "Directory for tensorboard log outputs, optional"),
)
_SWITCHES = (("--v3", True, "Use Clairvoyante version 3"), ("--v2", False, "Use Clairvoyante version 2"),
             ("--slim", False, "Train using the slim version of Clairvoyante, optional"))


def build_parser(description, cli=_CLI, switches=_SWITCHES):
    parser = argparse.ArgumentParser(description=description)
    for spec in cli:
        flag, typ, default, text = spec[:4]
        parser.add_argument(flag, type=typ, default=default, help=text, **(spec[4] if len(spec) > 4 else {}))
    for flag, default, text in switches:
        parser.add_argument(flag, type=param.str2bool, nargs='?', const=True, default=default, help=text)
    return parser


def pick_model(args):
    """the reference's --v2 / --v3 / --slim switch (train.py:16-27); v2 topologies are not built"""
    if args.v2:
        sys.exit("Clairvoyante v2 topologies are not part of this build (v3 / v3 slim only)")
    if args.slim:
        from . import clairvoyante_v3_slim as cv
    else:
        from . import clairvoyante_v3 as cv
    return cv


def main():
    parser = build_parser("Train Clairvoyante")
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    Run(args)


if __name__ == "__main__":
    main()
