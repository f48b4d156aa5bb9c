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


def _next_batch_size(ptr, validationStart):
    """train.py:95-102"""
    if ptr < validationStart:
        left = validationStart - ptr
        return left if left < param.trainBatchSize else param.trainBatchSize
    if ptr % param.predictBatchSize != 0:
        return param.predictBatchSize - (ptr % param.predictBatchSize)
    return param.predictBatchSize


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


def TrainAll(args, m, utils):
    """train.py:37-218"""
    from . import parallel
    rank, ws = parallel.world()
    logging.info("Loading the training dataset ...")
    if args.bin_fn is not None:
        total, XArrayCompressed, YArrayCompressed, posArrayCompressed = utils.LoadBin(args.bin_fn) \
            if hasattr(utils, "LoadBin") else _load_bin(args.bin_fn)
    else:
        total, XArrayCompressed, YArrayCompressed, posArrayCompressed = \
            utils.GetTrainingArray(args.tensor_fn, args.var_fn, args.bed_fn)
    logging.info("The size of training dataset: {}".format(total))

    summaryWriter = None
    if args.olog_dir is not None and rank == 0:
        summaryWriter = m.summaryFileWriter(args.olog_dir)

    logging.info("Start training ...")
    logging.info("Learning rate: %.2e" % m.setLearningRate(args.learning_rate))
    logging.info("L2 regularization lambda: %.2e" % m.setL2RegularizationLambda(args.lambd))

    validationLosses = []
    trainingStart = time.time()
    trainingTotal = int(total * param.trainingDatasetPercentage)
    validationStart = trainingTotal + 1
    numValItems = total - validationStart
    maxLearningRateSwitch = param.maxLearningRateSwitch

    def fetch(ptr, size):
        X, xn, xe = utils.DecompressArray(XArrayCompressed, ptr, size, total)
        Y, yn, ye = utils.DecompressArray(YArrayCompressed, ptr, size, total)
        if xn != yn or xe != ye:
            sys.exit("Inconsistency between decompressed arrays: %d/%d" % (xn, yn))
        return X, Y, xn, xe

    class _Job(Thread):
        """worker thread for the in-flight batch; re-raises in the caller (the reference
        loses exceptions raised inside its Thread targets)"""

        def __init__(self, fn, X, Y):
            Thread.__init__(self)
            self.fn, self.X, self.Y, self.err = fn, X, Y, None

        def run(self):
            try:
                self.fn(_shard(self.X, rank, ws), _shard(self.Y, rank, ws))
            except BaseException as e:
                self.err = e

    def val_loss(X, Y):
        v = float(m.getLoss(_shard(X, rank, ws), _shard(Y, rank, ws)))
        return parallel.allreduce_scalar(v, m) if ws > 1 else v

    c = 0
    i = 1 if args.chkpnt_fn is None else int(args.chkpnt_fn[-param.parameterOutputPlaceHolder:]) + 1
    epochStart = time.time()
    trainLossSum = 0
    validationLossSum = 0
    datasetPtr = 0
    XBatch, YBatch, XNum, _ = fetch(datasetPtr, param.trainBatchSize)
    datasetPtr += XNum
    while i < param.maxEpoch:
        training = datasetPtr < validationStart
        job = _Job(m.trainNoRT if training else m.getLossNoRT, XBatch, YBatch)
        job.start()
        XBatch2, YBatch2, XNum2, XEndFlag2 = fetch(datasetPtr, _next_batch_size(datasetPtr, validationStart))
        job.join()
        if job.err is not None:
            raise job.err
        XBatch = XBatch2; YBatch = YBatch2
        if training:
            trainLossSum += m.trainLossRTVal
            if summaryWriter is not None:
                summaryWriter.add_summary(m.trainSummaryRTVal, i)
        else:
            v = float(m.getLossLossRTVal)
            validationLossSum += parallel.allreduce_scalar(v, m) if ws > 1 else v
        datasetPtr += XNum2

        if XEndFlag2 != 0:
            validationLossSum += val_loss(XBatch, YBatch)
            logging.info(" ".join([str(i), "Training loss:", str(trainLossSum / trainingTotal), "Validation loss: ",
                                   str(validationLossSum / numValItems)]))
            logging.info("Epoch time elapsed: %.2f s" % (time.time() - epochStart))
            validationLosses.append((validationLossSum, i))
            if args.ochk_prefix is not None and rank == 0:
                parameterOutputPath = "%s-%%0%dd" % (args.ochk_prefix, param.parameterOutputPlaceHolder)
                m.saveParameters(os.path.abspath(parameterOutputPath % i))
            c += 1
            if c >= 6 and _zigzag(validationLosses):
                maxLearningRateSwitch -= 1
                if maxLearningRateSwitch == 0:
                    break
                logging.info("New learning rate: %.2e" % m.setLearningRate())
                logging.info("New L2 regularization lambda: %.2e" % m.setL2RegularizationLambda())
                c = 0
            i += 1
            trainLossSum = 0; validationLossSum = 0; datasetPtr = 0; epochStart = time.time()
            XBatch, YBatch, XNum, _ = fetch(datasetPtr, param.trainBatchSize)
            datasetPtr += XNum

    logging.info("Training time elapsed: %.2f s" % (time.time() - trainingStart))

    validationLosses.sort()
    i = validationLosses[0][1]
    logging.info("Best validation loss at batch: %d" % i)

    logging.info("Testing on the training and validation dataset ...")
    predictStart = time.time()
    predictBatchSize = param.predictBatchSize
    datasetPtr = 0
    bases = []; zs = []; ts = []; ls = []
    while True:
        XBatch, _, endFlag = utils.DecompressArray(XArrayCompressed, datasetPtr, predictBatchSize, total)
        base, z, t, l = m.predict(XBatch)
        bases.append(base); zs.append(z); ts.append(t); ls.append(l)
        datasetPtr += predictBatchSize
        if not (datasetPtr < total) or (endFlag != 0 and datasetPtr > predictBatchSize):
            break
    bases = np.concatenate(bases[:]); zs = np.concatenate(zs[:]); ts = np.concatenate(ts[:]); ls = np.concatenate(ls[:])
    logging.info("Prediciton time elapsed: %.2f s" % (time.time() - predictStart))

    YArray, _, _ = utils.DecompressArray(YArrayCompressed, 0, total, total)
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


def main():
    parser = argparse.ArgumentParser(description="Train Clairvoyante")
    parser.add_argument('--bin_fn', type=str, default=None,
                        help="Binary tensor input generated by tensor2Bin.py, tensor_fn, var_fn and bed_fn will be ignored")
    parser.add_argument('--tensor_fn', type=str, default="vartensors", help="Tensor input")
    parser.add_argument('--var_fn', type=str, default="truthvars", help="Truth variants list input")
    parser.add_argument('--bed_fn', type=str, default=None, help="High confident genome regions input in the BED format")
    parser.add_argument('--chkpnt_fn', type=str, default=None, help="Input a checkpoint for testing or continue training")
    parser.add_argument('--learning_rate', type=float, default=param.initialLearningRate,
                        help="Set the initial learning rate, default: %(default)s")
    parser.add_argument('--lambd', type=float, default=param.l2RegularizationLambda,
                        help="Set the l2 regularization lambda, default: %(default)s")
    parser.add_argument('--ochk_prefix', type=str, default=None,
                        help="Prefix for checkpoint outputs at each learning rate change, optional")
    parser.add_argument('--olog_dir', type=str, default=None, help="Directory for tensorboard log outputs, optional")
    parser.add_argument('--v3', type=param.str2bool, nargs='?', const=True, default=True, help="Use Clairvoyante version 3")
    parser.add_argument('--v2', type=param.str2bool, nargs='?', const=True, default=False, help="Use Clairvoyante version 2")
    parser.add_argument('--slim', type=param.str2bool, nargs='?', const=True, default=False,
                        help="Train using the slim version of Clairvoyante, optional")
    args = parser.parse_args()
    if len(sys.argv[1:]) == 0:
        parser.print_help()
        sys.exit(1)
    Run(args)


if __name__ == "__main__":
    main()
