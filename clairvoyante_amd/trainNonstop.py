"""Non-stop training driver: train.py's epoch schedule without the learning-rate decay, the early
stop and the final report -- one checkpoint per epoch until param.maxEpoch.  Same command line,
log lines and checkpoint names as /root/reference/clairvoyante/trainNonstop.py (Run :13-36,
TrainAll :39-129) and, with `validate=False`, as trainWithoutValidationNonstop.py (TrainAll
:39-111).

    python -m clairvoyante_amd.trainNonstop --bin_fn TENSORS.bin --ochk_prefix OUT/model

Schedule facts kept from the reference:
  * with validation the batch sequence is train.py's (`train.run_epoch`): trained iff the batch ends
    strictly before validationStart, last batch evaluated synchronously (trainNonstop.py:83-117);
  * without validation every batch is 10 000 (clipped at the end of the data set), the loss is
    divided by `total`, and the batch that reaches the end of the data set is fetched but NEVER
    trained -- the epoch is closed as soon as its end flag is seen
    (trainWithoutValidationNonstop.py:80-104);
  * --ochk_prefix is mandatory (:32-33).
"""
import logging
import os
import sys
import time

if __package__ in (None, ""):      # run as `python <dir>/trainNonstop.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param
from .train import _BatchStream, _Job, build_parser, load_dataset, pick_model, run_epoch

logging.basicConfig(format='%(message)s', level=logging.INFO)


def Run(args, validate=True):
    logging.info("Initializing model ...")
    from . import parallel
    from . import utils_v2 as utils
    cv = pick_model(args)
    utils.SetupEnv()
    parallel.init_from_env()
    m = cv.Clairvoyante()
    m.init()
    if args.ochk_prefix is None:
        sys.exit("--chk_prefix must be defined in nonstop training mode")
    if args.chkpnt_fn is not None:
        m.restoreParameters(os.path.abspath(args.chkpnt_fn))
    parallel.broadcast_parameters(m)
    TrainAll(args, m, utils, validate)


def _epoch_all_training(stream, m, rank, ws, writer, epoch):
    """trainWithoutValidationNonstop.py:76-104"""
    size = param.trainBatchSize
    total_loss = 0
    stream.rewind()
    X, Y, _start, _count, _ = stream.fetch(size)
    while True:
        job = _Job(m.trainNoRT, X, Y)                 # the stream hands out this rank's slice
        job.start()
        nxt = stream.fetch(size)
        job.finish()
        total_loss += m.trainLossRTVal
        if writer is not None:
            writer.add_summary(m.trainSummaryRTVal, epoch)
        X, Y, _start, _count, last = nxt
        if last:
            return total_loss


def TrainAll(args, m, utils, validate=True):
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
    if validate:
        trainingTotal = int(total * param.trainingDatasetPercentage)
        validationStart = trainingTotal + 1
        numValItems = total - validationStart
    else:
        trainingTotal = total
        validationStart = total + 1
    stream = _BatchStream(utils, XC, YC, total, validationStart, rank, ws)
    epoch = 1 if args.chkpnt_fn is None else int(args.chkpnt_fn[-param.parameterOutputPlaceHolder:]) + 1
    while epoch < param.maxEpoch:
        t_epoch = time.time()
        if validate:
            train_sum, val_sum = run_epoch(stream, m, rank, ws, writer, epoch, validationStart)
            logging.info(" ".join([str(epoch), "Training loss:", str(train_sum / trainingTotal), "Validation loss: ",
                                   str(val_sum / numValItems)]))
        else:
            train_sum = _epoch_all_training(stream, m, rank, ws, writer, epoch)
            logging.info(" ".join([str(epoch), "Training loss:", str(train_sum / trainingTotal)]))
        logging.info("Epoch time elapsed: %.2f s" % (time.time() - t_epoch))
        if rank == 0:
            m.saveParameters(os.path.abspath("%s-%0*d" % (args.ochk_prefix, param.parameterOutputPlaceHolder, epoch)))
        epoch += 1
    logging.info("Training time elapsed: %.2f s" % (time.time() - t_begin))


def main(validate=True):
    parser = build_parser("Train Clairvoyante Nonstop")
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    Run(args, validate)


if __name__ == "__main__":
    main()
