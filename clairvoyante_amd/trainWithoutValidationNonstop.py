"""Non-stop training on the WHOLE data set (no validation split): command line, log lines and
checkpoint names of /root/reference/clairvoyante/trainWithoutValidationNonstop.py (:13-111).
The loop lives in trainNonstop.TrainAll(validate=False).

    python -m clairvoyante_amd.trainWithoutValidationNonstop --bin_fn TENSORS.bin --ochk_prefix OUT/model
"""
from . import trainNonstop


def Run(args):
    trainNonstop.Run(args, validate=False)


def TrainAll(args, m, utils):
    trainNonstop.TrainAll(args, m, utils, validate=False)


def main():
    trainNonstop.main(validate=False)


if __name__ == "__main__":
    main()
