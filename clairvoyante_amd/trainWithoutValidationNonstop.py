"""Non-stop training on the WHOLE data set (no validation split): command line, log lines and
checkpoint names of /root/reference/clairvoyante/trainWithoutValidationNonstop.py (:13-111).
The loop lives in trainNonstop.TrainAll(validate=False).

    python -m clairvoyante_amd.trainWithoutValidationNonstop --bin_fn TENSORS.bin --ochk_prefix OUT/model
"""
if __package__ in (None, ""):      # run as `python <dir>/trainWithoutValidationNonstop.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import trainNonstop


def Run(args):
    trainNonstop.Run(args, validate=False)


def TrainAll(args, m, utils):
    trainNonstop.TrainAll(args, m, utils, validate=False)


def main():
    trainNonstop.main(validate=False)


if __name__ == "__main__":
    main()
