"""Writes the `.bin` training file: four back-to-back pickles (total, X blocks, Y blocks, position
blocks) of 500-item blosc chunks -- same command line and layout as
/root/reference/clairvoyante/tensor2Bin.py (Convert :16-28).

    python -m clairvoyante_amd.tensor2Bin --tensor_fn T.gz --var_fn V.gz --bed_fn B.bed --bin_fn OUT.bin
"""
import argparse
import logging
import pickle
import sys

if __package__ in (None, ""):      # run as `python <dir>/tensor2Bin.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param

logging.basicConfig(format='%(message)s', level=logging.INFO)


def Convert(args, utils):
    logging.info("Loading the dataset ...")
    total, XC, YC, PC = utils.GetTrainingArray(args.tensor_fn, args.var_fn, args.bed_fn)
    logging.info("Writing to binary ...")
    with open(args.bin_fn, "wb") as fh:
        for obj in (total, XC, YC, PC):
            pickle.dump(obj, fh)


def Run(args):
    from . import utils_v2 as utils
    utils.SetupEnv()
    Convert(args, utils)


def main():
    parser = argparse.ArgumentParser(description="Generate a binary format input tensor")
    for flag, default, text in (("--tensor_fn", "vartensors", "Tensor input"), ("--var_fn", "truthvars", "Truth variants list input"),
                                ("--bed_fn", None, "High confident genome regions input in the BED format"),
                                ("--bin_fn", None, "Output a binary tensor file")):
        parser.add_argument(flag, type=str, default=default, help=text)
    for flag, default, text in (("--v3", True, "Use Clairvoyante version 3"), ("--v2", False, "Use Clairvoyante version 2")):
        parser.add_argument(flag, type=param.str2bool, nargs='?', const=True, default=default, help=text)
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    Run(args)


if __name__ == "__main__":
    main()
