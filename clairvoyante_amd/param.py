"""Run-time constants of the hot path -- the same names, values and mutability as the reference's
`param` module (/root/reference/clairvoyante/param.py:1-35), which drivers read and sometimes
overwrite (callVar.py:40,42 sets NUM_THREADS).  Kept as one table so each value carries its
reference line."""
import sys

_TABLE = (
    # name, value, reference line
    ("NUM_THREADS", 12, "param.py:1   (host threads; meaningless for the GPU kernels, kept for the CLI)"),
    ("maxEpoch", 10000, "param.py:2"),
    ("parameterOutputPlaceHolder", 6, "param.py:3   (digits of the epoch suffix of checkpoint names)"),
    ("flankingBaseNum", 16, "param.py:6   (tensor = 2*16+1 positions)"),
    ("matrixNum", 4, "param.py:7"),
    ("bloscBlockSize", 500, "param.py:8   (items per compressed block of the .bin file)"),
    ("trainBatchSize", 10000, "param.py:11"),
    ("predictBatchSize", 1000, "param.py:12"),
    ("initialLearningRate", 0.001, "param.py:13"),
    ("learningRateDecay", 0.1, "param.py:14"),
    ("maxLearningRateSwitch", 3, "param.py:15"),
    ("trainingDatasetPercentage", 0.9, "param.py:16"),
    ("l2RegularizationLambda", 0.001, "param.py:19"),
    ("l2RegularizationLambdaDecay", 0.1, "param.py:20"),
    ("dropoutRateFC4", 0.5, "param.py:21"),
    ("dropoutRateFC5", 0.0, "param.py:22"),
)
for _name, _value, _where in _TABLE:
    globals()[_name] = _value
del _name, _value, _where

_TRUE = frozenset(("yes", "true", "t", "y", "1"))
_FALSE = frozenset(("no", "false", "f", "n", "0"))


def str2bool(v):
    """argparse type of the tri-state --v2/--v3/--slim/--showRef flags (param.py:28-35)"""
    key = v.lower()
    if key in _TRUE:
        return True
    if key in _FALSE:
        return False
    sys.exit('Boolean value expected.')
