"""Global constants of the reference (/root/reference/clairvoyante/param.py:1-35),
same names and defaults; drivers mutate NUM_THREADS at run time like the reference."""
import sys

NUM_THREADS = 12
maxEpoch = 10000
parameterOutputPlaceHolder = 6

# Tensor related parameters
flankingBaseNum = 16
matrixNum = 4
bloscBlockSize = 500

# Model hyperparameters
trainBatchSize = 10000
predictBatchSize = 1000
initialLearningRate = 0.001
learningRateDecay = 0.1
maxLearningRateSwitch = 3
trainingDatasetPercentage = 0.9

# Clairvoyante v3 specific
l2RegularizationLambda = 0.001
l2RegularizationLambdaDecay = 0.1
dropoutRateFC4 = 0.5
dropoutRateFC5 = 0.0


def str2bool(v):
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    elif v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    else:
        sys.exit('Boolean value expected.')
