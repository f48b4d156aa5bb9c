"""Drop-in for /root/reference/clairvoyante/clairvoyante_v3_slim.py:9-11 -- conv k(1,4)x8,
k(3,4)x16, k(5,4)x32, no pooling layers, fc4 36, fc5 18."""
from . import model as _model


class Clairvoyante(_model.Clairvoyante):
    def __init__(self, **kw):
        slim = dict(kernelSize1=(1, 4), kernelSize2=(3, 4), kernelSize3=(5, 4),
                    pollSize1=None, pollSize2=None, pollSize3=None,
                    numFeature1=8, numFeature2=16, numFeature3=32,
                    hiddenLayerUnits4=36, hiddenLayerUnits5=18)
        slim.update(kw)
        super(Clairvoyante, self).__init__(**slim)
