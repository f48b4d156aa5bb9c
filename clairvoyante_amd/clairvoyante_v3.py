"""Drop-in for /root/reference/clairvoyante/clairvoyante_v3.py: `import clairvoyante_v3 as cv;
m = cv.Clairvoyante()` gives the v3 full topology (conv k(1,4)x16 / k(2,4)x32 / k(3,4)x48 with
max-pools (5,1)/(4,1)/(3,1), fc4 336, fc5 168) on the MI355X kernels."""
from .model import Clairvoyante  # noqa: F401
