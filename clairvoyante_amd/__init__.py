"""MI355X-native Clairvoyante v3 pileup-CNN path (drop-in for the reference's
clairvoyante_v3 / clairvoyante_v3_slim model classes and the callVar / train loops).

All arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of
include/clairvoyante_amd.h; this package is the host-side mirror of the reference's
Python interface.  There is no CPU fallback: importing the model classes without
the built extension, or without a GPU, raises.
"""
__version__ = "0.1.0"
