"""skoots_amd -- MI355X-native implementation of the SKOOTS volumetric-inference hot path.

Drop-in for ``skoots.lib.eval.eval`` and the library functions it calls
(reference: buswinka/skoots, ``skoots/lib``).  All compute runs in hand-written HIP
kernels (``libskoots_hip.so``, C ABI in ``include/skoots_hip.h``); there is no CPU or
eager-PyTorch fallback: importing :mod:`skoots_amd._ffi` fails loudly when the
library is missing.
"""
__version__ = "0.1.0"
