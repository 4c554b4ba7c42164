"""astrild_amd — MI355X (gfx950) native drop-in for astrild's post-processing
hot path: particle->grid mass assignment, 3D-FFT power/bi-spectrum and the
Ray-Ramses weak-lensing kappa-map stack.  Hand-written HIP kernels + rocFFT
behind a ctypes C-ABI (include/astrild_hip.h); no CPU fallback.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
