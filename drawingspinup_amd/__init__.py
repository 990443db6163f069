"""drawingspinup_amd — MI355X (gfx950) hot path of DrawingSpinUp.

Host side: Python mirrors of the reference's operator interfaces; device side: hand-written
HIP kernels behind the C ABI of include/dsu_hip.h (libdsu_hip.so, loaded by ._lib).
"""
__version__ = "0.1.0"
