"""hipdrt: MI355X-native (gfx950) implementation of hybrid-drt's data-parallel hot path.

Kernel-matrix construction (hybdrt/matrices) + the coneqp-trajectory QP inside the hierarchical-Bayesian
hyper-parameter loop (hybdrt/models/qphb.py, DRT._qphb_fit_core), batched over spectra.  Host code is
Python calling hand-written FP64 HIP kernels through a ctypes C-ABI (include/hipdrt.h).  There is no CPU
fallback: every compute entry point raises if libhipdrt.so is missing or no GPU is visible.
"""
__version__ = "0.1.0"
