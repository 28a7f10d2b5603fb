"""vilma_amd: the `vilma fit` variational-inference hot path on AMD Instinct MI355X.

Python host code (the reference's CLI / class API / file formats) over hand-written HIP
kernels for gfx950 reached through a C-ABI (include/vilma_hip.h, libvilma_hip.so).
"""
VERSION = '0.0.16+mi355x.1'
