"""MI355X-native kernels for SGLang's RadixAttention + quantized-matmul decode path.

`libsglk.so` (HIP, gfx950) is the product; this package is the Python host side that mirrors the
reference's operator interfaces (sgl_kernel op surface, attention backend, AWQ linear method, radix
cache) on top of its C ABI.
"""
from ._lib import LIB_PATH, SglkError  # noqa: F401

__version__ = "0.1.0"
