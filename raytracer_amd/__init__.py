"""MI355X path-tracing backend for the reference renderer's hot path.

The product is two shared libraries built by the top-level Makefile:

* ``lib/librt_hip.so``  — HIP kernels behind the C ABI of ``include/rt_hip.h``
* ``lib/librt_host.so`` — the C++ host layer that keeps the reference's scene API

This package only binds them through ctypes (``raytracer_amd.capi``).  There is
no CPU fallback: importing ``capi`` raises if a library is missing.
"""

from . import capi  # noqa: F401

__all__ = ["capi"]
