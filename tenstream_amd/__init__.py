"""tenstream_amd -- MI355X-native back-end for TenStream's pprts diffuse/direct flux solve.

The product is the HIP library `tenstream_amd/lib/libtsx.so` behind the C-ABI in `include/tsx.h`;
this package is the thin host-side mirror used by tests and bench.py (no CPU fallback).
"""
from .solver import DiffuseSolver, KspInfo  # noqa: F401

__all__ = ["DiffuseSolver", "KspInfo"]
