"""Device buffers for the GPU tests of the *_dev entry points, through the HIP runtime that
libstag_hip.so already brought into the process (ctypes; no torch, so no second copy of the
runtime is initialised next to it)."""
import ctypes

import numpy as np

from stag_amd import _lib

_hip = None


def hip():
    global _hip
    if _hip is None:
        _lib.lib()                                   # loads libamdhip64 as a dependency
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64 not found")
        _hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _hip.hipFree.argtypes = [ctypes.c_void_p]
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with HIP error %d" % (what, rc))


class DevArray:
    """A device copy of a NumPy array (or zeros of a shape/dtype); .get() copies it back."""

    def __init__(self, host=None, shape=None, dtype=None):
        if host is not None:
            host = np.ascontiguousarray(host)
            shape, dtype = host.shape, host.dtype
        self.shape, self.dtype = tuple(np.atleast_1d(shape)) if np.isscalar(shape) else tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctypes.c_void_p()
        _ok(hip().hipSetDevice(0), "hipSetDevice")
        _ok(hip().hipMalloc(ctypes.byref(self.ptr), max(self.nbytes, 16)), "hipMalloc")
        if host is not None and self.nbytes:
            _ok(hip().hipMemcpy(self.ptr, host.ctypes.data_as(ctypes.c_void_p), self.nbytes, 1), "hipMemcpy H2D")
        else:
            _ok(hip().hipMemset(self.ptr, 0, max(self.nbytes, 16)), "hipMemset")

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
            _ok(hip().hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), self.ptr, self.nbytes, 2), "hipMemcpy D2H")
        return out

    def __del__(self):
        try:
            if self.ptr:
                hip().hipFree(self.ptr)
        except Exception:
            pass
