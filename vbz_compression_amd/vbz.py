"""numpy front end with the interface of the reference's pyvbz binding
(python/pyvbz/vbz/__init__.py:12-76): compress / decompress / decompressed_size.

Same defaults (zig-zag iff the dtype is signed, zstd level 1, vbz version 0) and the same sized
wire format, but every call runs on the MI355X through the C ABI of libvbz_hip.so.
"""
import ctypes

import numpy as np

from . import _lib


class VbzError(RuntimeError):
    def __init__(self, code):
        self.code = int(code)
        super().__init__(_lib.error_string(code))


def _options(dtype, zigzag, zlevel, version):
    dtype = np.dtype(dtype)
    if zigzag is None:
        zigzag = dtype.kind == "i"  # pyvbz: signed types get delta zig-zag
    return _lib.CompressionOptions(bool(zigzag), dtype.itemsize, int(zlevel), int(version))


def _check(ret):
    if _lib.is_error(ret):
        raise VbzError(ret)
    return int(ret)


def compress(data, zigzag=None, zlevel=1, version=0):
    """Compress a numpy integer array; returns a uint8 array in the sized format
    [u32 original byte count][payload] (reference vbz.cpp:302-330)."""
    L = _lib.load()
    data = np.ascontiguousarray(data)
    opts = _options(data.dtype, zigzag, zlevel, version)
    bound = _check(L.vbz_max_compressed_size(data.nbytes, ctypes.byref(opts)))
    out = np.empty(bound, dtype=np.uint8)
    n = _check(L.vbz_compress_sized(data.ctypes.data if data.size else None, data.nbytes, out.ctypes.data, bound, ctypes.byref(opts)))
    return out[:n].copy()


def decompressed_size(data, dtype, zigzag=None, zlevel=1, version=0):
    L = _lib.load()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    opts = _options(dtype, zigzag, zlevel, version)
    return _check(L.vbz_decompressed_size(data.ctypes.data, data.nbytes, ctypes.byref(opts)))


def decompress(data, dtype, zigzag=None, zlevel=1, version=0):
    """Inverse of compress(); returns an array of `dtype`."""
    L = _lib.load()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    dtype = np.dtype(dtype)
    opts = _options(dtype, zigzag, zlevel, version)
    size = _check(L.vbz_decompressed_size(data.ctypes.data, data.nbytes, ctypes.byref(opts)))
    out = np.empty(max(size, 1), dtype=np.uint8)
    n = _check(L.vbz_decompress_sized(data.ctypes.data, data.nbytes, out.ctypes.data, size, ctypes.byref(opts)))
    return out[:n].view(dtype).copy()


def compress_raw(data, opts, sized=False):
    """vbz_compress / vbz_compress_sized with explicit options; returns bytes or an error code (int)."""
    L = _lib.load()
    data = np.ascontiguousarray(data)
    bound = L.vbz_max_compressed_size(data.nbytes, ctypes.byref(opts))
    if _lib.is_error(bound):
        return int(bound)
    out = np.empty(bound + 16, dtype=np.uint8)
    fn = L.vbz_compress_sized if sized else L.vbz_compress
    n = fn(data.ctypes.data if data.size else None, data.nbytes, out.ctypes.data, bound, ctypes.byref(opts))
    if _lib.is_error(n):
        return int(n)
    return out[:n].copy()


def decompress_raw(data, nbytes, opts, sized=False):
    """vbz_decompress / vbz_decompress_sized with explicit options; returns bytes or an error code (int)."""
    L = _lib.load()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.empty(max(nbytes, 1), dtype=np.uint8)
    fn = L.vbz_decompress_sized if sized else L.vbz_decompress
    n = fn(data.ctypes.data if data.size else None, data.nbytes, out.ctypes.data, nbytes, ctypes.byref(opts))
    if _lib.is_error(n):
        return int(n)
    return out[:n].copy()
