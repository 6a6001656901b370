"""numpy front end with the interface of the reference's pyvbz binding
(python/pyvbz/vbz/__init__.py:12-76): compression_options / compress / decompress (+ decompressed_size),
same positional signatures: compress(data, options=None), decompress(data, dtype, options=None).

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


def compression_options(zigzag, size, zlevel=1, version=0):
    """pyvbz.compression_options (python/pyvbz/vbz/__init__.py:12-18): a CompressionOptions object."""
    return _lib.CompressionOptions(bool(zigzag), int(size), int(zlevel), int(version))


def _options(dtype, options, zigzag, zlevel, version):
    """`options` as in pyvbz (an object from compression_options(), or None for the defaults: zig-zag iff the dtype
    is signed, level 1, version 0).  zigzag / zlevel / version are keyword extras that adjust the defaults."""
    if options is not None:
        if not isinstance(options, _lib.CompressionOptions):
            raise TypeError("options must come from compression_options(), not %r (zigzag / zlevel / version are keyword arguments)"
                            % type(options).__name__)
        return options
    dtype = np.dtype(dtype)
    if zigzag is None:
        zigzag = dtype.kind == "i"  # pyvbz: signed types get delta zig-zag
    return compression_options(zigzag, dtype.itemsize, zlevel, version)


def _check(ret):
    if _lib.is_error(ret):
        raise VbzError(ret)
    return int(ret)


def compress(data, options=None, *, zigzag=None, zlevel=1, version=0):
    """pyvbz.compress(data, options=None) (python/pyvbz/vbz/__init__.py:21-44): compress a numpy integer array;
    returns a uint8 array in the sized format [u32 original byte count][payload] (reference vbz.cpp:302-330)."""
    L = _lib.load()
    data = np.ascontiguousarray(data)
    opts = _options(data.dtype, options, zigzag, zlevel, version)
    bound = _check(L.vbz_max_compressed_size(data.nbytes, ctypes.byref(opts)))
    out = np.empty(bound, dtype=np.uint8)
    n = _check(L.vbz_compress_sized(data.ctypes.data if data.size else None, data.nbytes, out.ctypes.data, bound, ctypes.byref(opts)))
    return out[:n]  # a view of the output buffer, as in pyvbz


def decompressed_size(data, dtype, options=None, *, zigzag=None, zlevel=1, version=0):
    L = _lib.load()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    opts = _options(dtype, options, zigzag, zlevel, version)
    return _check(L.vbz_decompressed_size(data.ctypes.data, data.nbytes, ctypes.byref(opts)))


def decompress(data, dtype, options=None, *, zigzag=None, zlevel=1, version=0):
    """pyvbz.decompress(data, dtype, options=None) (python/pyvbz/vbz/__init__.py:47-76); returns an array of `dtype`
    (a view of the output buffer, as in pyvbz)."""
    L = _lib.load()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    dtype = np.dtype(dtype)
    opts = _options(dtype, options, zigzag, zlevel, version)
    size = _check(L.vbz_decompressed_size(data.ctypes.data, data.nbytes, ctypes.byref(opts)))
    out = np.empty(max(size, dtype.itemsize), dtype=np.uint8)
    n = _check(L.vbz_decompress_sized(data.ctypes.data, data.nbytes, out.ctypes.data, size, ctypes.byref(opts)))
    return out[: n - n % dtype.itemsize].view(dtype)


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
    return out[:n]   # (a view of the buffer the library wrote: no second copy of a large result)
