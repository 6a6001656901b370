"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ERRORS = {
    0xFFFFFFFF: "VBZ_ZSTD_ERROR",
    0xFFFFFFFE: "VBZ_INPUT_SIZE_ERROR",
    0xFFFFFFFD: "VBZ_INTEGER_SIZE_ERROR",
    0xFFFFFFFC: "VBZ_DESTINATION_SIZE_ERROR",
    0xFFFFFFFB: "VBZ_STREAMVBYTE_STREAM_ERROR",
    0xFFFFFFFA: "VBZ_VERSION_ERROR",
    0xFFFFFFF9: "VBZ_OUT_OF_MEMORY_ERROR",
}
FIRST_ERROR = 0xFFFFFFF9


class Options(ctypes.Structure):
    _fields_ = [
        ("perform_delta_zig_zag", ctypes.c_bool),
        ("integer_size", ctypes.c_uint),
        ("zstd_compression_level", ctypes.c_uint),
        ("vbz_version", ctypes.c_uint),
    ]


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("vbz_oracle.c", "zstd_restate.c", "vbz_oracle_bench.c", "vbz_oracle_fuzz.c", "vbz_oracle_simd.c", "vbz_oracle.h", "Makefile")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build_oracle())
        vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
        op = ctypes.POINTER(Options)
        L.vbo_max_streamvbyte_size.restype = u32
        L.vbo_max_streamvbyte_size.argtypes = [sz, u32]
        for name in ("vbo_streamvbyte_compress", "vbo_streamvbyte_decompress"):
            f = getattr(L, name)
            f.restype = u32
            f.argtypes = [vp, u32, vp, u32, ctypes.c_int, ctypes.c_bool, ctypes.c_uint]
        L.vbo_max_compressed_size.restype = u32
        L.vbo_max_compressed_size.argtypes = [u32, op]
        for name in ("vbo_compress", "vbo_decompress", "vbo_compress_sized", "vbo_decompress_sized"):
            f = getattr(L, name)
            f.restype = u32
            f.argtypes = [vp, u32, vp, u32, op]
        L.vbo_decompressed_size.restype = u32
        L.vbo_decompressed_size.argtypes = [vp, u32, op]
        L.vbo_error_string.restype = ctypes.c_char_p
        L.vbo_error_string.argtypes = [u32]
        L.vbo_is_error.restype = ctypes.c_bool
        L.vbo_is_error.argtypes = [u32]
        L.vbo_filter.restype = sz
        L.vbo_filter.argtypes = [ctypes.c_uint, sz, ctypes.POINTER(ctypes.c_uint), sz, ctypes.POINTER(sz), ctypes.POINTER(vp)]
        L.vbo_zstd_version.restype = ctypes.c_char_p
        L.vbo_zstd_compress.restype = sz
        L.vbo_zstd_compress.argtypes = [vp, sz, vp, sz, ctypes.c_int]
        L.vbo_zstd_decompress.restype = sz
        L.vbo_zstd_decompress.argtypes = [vp, sz, vp, sz]
        L.vbo_zstd_bound.restype = sz
        L.vbo_zstd_bound.argtypes = [sz]
        L.vbo_zstd_content_size.restype = ctypes.c_ulonglong
        L.vbo_zstd_content_size.argtypes = [vp, sz]
        L.vbo_zstd_restate_decompress.restype = sz
        L.vbo_zstd_restate_decompress.argtypes = [vp, sz, vp, sz]
        L.vbo_mix64.restype = ctypes.c_uint64
        L.vbo_mix64.argtypes = [ctypes.c_uint64]
        L.vbo_synth_read_length.restype = u32
        L.vbo_synth_read_length.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        L.vbo_synth_signal.restype = None
        L.vbo_synth_signal.argtypes = [ctypes.c_uint64, ctypes.c_uint64, vp, sz]
        L.vbo_synth_u32.restype = None
        L.vbo_synth_u32.argtypes = [ctypes.c_uint64, ctypes.c_uint64, vp, sz]
        L.vbo_bench_roundtrip.restype = ctypes.c_int
        L.vbo_bench_roundtrip.argtypes = [u32, ctypes.c_int, ctypes.c_double, op, ctypes.POINTER(ctypes.c_double)]
        L.vbo_bench_roundtrip_u32.restype = ctypes.c_int
        L.vbo_bench_roundtrip_u32.argtypes = [u32, u32, ctypes.c_int, ctypes.c_double, op, ctypes.POINTER(ctypes.c_double)]
        L.vbo_simd_available.restype = ctypes.c_int
        L.vbo_use_simd_svb.restype = None
        L.vbo_use_simd_svb.argtypes = [ctypes.c_int]
        for name in ("vbo_i16zz_compress_simd",):
            getattr(L, name).restype = u32
            getattr(L, name).argtypes = [vp, u32, vp]
        L.vbo_i16zz_decompress_simd.restype = u32
        L.vbo_i16zz_decompress_simd.argtypes = [vp, u32, vp, u32]
        L.vbo_fuzz_max_destination.restype = u32
        L.vbo_fuzz_max_destination.argtypes = [u32, op]
        L.vbo_fuzz_decompress_sweep.restype = ctypes.c_int
        L.vbo_fuzz_decompress_sweep.argtypes = [vp, u32, op, u32, vp]
        _lib = L
    return _lib


def fuzz_sweep(data, opts):
    """Replay of the reference fuzz target's decompress half for one input and option set (oracle/vbz_oracle_fuzz.c):
    returns (max_destination, results[max_destination + 1, 2]) -- column 0 vbz_decompress, column 1 vbz_decompress_sized."""
    a, p, n = _buf(np.frombuffer(bytes(data), np.uint8))
    G = lib().vbo_fuzz_max_destination(n, ctypes.byref(opts))
    res = np.zeros((G + 1, 2), np.uint32)
    rc = lib().vbo_fuzz_decompress_sweep(p, n, ctypes.byref(opts), G, res.ctypes.data)
    assert rc == 0
    return G, res


def options(zigzag=True, size=2, level=1, version=0):
    return Options(bool(zigzag), int(size), int(level), int(version))


def _buf(a):
    a = np.ascontiguousarray(a)
    return a, (a.ctypes.data if a.size else None), a.nbytes


def is_error(v):
    return v >= FIRST_ERROR


def svb_compress(arr, size, zigzag, version=0):
    a, p, n = _buf(arr)
    cap = lib().vbo_max_streamvbyte_size(size, n)
    assert not is_error(cap), ERRORS.get(cap)
    out = np.zeros(cap + 32, np.uint8)
    r = lib().vbo_streamvbyte_compress(p, n, out.ctypes.data, cap, size, zigzag, version)
    if is_error(r):
        return r
    return out[:r].copy()


def svb_decompress(buf, nbytes, size, zigzag, version=0):
    a, p, n = _buf(np.frombuffer(bytes(buf), np.uint8) if not isinstance(buf, np.ndarray) else buf)
    out = np.zeros(max(nbytes, 1), np.uint8)
    r = lib().vbo_streamvbyte_decompress(p, n, out.ctypes.data, nbytes, size, zigzag, version)
    if is_error(r):
        return r
    return out[:r].copy()


SIMD_DECLINED = 0xFFFFFF9C   # (vbo_size_t)-100: not a stream the SSSE3 decoder takes


def simd_available():
    return bool(lib().vbo_simd_available())


def i16zz_compress_simd(arr):
    """The SSSE3 form of the int16 zig-zag stage (oracle/vbz_oracle_simd.c: CPU baseline leg only)."""
    a, p, n = _buf(arr)
    out = np.zeros(n // 2 // 4 + n + 64, np.uint8)
    r = lib().vbo_i16zz_compress_simd(p, n, out.ctypes.data)
    return out[:r].copy()


def i16zz_decompress_simd(buf, nbytes):
    a, p, n = _buf(np.frombuffer(bytes(buf), np.uint8) if not isinstance(buf, np.ndarray) else buf)
    src = np.zeros(n + 64, np.uint8)   # (the vector loads stay inside the stream by the 32-byte rule; slack for safety)
    src[:n] = a.view(np.uint8).reshape(-1)[:n]
    out = np.zeros(max(nbytes, 1) + 16, np.uint8)
    r = lib().vbo_i16zz_decompress_simd(src.ctypes.data, n, out.ctypes.data, nbytes)
    if r >= FIRST_ERROR or r == SIMD_DECLINED:
        return r
    return out[:r].copy()


def max_compressed_size(nbytes, opts):
    return lib().vbo_max_compressed_size(nbytes, ctypes.byref(opts))


def compress(arr, opts, sized=False):
    a, p, n = _buf(arr)
    cap = max_compressed_size(n, opts)
    if is_error(cap):
        return cap
    out = np.zeros(cap + 32, np.uint8)
    fn = lib().vbo_compress_sized if sized else lib().vbo_compress
    r = fn(p, n, out.ctypes.data, cap, ctypes.byref(opts))
    if is_error(r):
        return r
    return out[:r].copy()


def decompress(buf, nbytes, opts, sized=False):
    a, p, n = _buf(buf if isinstance(buf, np.ndarray) else np.frombuffer(bytes(buf), np.uint8))
    out = np.zeros(max(nbytes, 1), np.uint8)
    fn = lib().vbo_decompress_sized if sized else lib().vbo_decompress
    r = fn(p, n, out.ctypes.data, nbytes, ctypes.byref(opts))
    if is_error(r):
        return r
    return out[:r].copy()


def zstd_compress(data, level=1):
    a, p, n = _buf(data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), np.uint8))
    cap = lib().vbo_zstd_bound(n)
    out = np.zeros(cap + 8, np.uint8)
    r = lib().vbo_zstd_compress(out.ctypes.data, cap, p, n, level)
    assert r != 2**64 - 1
    return out[:r].copy()


def zstd_decompress(frame, cap):
    a, p, n = _buf(frame if isinstance(frame, np.ndarray) else np.frombuffer(bytes(frame), np.uint8))
    out = np.zeros(max(cap, 1), np.uint8)
    r = lib().vbo_zstd_decompress(out.ctypes.data, cap, p, n)
    if r == 2**64 - 1:
        return None
    return out[:r].copy()


def zstd_restate_decompress(frame, cap):
    a, p, n = _buf(frame if isinstance(frame, np.ndarray) else np.frombuffer(bytes(frame), np.uint8))
    out = np.zeros(max(cap, 1), np.uint8)
    r = lib().vbo_zstd_restate_decompress(out.ctypes.data, cap, p, n)
    if r == 2**64 - 1:
        return None
    return out[:r].copy()


def zstd_content_size(frame):
    a, p, n = _buf(frame if isinstance(frame, np.ndarray) else np.frombuffer(bytes(frame), np.uint8))
    return lib().vbo_zstd_content_size(p, n)


def synth_signal(seed, read_index, n):
    out = np.zeros(n, np.int16)
    lib().vbo_synth_signal(seed, read_index, out.ctypes.data, n)
    return out


def synth_u32(seed, read_index, n):
    out = np.zeros(n, np.uint32)
    lib().vbo_synth_u32(seed, read_index, out.ctypes.data, n)
    return out


def synth_read_length(seed, read_index):
    return lib().vbo_synth_read_length(seed, read_index)


def fnv1a64(data):
    h = 0xCBF29CE484222325
    for b in bytes(data):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def bench_roundtrip(n_reads, threads, min_seconds, opts, u32_count=0, simd=False):
    """Threaded CPU timing of encode+decode over reads [0, n_reads) of the synthetic workload (u32_count: buffers of that
    many uint32 values of the config-4 generator instead of int16 reads).  simd: the int16 zig-zag stage in its SSSE3 form."""
    out = (ctypes.c_double * 6)()
    lib().vbo_use_simd_svb(1 if simd else 0)
    try:
        if u32_count:
            rc = lib().vbo_bench_roundtrip_u32(n_reads, u32_count, threads, min_seconds, ctypes.byref(opts), out)
        else:
            rc = lib().vbo_bench_roundtrip(n_reads, threads, min_seconds, ctypes.byref(opts), out)
    finally:
        lib().vbo_use_simd_svb(0)
    if rc != 0:
        raise RuntimeError("oracle bench failed (%d)" % rc)
    return dict(raw_bytes=out[0], comp_bytes=out[1], best_s=out[2], enc_thread_s=out[3], dec_thread_s=out[4], passes=int(out[5]))
