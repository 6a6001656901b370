"""ctypes loader of libvbz_hip.so (the C ABI in include/vbz.h and include/vbz_gpu.h).

The library is the product: if it is missing this module raises -- there is no Python or CPU
fallback for either stage of the codec.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VBZ_HIP_LIB: another build of the same library (tools/ab_libs.py; the experiments build lib/libvbz_hip_x.so that carries the
# timed kernel instantiations and the known-slower variants for tools/ and the tests that keep them honest)
LIB_PATH = os.environ.get("VBZ_HIP_LIB") or os.path.join(HERE, "lib", "libvbz_hip.so")
EXPERIMENTS_LIB_PATH = os.path.join(HERE, "lib", "libvbz_hip_x.so")
PLUGIN_PATH = os.path.join(HERE, "lib", "libvbz_hdf_plugin.so")

VBZ_ZSTD_ERROR = 0xFFFFFFFF
VBZ_INPUT_SIZE_ERROR = 0xFFFFFFFE
VBZ_INTEGER_SIZE_ERROR = 0xFFFFFFFD
VBZ_DESTINATION_SIZE_ERROR = 0xFFFFFFFC
VBZ_STREAMVBYTE_STREAM_ERROR = 0xFFFFFFFB
VBZ_VERSION_ERROR = 0xFFFFFFFA
VBZ_OUT_OF_MEMORY_ERROR = 0xFFFFFFF9
VBZ_DEVICE_ERROR = 0xFFFFFFF8  # vbz_gpu.h results only
VBZ_FIRST_ERROR = VBZ_OUT_OF_MEMORY_ERROR  # the reference's value (vbz/vbz.h:22)


class CompressionOptions(ctypes.Structure):
    """struct CompressionOptions of include/vbz.h (reference vbz/vbz.h:29-53), 16 bytes."""

    _fields_ = [
        ("perform_delta_zig_zag", ctypes.c_bool),
        ("integer_size", ctypes.c_uint),
        ("zstd_compression_level", ctypes.c_uint),
        ("vbz_version", ctypes.c_uint),
    ]


class GpuBatch(ctypes.Structure):
    """struct vbz_gpu_batch of include/vbz_gpu.h."""

    _fields_ = [
        ("n_reads", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("src", ctypes.c_void_p),
        ("src_off", ctypes.c_void_p),
        ("src_size", ctypes.c_void_p),
        ("src_bytes", ctypes.c_uint64),
        ("dst", ctypes.c_void_p),
        ("dst_off", ctypes.c_void_p),
        ("dst_cap", ctypes.c_void_p),
        ("dst_bytes", ctypes.c_uint64),
        ("result", ctypes.c_void_p),
    ]


C_API = [
    "vbz_is_error",
    "vbz_error_string",
    "vbz_max_compressed_size",
    "vbz_compress",
    "vbz_decompress",
    "vbz_compress_sized",
    "vbz_decompress_sized",
    "vbz_decompressed_size",
]
GPU_API = [
    "vbz_gpu_create",
    "vbz_gpu_destroy",
    "vbz_gpu_stream",
    "vbz_gpu_last_error",
    "vbz_gpu_set_trailers",
    "vbz_gpu_set_canonical",
    "vbz_gpu_synchronize",
    "vbz_gpu_compress_batch",
    "vbz_gpu_decompress_batch",
    "vbz_gpu_svb_compress_batch",
    "vbz_gpu_svb_decompress_batch",
    "vbz_gpu_zstd_compress_batch",
    "vbz_gpu_zstd_decompress_batch",
    "vbz_gpu_synth_lengths",
    "vbz_gpu_synth_signal",
    "vbz_gpu_synth_u32",
    "vbz_gpu_profile_enable",
    "vbz_gpu_profile_read",
    "vbz_gpu_profile_reset",
    "vbz_gpu_decode_paths",
    "vbz_gpu_decode_literals_ahead",
    "vbz_gpu_decode_span_paths",
    "vbz_gpu_version",
]

_lib = None


def load():
    """Load libvbz_hip.so and declare the prototypes. Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s is missing: build it with `python -m vbz_compression_amd.build` "
            "(there is no CPU fallback for the VBZ codec in this package)" % LIB_PATH
        )
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
    op = ctypes.POINTER(CompressionOptions)
    bp = ctypes.POINTER(GpuBatch)
    L.vbz_is_error.restype = ctypes.c_bool
    L.vbz_is_error.argtypes = [u32]
    L.vbz_error_string.restype = ctypes.c_char_p
    L.vbz_error_string.argtypes = [u32]
    L.vbz_max_compressed_size.restype = u32
    L.vbz_max_compressed_size.argtypes = [u32, op]
    for name in ("vbz_compress", "vbz_decompress", "vbz_compress_sized", "vbz_decompress_sized"):
        f = getattr(L, name)
        f.restype = u32
        f.argtypes = [vp, u32, vp, u32, op]
    L.vbz_decompressed_size.restype = u32
    L.vbz_decompressed_size.argtypes = [vp, u32, op]
    L.vbz_gpu_create.restype = vp
    L.vbz_gpu_create.argtypes = [ctypes.c_int, vp]
    L.vbz_gpu_destroy.restype = None
    L.vbz_gpu_destroy.argtypes = [vp]
    L.vbz_gpu_stream.restype = vp
    L.vbz_gpu_stream.argtypes = [vp]
    L.vbz_gpu_last_error.restype = ctypes.c_char_p
    L.vbz_gpu_last_error.argtypes = [vp]
    L.vbz_gpu_set_trailers.restype = None
    L.vbz_gpu_set_trailers.argtypes = [vp, ctypes.c_int]
    if hasattr(L, "vbz_gpu_set_canonical"):   # (builds of earlier rounds, loaded through VBZ_HIP_LIB, do not have it)
        L.vbz_gpu_set_canonical.restype = None
        L.vbz_gpu_set_canonical.argtypes = [vp, ctypes.c_int]
    L.vbz_gpu_synchronize.restype = ctypes.c_int
    L.vbz_gpu_synchronize.argtypes = [vp]
    for name in ("vbz_gpu_compress_batch", "vbz_gpu_decompress_batch"):
        f = getattr(L, name)
        f.restype = ctypes.c_int
        f.argtypes = [vp, bp, op, ctypes.c_int]
    for name in ("vbz_gpu_svb_compress_batch", "vbz_gpu_svb_decompress_batch"):
        f = getattr(L, name)
        f.restype = ctypes.c_int
        f.argtypes = [vp, bp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.vbz_gpu_zstd_compress_batch.restype = ctypes.c_int
    L.vbz_gpu_zstd_compress_batch.argtypes = [vp, bp, vp]
    L.vbz_gpu_zstd_decompress_batch.restype = ctypes.c_int
    L.vbz_gpu_zstd_decompress_batch.argtypes = [vp, bp]
    L.vbz_gpu_synth_lengths.restype = ctypes.c_int
    L.vbz_gpu_synth_lengths.argtypes = [vp, u64, u64, u32, vp]
    for name in ("vbz_gpu_synth_signal", "vbz_gpu_synth_u32"):
        f = getattr(L, name)
        f.restype = ctypes.c_int
        f.argtypes = [vp, u64, u64, u32, vp, vp, vp]
    L.vbz_gpu_profile_enable.restype = None
    L.vbz_gpu_profile_enable.argtypes = [vp, ctypes.c_int]
    L.vbz_gpu_profile_reset.restype = None
    L.vbz_gpu_profile_reset.argtypes = [vp]
    L.vbz_gpu_profile_read.restype = ctypes.c_int
    L.vbz_gpu_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    L.vbz_gpu_decode_paths.restype = ctypes.c_int
    L.vbz_gpu_decode_paths.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.vbz_gpu_decode_literals_ahead.restype = ctypes.c_int
    L.vbz_gpu_decode_literals_ahead.argtypes = [vp]
    if hasattr(L, "vbz_gpu_decode_span_paths"):   # (tools/ab_libs.py, tools/compare_libs.py load builds of earlier rounds through VBZ_HIP_LIB)
        L.vbz_gpu_decode_span_paths.restype = ctypes.c_int
        L.vbz_gpu_decode_span_paths.argtypes = [vp, ctypes.POINTER(u32)]
    L.vbz_gpu_version.restype = ctypes.c_char_p
    L.vbz_gpu_version.argtypes = []
    _lib = L
    return L


def is_error(v):
    return int(v) >= VBZ_DEVICE_ERROR


def error_string(v):
    return load().vbz_error_string(int(v)).decode()
