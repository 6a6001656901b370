"""CPU tests of the drop-in boundary: the C-ABI libraries load without a GPU and export every symbol
the headers in include/ declare; the host-only entry points behave like the reference's."""
import ctypes
import json
import os
import re

import pytest

from vbz_compression_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))


def _declared(header, macro):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))  # drop the macro definitions
    return re.findall(macro + r"\s+[^;{]*?\b(\w+)\s*\(", text)


def test_every_declared_symbol_is_exported():
    L = _lib.load()
    names = _declared("vbz.h", "VBZ_EXPORT") + _declared("vbz_gpu.h", "VBZ_EXPORT")
    assert len(names) >= 8 + 16
    assert set(_lib.C_API + _lib.GPU_API) == set(names)
    for n in names:
        assert hasattr(L, n), n
    P = ctypes.CDLL(_lib.PLUGIN_PATH)
    pnames = _declared("vbz_hdf_plugin.h", "VBZ_HDF_PLUGIN_EXPORT")
    assert set(pnames) == {"vbz_filter", "vbz_plugin_info", "H5PLget_plugin_type", "H5PLget_plugin_info"}
    for n in pnames:
        assert hasattr(P, n), n


def test_options_struct_layout_matches_reference():
    # reference vbz/vbz.h:29-53: bool @0, unsigned @4, @8, @12, sizeof 16
    C = _lib.CompressionOptions
    assert ctypes.sizeof(C) == 16
    assert (C.perform_delta_zig_zag.offset, C.integer_size.offset, C.zstd_compression_level.offset, C.vbz_version.offset) == (0, 4, 8, 12)


def test_host_only_entry_points():
    L = _lib.load()
    C = _lib.CompressionOptions
    for k in KAT["size_pins"]:
        o = C(True, 2, k["level"], 1)
        assert L.vbz_max_compressed_size(k["samples"] * 2, ctypes.byref(o)) == k["max"]
    assert L.vbz_max_compressed_size(10, ctypes.byref(C(True, 3, 1, 0))) == _lib.VBZ_INTEGER_SIZE_ERROR
    assert L.vbz_max_compressed_size(10, ctypes.byref(C(True, 2, 1, 7))) == _lib.VBZ_VERSION_ERROR
    assert L.vbz_max_compressed_size(3, ctypes.byref(C(True, 2, 1, 0))) == _lib.VBZ_INPUT_SIZE_ERROR
    assert L.vbz_error_string(_lib.VBZ_STREAMVBYTE_STREAM_ERROR) == b"VBZ_STREAMVBYTE_STREAM_ERROR"
    assert L.vbz_error_string(_lib.VBZ_DEVICE_ERROR) == b"VBZ_DEVICE_ERROR"
    assert L.vbz_error_string(7) == b"VBZ_UNKNOWN_ERROR"
    assert L.vbz_is_error(_lib.VBZ_DEVICE_ERROR) and L.vbz_is_error(_lib.VBZ_ZSTD_ERROR) and not L.vbz_is_error(0xFFFFFFF7)
    buf = (ctypes.c_uint8 * 8)(20, 0, 0, 0, 1, 2, 3, 4)
    assert L.vbz_decompressed_size(buf, 8, ctypes.byref(C(True, 4, 0, 0))) == 20
    assert L.vbz_decompressed_size(buf, 3, ctypes.byref(C(True, 4, 0, 0))) == _lib.VBZ_INPUT_SIZE_ERROR


def test_plugin_descriptor():
    P = ctypes.CDLL(_lib.PLUGIN_PATH)
    P.H5PLget_plugin_type.restype = ctypes.c_int
    assert P.H5PLget_plugin_type() == 0  # H5PL_TYPE_FILTER

    class H5Z(ctypes.Structure):
        _fields_ = [("version", ctypes.c_int), ("id", ctypes.c_int), ("enc", ctypes.c_uint), ("dec", ctypes.c_uint),
                    ("name", ctypes.c_char_p), ("can_apply", ctypes.c_void_p), ("set_local", ctypes.c_void_p), ("filter", ctypes.c_void_p)]

    P.H5PLget_plugin_info.restype = ctypes.POINTER(H5Z)
    info = P.H5PLget_plugin_info().contents
    assert (info.version, info.id, info.enc, info.dec, info.name) == (1, 32020, 1, 1, b"vbz")
    assert info.can_apply is None and info.set_local is None and info.filter


def test_no_silent_cpu_fallback():
    """Without a GPU the compute entry points must fail loudly, never fall back to a CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    L = _lib.load()
    C = _lib.CompressionOptions
    src = (ctypes.c_int16 * 16)(*range(16))
    dst = (ctypes.c_uint8 * 256)()
    r = L.vbz_compress(src, 32, dst, 256, ctypes.byref(C(True, 2, 1, 0)))
    # the reference's own code (a binary compiled against the reference header tests ret >= -7), cause on stderr
    assert r == _lib.VBZ_OUT_OF_MEMORY_ERROR and r >= _lib.VBZ_FIRST_ERROR
    assert not L.vbz_gpu_create(0, None)


def test_fast5_repacker_builds_and_lists_without_a_gpu(tmp_path):
    """bin/vbz_fast5_repack exists after build() and reads a gzip fast5 (no codec work, so no GPU): the samples of
    the reference's test file match the sha256 the golden index holds for them."""
    import hashlib
    import json

    import numpy as np

    from vbz_compression_amd import fast5

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    assert os.path.exists(fast5.TOOL)
    try:
        reads = fast5.list_fast5(os.path.join(golden, "multi_fast5_zip.fast5"), export_signal=str(tmp_path / "sig"))
    except fast5.Hdf5NotFound:
        pytest.skip("no libhdf5 >= 1.10.3 here")
    idx = {e["read"]: e for e in json.load(open(os.path.join(golden, "fast5_chunks.json")))}
    sig = np.fromfile(str(tmp_path / "sig"), np.int16)
    pos = 0
    assert len(reads) == 10
    for r in reads:
        assert r["samples"] == idx[r["name"]]["samples"] and r["filters"] == [1]
        assert hashlib.sha256(sig[pos : pos + r["samples"]].tobytes()).hexdigest() == idx[r["name"]]["raw_sha256"]
        pos += r["samples"]
