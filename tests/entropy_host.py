"""Host harness around vbz_compression_amd/csrc/zstd_entropy.h (tests only): the serial statement of
the Huffman table construction, compiled with g++, driven through ctypes."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host", "entropy_harness.cpp")
HDR = os.path.join(ROOT, "vbz_compression_amd", "csrc", "zstd_entropy.h")
SO = os.path.join(HERE, "host", "libentropy_harness.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(SO):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.dirname(HDR), "-o", SO, SRC])
        _lib = ctypes.CDLL(SO)
    return _lib


def tree_description(data, package_merge=False):
    """(table log, code lengths[256], tree description bytes) for the byte histogram of `data`,
    as libzstd's HUF_buildCTable / HUF_writeCTable produce them for a block of len(data) <= 128 KiB;
    package_merge: code lengths by huf_build_pm (what the device encoder builds) instead of libzstd's construction."""
    H = lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    cnt = np.bincount(data, minlength=256).astype(np.uint32)
    maxsym = int(np.nonzero(cnt)[0].max())
    u8p, u16p, u32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint32)
    n = min(len(data), 128 << 10)
    hl = H.h_optimal_table_log(11, n, maxsym, 1)
    nb = np.zeros(256, np.uint8)
    code = np.zeros(256, np.uint16)
    tl = (H.h_huf_build_pm if package_merge else H.h_huf_build)(cnt.ctypes.data_as(u32p), maxsym, hl, nb.ctypes.data_as(u8p), code.ctypes.data_as(u16p))
    out = np.zeros(300, np.uint8)
    ts = H.h_huf_write_tree(out.ctypes.data_as(u8p), 300, nb.ctypes.data_as(u8p), maxsym, tl)
    return tl, nb, (bytes(out[:ts]) if ts > 0 else None)


def parse_first_block_literals(frame):
    """For a zstd frame: (literals type, regenerated size, compressed size, tree description bytes or None,
    number-of-sequences byte) of its first block, or None if that block is not a compressed block."""
    b = bytes(frame)
    fhd = b[4]
    single = (fhd >> 5) & 1
    pos = 5 + (0 if single else 1) + [1 if single else 0, 2, 4, 8][fhd >> 6]
    bh = b[pos] | b[pos + 1] << 8 | b[pos + 2] << 16
    pos += 3
    if (bh >> 1) & 3 != 2:
        return None
    v = int.from_bytes(b[pos : pos + 5], "little")
    t, fmt = v & 3, (v >> 2) & 3
    if t < 2:
        return (t, None, None, None, None)
    if fmt < 2:
        regen, cs, h = (v >> 4) & 0x3FF, (v >> 14) & 0x3FF, 3
    elif fmt == 2:
        regen, cs, h = (v >> 4) & 0x3FFF, (v >> 18) & 0x3FFF, 4
    else:
        regen, cs, h = (v >> 4) & 0x3FFFF, (v >> 22) & 0x3FFFF, 5
    q = pos + h
    tree = None
    if t == 2:
        hb = b[q]
        tl = 1 + hb if hb < 128 else 1 + (hb - 127 + 1) // 2
        tree = b[q : q + tl]
    return (t, regen, cs, tree, b[q + cs])
