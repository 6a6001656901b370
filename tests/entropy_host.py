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


def weights_from_tree(tree):
    """Huffman code lengths [256] from a tree description (RFC 8878 4.2.1): direct 4-bit weights or FSE-coded weights."""
    t = bytes(tree)
    hb = t[0]
    if hb >= 128:
        nw = hb - 127
        w = []
        for i in range(nw):
            b = t[1 + i // 2]
            w.append(b >> 4 if i % 2 == 0 else b & 15)
    else:
        data = t[1 : 1 + hb]
        val = int.from_bytes(data, "little")
        pos = 0

        def take(k):
            nonlocal pos
            v = (val >> pos) & ((1 << k) - 1)
            pos += k
            return v

        log = take(4) + 5
        remaining, threshold, nbits = (1 << log) + 1, 1 << log, log + 1
        norm, prev0 = [], False
        while remaining > 1 and len(norm) <= 255:
            if prev0:
                while True:
                    r = take(2)
                    norm.extend([0] * r)
                    if r != 3:
                        break
                prev0 = False
                continue
            mx = (2 * threshold - 1) - remaining
            v = (val >> pos) & ((1 << nbits) - 1)
            if (v & (threshold - 1)) < mx:
                c = v & (threshold - 1)
                pos += nbits - 1
            else:
                c = v & (2 * threshold - 1)
                if c >= threshold:
                    c -= mx
                pos += nbits
            c -= 1
            remaining -= abs(c)
            norm.append(c)
            prev0 = c == 0
            while remaining < threshold:
                nbits -= 1
                threshold >>= 1
        assert remaining == 1
        hdr = (pos + 7) // 8
        size = 1 << log
        sym, high = [0] * size, size - 1
        for s, c in enumerate(norm):
            if c == -1:
                sym[high] = s
                high -= 1
        step, p = (size >> 1) + (size >> 3) + 3, 0
        for s, c in enumerate(norm):
            for _ in range(max(c, 0)):
                sym[p] = s
                p = (p + step) & (size - 1)
                while p > high:
                    p = (p + step) & (size - 1)
        nxt = [1 if c == -1 else c for c in norm]
        tab = []
        for u in range(size):
            s = sym[u]
            ns = nxt[s]
            nxt[s] += 1
            nb = log - (ns.bit_length() - 1)
            tab.append((s, nb, (ns << nb) - size))
        stream = int.from_bytes(data[hdr:], "little")
        top = len(data[hdr:]) * 8 - 1
        while not (stream >> top) & 1:
            top -= 1
        bitpos = top  # unread bits are [0, bitpos)
        left = [bitpos]

        def rd(k):
            left[0] -= k
            if left[0] >= 0:
                return (stream >> left[0]) & ((1 << k) - 1)
            have = k + left[0]  # bits that exist; the rest read as zero
            return ((stream & ((1 << max(have, 0)) - 1)) << (k - max(have, 0))) if have > 0 else 0

        s1, s2 = rd(log), rd(log)
        w = []
        while True:
            e = tab[s1]
            w.append(e[0])
            s1 = e[2] + rd(e[1])
            if left[0] < 0:
                w.append(tab[s2][0])
                break
            e = tab[s2]
            w.append(e[0])
            s2 = e[2] + rd(e[1])
            if left[0] < 0:
                w.append(tab[s1][0])
                break
    total = sum((1 << (x - 1)) for x in w if x)
    log2 = total.bit_length()
    rest = (1 << log2) - total
    assert rest and rest & (rest - 1) == 0
    w.append(rest.bit_length())
    nb = np.zeros(256, np.uint8)
    for s, x in enumerate(w):
        nb[s] = log2 + 1 - x if x else 0
    return nb
