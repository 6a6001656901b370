"""CPU tests of the host logic in vbz_compression_amd/csrc/zstd_entropy.h: for the same byte
histogram, the Huffman code lengths and the tree description must be the ones libzstd emits."""
import numpy as np
import pytest

import entropy_host as E
import oracle_lib as O


def _blocks(rng, count):
    for it in range(count):
        n = int(rng.integers(300, 131072))
        kind = it % 4
        if kind == 0:
            data = np.clip(rng.normal(128, rng.uniform(3, 40), n), 0, 255).astype(np.uint8)
        elif kind == 1:
            data = np.minimum(rng.geometric(rng.uniform(0.02, 0.5), n), 255).astype(np.uint8)
        elif kind == 2:
            a = O.synth_signal(5, it, n)
            data = O.svb_compress(a, 2, True, 0)[(n + 3) // 4 :][:n].copy()
        else:
            data = (rng.integers(0, 256, n) * rng.integers(0, 256, n) >> 8).astype(np.uint8)
        yield data


def test_tree_description_identical_to_libzstd():
    if O.lib().vbo_zstd_version() != b"1.4.8":
        pytest.skip("pinned against libzstd 1.4.8")
    rng = np.random.default_rng(5)
    checked = 0
    for data in _blocks(rng, 160):
        frame = O.zstd_compress(data, 1)
        lit = E.parse_first_block_literals(frame)
        # only frames where libzstd found no matches: one block, Huffman literals, zero sequences
        if lit is None or lit[0] != 2 or lit[4] != 0 or lit[1] != len(data):
            continue
        log, nb, tree = E.tree_description(data)
        assert tree == lit[3]
        checked += 1
    assert checked > 100


def test_code_is_complete_and_length_limited():
    rng = np.random.default_rng(6)
    for data in _blocks(rng, 40):
        log, nb, tree = E.tree_description(data)
        assert nb.max() <= 11 and log == nb.max()
        assert sum(2.0 ** -int(x) for x in nb if x) == 1.0


def _build(fn, cnt, maxsym, limit):
    import ctypes

    u8p, u16p, u32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint32)
    nb = np.zeros(256, np.uint8)
    code = np.zeros(256, np.uint16)
    tl = fn(cnt.ctypes.data_as(u32p), maxsym, limit, nb.ctypes.data_as(u8p), code.ctypes.data_as(u16p))
    return tl, nb, code


def test_package_merge_is_optimal_on_small_alphabets():
    """huf_build_pm (the serial statement of what the device builds) against exhaustive search: no complete or incomplete
    prefix code within the length limit is cheaper."""
    import itertools

    H = E.lib()
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(400):
        n = int(rng.integers(2, 7))
        limit = int(rng.integers(max(1, int(np.ceil(np.log2(n)))), 5))
        c = rng.integers(1, 50, n).astype(np.uint32) if rng.random() < 0.7 else (rng.integers(1, 4, n) ** 5).astype(np.uint32)
        cnt = np.zeros(256, np.uint32)
        cnt[:n] = c
        tl, nb, code = _build(H.h_huf_build_pm, cnt, n - 1, limit)
        assert nb[:n].min() >= 1 and nb[:n].max() <= limit and tl == nb[:n].max()
        assert sum(2.0 ** -int(x) for x in nb[:n]) == 1.0
        cost = int((c.astype(np.int64) * nb[:n]).sum())
        best = min(sum(int(a) * b for a, b in zip(c, ls)) for ls in itertools.product(range(1, limit + 1), repeat=n)
                   if sum(2.0 ** -l for l in ls) <= 1.0)
        assert cost == best, (c, limit, nb[:n])
        checked += 1
    assert checked == 400


def test_package_merge_never_longer_than_libzstd_construction():
    """Same histogram, same limit: the package-merge lengths cost at most what libzstd's tree + HUF_setMaxHeight cost,
    the code is complete, canonical and prefix-free."""
    H = E.lib()
    rng = np.random.default_rng(2)
    total_pm = total_z = 0
    for data in _blocks(rng, 60):
        cnt = np.bincount(data, minlength=256).astype(np.uint32)
        if np.count_nonzero(cnt) < 2:
            continue
        maxsym = int(np.nonzero(cnt)[0].max())
        limit = H.h_optimal_table_log(11, min(len(data), 128 << 10), maxsym, 1)
        t1, nb1, c1 = _build(H.h_huf_build, cnt, maxsym, limit)
        t2, nb2, c2 = _build(H.h_huf_build_pm, cnt, maxsym, limit)
        k1, k2 = int((cnt.astype(np.int64) * nb1).sum()), int((cnt.astype(np.int64) * nb2).sum())
        assert k2 <= k1 and nb2.max() <= limit and t2 == nb2.max()
        assert ((nb2 > 0) == (cnt > 0)).all()
        assert sum(2.0 ** -int(x) for x in nb2 if x) == 1.0
        words = sorted(format(int(c2[s]), "0%db" % nb2[s]) for s in range(256) if nb2[s])
        assert all(not b.startswith(a) for a, b in zip(words, words[1:]))
        total_pm += k2
        total_z += k1
    assert total_pm <= total_z
