"""GPU tests (-m gpu) of the large-read path: batches of few, large reads (BASELINE.json configs[0] and [3]: one 400 k-sample
int16 read, one 10 M-element uint32 buffer) are spread over many workgroups -- segmented svb kernels, one wavefront per
span of the entropy stage, a span index behind the frame.  Results must be what the one-workgroup-per-read kernels
give: svb bytes identical to the oracle's, frames that the reference's decoder (oracle + libzstd) reads, decoding of
the oracle's frames, and an index that is verified, never trusted."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O
from vbz_compression_amd import _lib, batch

import gpu_util as G

pytestmark = pytest.mark.gpu

IDX_MAGIC, CP_MAGIC = 0x184D2A5C, 0x184D2A5B


def _trailers(f):
    """(body, [trailers...]) of a compressed buffer: the skippable frames this library appends, found from the end"""
    out = []
    while len(f) >= 24:
        tb = int(f[-4:].view("<u4")[0])
        if tb < 16 or tb > len(f) - 9:
            break
        m = f[len(f) - tb : len(f) - tb + 8].view("<u4")
        if int(m[0]) not in (IDX_MAGIC, CP_MAGIC) or int(m[1]) != tb - 8:
            break
        out.append(f[len(f) - tb :])
        f = f[: len(f) - tb]
    return f, out


def _large_cases():
    rng = np.random.default_rng(8)
    cases = [
        ("config1 int16 400k", O.synth_signal(5, 0, 400000), (True, 2, 1, 1)),
        ("int16 1.3M wraps", rng.integers(-32768, 32767, 1300003, endpoint=True).astype(np.int16), (True, 2, 1, 0)),
        ("config4 uint32 3M", O.synth_u32(5, 3, 3_000_000), (False, 4, 3, 0)),
        ("int32 zigzag 700k", rng.integers(-2**20, 2**20, 700001).astype(np.int32), (True, 4, 1, 0)),
        ("int8 2M", rng.integers(-100, 100, 2_000_003).astype(np.int8), (True, 1, 1, 0)),
        ("uint16 no zz 900k", rng.integers(0, 40000, 900001).astype(np.uint16), (False, 2, 1, 0)),
        ("zeros int16 600k", np.zeros(600000, np.int16), (True, 2, 1, 1)),
    ]
    return cases


@pytest.mark.parametrize("name,a,o", _large_cases(), ids=[c[0] for c in _large_cases()])
def test_large_read_both_ways(name, a, o):
    """One large read per call (n_reads = 1: the batch rule picks the large-read path by itself)."""
    go = _lib.CompressionOptions(*o)
    oo = O.options(*o)
    for sized in (False, True):
        # svb stage alone: byte-identical to the oracle
        g0 = G.compress([a], _lib.CompressionOptions(o[0], o[1], 0, o[3]), sized=sized)[0]
        r0 = O.compress(a, O.options(o[0], o[1], 0, o[3]), sized=sized)
        assert not isinstance(g0, int) and g0.tobytes() == r0.tobytes()
        assert G.decompress([r0], [a.nbytes], _lib.CompressionOptions(o[0], o[1], 0, o[3]), sized=sized)[0].tobytes() == a.tobytes()
        # whole path
        g = G.compress([a], go, sized=sized)[0]
        assert not isinstance(g, int), hex(g)
        back = O.decompress(g, a.nbytes, oo, sized=sized)             # the reference's decoder reads the device's frame
        assert not isinstance(back, int) and back.tobytes() == a.tobytes()
        r = O.compress(a, oo, sized=sized)
        mine = G.decompress([r, g], [a.nbytes, a.nbytes], go, sized=sized)
        assert not isinstance(mine[0], int) and mine[0].tobytes() == a.tobytes()    # libzstd's frame (no index: ordinary decoder)
        assert not isinstance(mine[1], int) and mine[1].tobytes() == a.tobytes()    # own frame, decoded in spans
        assert len(r) < 2000 or len(g) <= len(r) * 1.03, (len(g), len(r))
        body, tr = _trailers(g[4:] if sized else g)
        if a.nbytes * 1.0 > (600 << 10) and a.any():
            assert tr and int(tr[0][:4].view("<u4")[0]) == IDX_MAGIC, "a frame of several spans carries the span index"


def test_span_index_is_verified_not_trusted():
    """Damage to the index (offsets that are no block boundaries, swapped entries, wrong counts, a missing or truncated
    trailer) costs speed, never correctness: the frame still decodes to the same samples, as libzstd says it should."""
    a = O.synth_signal(5, 1, 1_500_000)
    o = (True, 2, 1, 1)
    go, oo = _lib.CompressionOptions(*o), O.options(*o)
    g = G.compress([a], go)[0]
    body, tr = _trailers(g)
    assert tr and int(tr[0][:4].view("<u4")[0]) == IDX_MAGIC
    idx = tr[0]
    ns = int(idx[8:12].view("<u4")[0]) & 0x7FFFFFFF   # (bit 31: spans that begin with a treeless block)
    assert ns >= 6 and len(idx) == 16 + 8 * ns
    rng = np.random.default_rng(4)
    variants = [g, np.ascontiguousarray(g[: len(g) - len(idx)])]
    for k in range(24):
        d = g.copy()
        base = len(g) - len(idx)
        e = base + 12 + 8 * int(rng.integers(0, ns))
        if k % 4 == 0:      # a frame offset that is not a block boundary
            d[e : e + 4] = np.array([int(d[e : e + 4].view("<u4")[0]) + int(rng.integers(1, 9))], "<u4").view(np.uint8)
        elif k % 4 == 1:    # a content offset that is off
            d[e + 4 : e + 8] = np.array([int(d[e + 4 : e + 8].view("<u4")[0]) + int(rng.integers(1, 4096))], "<u4").view(np.uint8)
        elif k % 4 == 2:    # a bit anywhere in the trailer
            at = base + int(rng.integers(0, len(idx)))
            d[at] ^= 1 << int(rng.integers(0, 8))
        else:               # two entries swapped
            e2 = base + 12 + 8 * int(rng.integers(0, ns))
            t = d[e : e + 8].copy()
            d[e : e + 8] = d[e2 : e2 + 8]
            d[e2 : e2 + 8] = t
        variants.append(d)
    # an index that points every span at a TRUE block boundary of a later span (treeless blocks there): must not be believed
    d = g.copy()
    base = len(g) - len(idx)
    ent = d[base + 12 : base + 12 + 8 * ns].view("<u4").reshape(ns, 2).copy()
    ent[2] = ent[3]
    d[base + 12 + 16 : base + 12 + 24] = ent[2].view(np.uint8)
    variants.append(d)
    got = G.decompress(variants, [a.nbytes] * len(variants), go)
    for v, gq in zip(variants, got):
        lz = O.decompress(v, a.nbytes, oo)
        if isinstance(lz, int):     # damage that breaks the skippable frame's own framing: both refuse
            assert isinstance(gq, int), (gq, lz)
        else:
            assert not isinstance(gq, int), hex(gq)
            assert gq.tobytes() == a.tobytes()


def _blocks(body, start):
    """[(offset, block type, literals type or None, size)] of the frame's blocks from `start` on"""
    out, pos = [], start
    while True:
        h = int(body[pos]) | int(body[pos + 1]) << 8 | int(body[pos + 2]) << 16
        bt, size = (h >> 1) & 3, h >> 3
        out.append((pos, bt, int(body[pos + 3]) & 3 if bt == 2 else None, size))
        pos += 3 + (1 if bt == 1 else size)
        if h & 1:
            return out, pos


def test_shared_tables_one_tree_for_the_data_bytes():
    """The data bytes of a large read are packed with ONE table (zstd_encode.hip, SpanRegion): their first span's block carries
    the tree description, every later span is one treeless block, and the index says so (bit 31 of the span count).  The
    decoder gives every such span the block with the tree and decodes the frame span by span -- no second launch
    (vbz_gpu_decode_span_paths) --; libzstd reads the frame; VBZ_HIP_SHARED_TABLES=0 writes round 4's frames (a tree per span), which
    decode the same way."""
    c = G.codec()
    a = O.synth_signal(5, 21, 300_000)
    o = (True, 2, 1, 1)
    go, oo = _lib.CompressionOptions(*o), O.options(*o)
    g = G.compress([a], go)[0]
    body, tr = _trailers(g)
    idx = tr[0]
    word = int(idx[8:12].view("<u4")[0])
    ns = word & 0x7FFFFFFF
    assert word >> 31 == 1 and len(idx) == 16 + 8 * ns
    starts = [int(idx[12 + 8 * j : 16 + 8 * j].view("<u4")[0]) for j in range(ns)]
    content = [int(idx[16 + 8 * j : 20 + 8 * j].view("<u4")[0]) for j in range(ns)]
    blocks, end = _blocks(body, starts[0])
    assert end == len(body)
    K = (a.size + 3) // 4                                   # the control bytes come first
    first_data = content.index(K)
    at = {b[0]: b for b in blocks}
    assert at[starts[first_data]][2] == 2                   # Compressed_Literals_Block: the tree
    later = [at[starts[j]] for j in range(first_data + 1, ns)]
    assert len(later) >= 30 and all(b[1] == 2 and b[2] == 3 for b in later), "treeless blocks behind the span with the tree"
    assert all(content[j + 1] - content[j] <= 8192 for j in range(first_data, ns - 1))
    assert O.decompress(g, a.nbytes, oo).tobytes() == a.tobytes()
    assert G.decompress([g], [a.nbytes], go)[0].tobytes() == a.tobytes()
    assert c.decode_span_paths() == (1, 1), "decoded span by span"
    # round 4's frames: the same entry points read them
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, oracle_lib as O, gpu_util as G\nfrom vbz_compression_amd import _lib\n"
            "a = O.synth_signal(5, 21, 300000)\ng = G.compress([a], _lib.CompressionOptions(True, 2, 1, 1))[0]\n"
            "sys.stdout.buffer.write(g.tobytes())" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    old = np.frombuffer(subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VBZ_HIP_SHARED_TABLES="0"), capture_output=True, check=True).stdout, np.uint8)
    body_o, tr_o = _trailers(old)
    assert int(tr_o[0][8:12].view("<u4")[0]) >> 31 == 0
    assert abs(len(old) - len(g)) < 0.004 * len(g), (len(old), len(g))     # the ratio is what it was
    assert G.decompress([old], [a.nbytes], go)[0].tobytes() == a.tobytes()
    assert c.decode_span_paths() == (1, 1)


def test_inherited_trees_are_verified_not_trusted():
    """What a treeless span decodes with is the tree of an EARLIER block, which the decoder's plan picks from the index (the last span
    whose first block brings a tree) -- a claim like every other in the index.  Frames whose index or blocks make that claim false
    (the flag without treeless spans, no flag with them, a span boundary moved into the treeless run, the tree span's entry
    dropped) still decode to what libzstd says, or are refused where libzstd refuses."""
    a = O.synth_signal(5, 22, 260_000)
    o = (True, 2, 1, 1)
    go, oo = _lib.CompressionOptions(*o), O.options(*o)
    g = G.compress([a], go)[0]
    body, tr = _trailers(g)
    idx = tr[0]
    base = len(g) - len(idx)
    ns = int(idx[8:12].view("<u4")[0]) & 0x7FFFFFFF
    content = [int(idx[16 + 8 * j : 20 + 8 * j].view("<u4")[0]) for j in range(ns)]
    first_data = content.index((a.size + 3) // 4)
    variants = []
    d = g.copy()                                            # no flag: the treeless spans fail, the frame goes to the second launch
    d[base + 8 : base + 12] = np.array([ns], "<u4").view(np.uint8)
    variants.append(d)
    for drop in (first_data, first_data + 1, first_data + 5, ns - 1):     # an index without one of its entries
        ent = np.delete(idx[12 : 12 + 8 * ns].view("<u4").reshape(ns, 2), drop, axis=0)
        tb = 16 + 8 * (ns - 1)
        t = np.concatenate([np.array([IDX_MAGIC, tb - 8, (ns - 1) | 0x80000000], "<u4"), ent.reshape(-1), np.array([tb], "<u4")]).view(np.uint8)
        variants.append(np.concatenate([g[:base], t]))
    d = g.copy()                                            # the tree span's block made treeless: no tree anywhere for the data bytes
    starts = [int(idx[12 + 8 * j : 16 + 8 * j].view("<u4")[0]) for j in range(ns)]
    d[starts[first_data] + 3] |= 1
    variants.append(d)
    d = g.copy()                                            # a later treeless block claims a tree of its own (garbage for a description)
    d[starts[first_data + 3] + 3] &= 0xFE
    variants.append(d)
    got = G.decompress(variants, [a.nbytes] * len(variants), go)
    for k, (v, gq) in enumerate(zip(variants, got)):
        lz = O.decompress(v, a.nbytes, oo)
        if isinstance(lz, int):
            assert isinstance(gq, int), (k, gq if isinstance(gq, int) else "data", lz)
        else:
            assert not isinstance(gq, int), (k, hex(gq))
            assert gq.tobytes() == lz.tobytes() == a.tobytes(), k


def test_large_reads_in_one_batch_with_errors():
    """Several large reads of different kinds of trouble in one batch: an empty read, a size that is not a multiple of the
    integer size, a destination that is too small, a truncated frame -- each gets the oracle's verdict, the others are unaffected."""
    c = G.codec()
    o = (True, 2, 1, 1)
    go, oo = _lib.CompressionOptions(*o), O.options(*o)
    reads = [O.synth_signal(5, 10, 700000), np.zeros(0, np.int16), O.synth_signal(5, 11, 1200001), O.synth_signal(5, 12, 524288)]
    odd = np.frombuffer(O.synth_signal(5, 13, 600000).tobytes()[:-1], np.uint8)
    got = G.compress(reads + [odd], go)
    for a, g in zip(reads, got):
        assert not isinstance(g, int)
        assert O.decompress(g, a.nbytes, oo).tobytes() == a.tobytes()
    assert got[-1] == O.compress(odd, oo)      # VBZ_INPUT_SIZE_ERROR
    frames = [got[0], got[1], got[2][: len(got[2]) // 2], got[3], got[0]]
    sizes = [reads[0].nbytes, 0, reads[2].nbytes, reads[3].nbytes - 2, reads[0].nbytes + 2]
    back = G.decompress(frames, sizes, go)
    for f, n, b in zip(frames, sizes, back):
        want = O.decompress(f, n, oo)
        if isinstance(want, int):
            assert isinstance(b, int) and (b == want or (b in (0xFFFFFFFB, 0xFFFFFFFF) and want in (0xFFFFFFFC, 0xFFFFFFFF, 0xFFFFFFFB))), (hex(b), hex(want))
        else:
            assert not isinstance(b, int) and b.tobytes() == want.tobytes()


def test_whole_suite_on_the_large_read_path():
    """VBZ_HIP_SEGMENTED=1 sends EVERY batch -- also the thousands of small reads of the other tests -- through the segmented
    kernels: the bit-exact svb tests, the known answers, the corruption tests and the fuzz corpus must not notice."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VBZ_HIP_SEGMENTED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_fuzz_corpus.py"), "-k",
                        "not bench_line and not fast5 and not h5repack and not hdf5 and not cpp_caller and not many_threads"],
                       capture_output=True, text=True, timeout=3000, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("kind,count,o", [("int16", 800_000_000, (True, 2, 1, 1)), ("int16", 1_000_000_000, (True, 2, 1, 1)), ("uint32", 1_000_000_000, (False, 4, 3, 0))])
def test_one_read_near_the_size_type_limit(kind, count, o):
    """vbz_size_t is 32 bits: the largest read the interface can describe is the one whose worst-case size still fits it.  int16 reads of
    800 M and 1 000 M samples (1.6 / 2.0 GB; the destination slot of the second, vbz_max_compressed_size, is 4.27 GB -- 28 MB below
    2^32 -- and its svb stream 2.5 GB: byte positions beyond 2^31) and one uint32 buffer of 1 000 M values (4.0 GB of input), generated
    on the device: the round trip is exact on the device, and the reference's decoder (oracle + libzstd) reads the frame -- spans,
    index trailer and all."""
    c = G.codec()
    dev = c.device
    opts = c.options(*o)
    size = o[1]
    nbytes = count * size
    with torch.cuda.stream(c.stream):
        off = torch.zeros(1, dtype=torch.int64, device=dev)
        lens = torch.tensor([count], dtype=torch.int32, device=dev)
        raw = torch.zeros(nbytes + 64, dtype=torch.uint8, device=dev)
        (c.synth_u32 if kind == "uint32" else c.synth_signal)(5, 3, raw, off, lens)
        cap = c.L.vbz_max_compressed_size(nbytes, ctypes.byref(opts))
        assert not _lib.is_error(cap) and cap > nbytes
        comp = torch.zeros(cap + 64, dtype=torch.uint8, device=dev)
        back = torch.zeros_like(raw)
        size32 = torch.tensor([nbytes], dtype=torch.int64).to(torch.int32).to(dev)
        cap32 = torch.tensor([cap], dtype=torch.int64).to(torch.int32).to(dev)
        csize = torch.zeros(1, dtype=torch.int32, device=dev)
        res = torch.zeros(1, dtype=torch.int32, device=dev)
        c.compress(raw, off, size32, comp, off, cap32, csize, opts)
        c.decompress(comp, off, csize, back, off, size32, res, opts)
    torch.cuda.synchronize()
    cs, rs = int(csize[0]) & 0xFFFFFFFF, int(res[0]) & 0xFFFFFFFF
    assert rs == nbytes and torch.equal(raw, back), (hex(cs), hex(rs))
    assert 2.2 < nbytes / cs < 2.7
    frame = comp[:cs].cpu().numpy()
    host = raw[:nbytes].cpu().numpy()
    del raw, comp, back
    torch.cuda.empty_cache()
    want = O.decompress(frame, nbytes, O.options(*o))
    assert not isinstance(want, int) and want.view(np.uint8).tobytes() == host.tobytes()
    del want
    if kind == "int16" and count == 1_000_000_000:
        # the same read through the single-buffer C API of vbz.h, host memory in and out: the bytes the batched entry point wrote
        L = c.L
        go = _lib.CompressionOptions(*o)
        out = np.empty(cap, np.uint8)
        n = L.vbz_compress(host.ctypes.data, nbytes, out.ctypes.data, cap, ctypes.byref(go))
        assert n == cs and out[:n].tobytes() == frame.tobytes()
        hback = np.empty(nbytes, np.uint8)
        m = L.vbz_decompress(out.ctypes.data, n, hback.ctypes.data, nbytes, ctypes.byref(go))
        assert m == nbytes and hback.tobytes() == host.tobytes()


def test_segment_tables_either_side_of_the_self_prefix_bound():
    """Calls of up to 1024 segments (16 384 int16 samples each) add up the lengths in front of a segment inside the segment's workgroup,
    larger calls get scan launches: one read of ~1000 segments, one of ~1100, and two reads in one call that share a table of ~1000
    (the second read's sums start in the middle of the table) -- svb bytes identical to the oracle's, the round trip exact, and a
    stream cut short refused with the oracle's verdict on both sides of the bound."""
    rng = np.random.default_rng(77)
    o = (True, 2, 0, 0)   # zig-zag + svb, no entropy stage: the bytes are the reference's
    go, oo = _lib.CompressionOptions(*o), O.options(*o)

    def signal(n):
        a = np.cumsum(rng.integers(-300, 300, n)).astype(np.int64)
        a[rng.integers(0, n, n // 50)] += rng.integers(-30000, 30000, n // 50)   # (two-byte deltas here and there)
        return a.astype(np.int16)

    for sizes in ([1000 * 16384 - 5], [1100 * 16384 + 7], [640 * 16384 + 11, 355 * 16384 - 3]):
        reads = [signal(n) for n in sizes]
        got = G.compress(reads, go)
        for a, g in zip(reads, got):
            want = O.compress(a, oo)
            assert not isinstance(g, int) and g.tobytes() == want.tobytes()
        back = G.decompress(got, [a.nbytes for a in reads], go)
        for a, b in zip(reads, back):
            assert not isinstance(b, int) and b.tobytes() == a.tobytes()
        cut = [g[: len(g) - 3] for g in got]
        back = G.decompress(cut, [a.nbytes for a in reads], go)
        for c, a, b in zip(cut, reads, back):
            want = O.decompress(c, a.nbytes, oo)
            assert isinstance(want, int) and isinstance(b, int) and b == want, (hex(b) if isinstance(b, int) else "decoded", hex(want))


def test_svb_suites_with_the_scan_launches():
    """The segmented svb kernels of a call with few segments add up the lengths in front of a segment by themselves; a call with
    more than 1024 segments (32 MB of samples) gets scan launches between the passes instead.  VBZ_HIP_SEG_SELF_MAX=0 gives every
    call the scan launches: the bit-exact svb tests, the known answers and the error verdicts must not notice (the default is what
    test_whole_suite_on_the_large_read_path runs)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VBZ_HIP_SEGMENTED="1", VBZ_HIP_SEG_SELF_MAX="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"), "-k",
                        "svb or c_abi_known_answers or c_abi_error or batch_ragged or unaligned_arena or descriptor_tables"],
                       capture_output=True, text=True, timeout=3000, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_whole_suite_on_the_one_wavefront_path():
    """The other way round: small batches -- most of what the other tests send -- take the large-read path by default (a call
    with a handful of reads cannot fill the device with one wavefront per read), so the kernels of the headline workload, one
    workgroup / one wavefront per read, get the same suites with VBZ_HIP_SEGMENTED=0."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VBZ_HIP_SEGMENTED="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_fuzz_corpus.py"), os.path.join(root, "tests", "test_gpu_repeats.py"), "-k",
                        "not bench_line and not fast5 and not h5repack and not hdf5 and not cpp_caller and not many_threads"],
                       capture_output=True, text=True, timeout=3000, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_trailers_are_an_option_of_the_context():
    """vbz_gpu_set_trailers(ctx, 0): every compressed buffer is ONE plain zstd frame (nothing behind it), with the same
    blocks -- run sequences included -- as with the trailers; both decode on the device and through libzstd."""
    c = G.codec()
    o = (True, 2, 1, 1)
    go, oo = _lib.CompressionOptions(*o), O.options(*o)
    reads = [O.synth_signal(5, 20, 100000), O.synth_signal(5, 21, 900000)]   # the one-wavefront path and the span path
    with_tr = [G.compress([a], go)[0] for a in reads]
    c.set_trailers(False)
    try:
        without = [G.compress([a], go)[0] for a in reads]
    finally:
        c.set_trailers(True)
    for a, w, p in zip(reads, with_tr, without):
        body, tr = _trailers(w)
        assert tr, "by default the buffer ends in skippable frames"
        assert _trailers(p)[1] == [] and p.tobytes() == body.tobytes()       # the same frame, nothing behind it
        assert O.zstd_content_size(p) == O.zstd_content_size(w)
        for f in (w, p):
            assert O.decompress(f, a.nbytes, oo).tobytes() == a.tobytes()
        back = G.decompress([w, p], [a.nbytes, a.nbytes], go)
        assert back[0].tobytes() == a.tobytes() and back[1].tobytes() == a.tobytes()


def test_buffer_that_ends_like_a_trailer_size():
    """The trailers are found from the END of the buffer (their last word is their size).  A plain frame whose last bytes
    happen to look like a size -- e.g. int16 samples stored without zig-zag whose last value is -8: the stream ends in
    f8 ff ff ff -- must not send the decoder anywhere (found by tools/soak.py: 0xFFFFFFF8 + 16 wrapped around)."""
    rng = np.random.default_rng(31)
    go, oo = _lib.CompressionOptions(False, 2, 1, 0), O.options(False, 2, 1, 0)
    reads = []
    for last in (-8, -16, -24, -1, 24, 32, 64, 200, 272, 280):
        for n in (40, 64, 200):
            a = rng.integers(-32768, 32767, n, endpoint=True).astype(np.int16)
            a[-1] = last
            a[-2] = -1 if last < 0 else 0
            reads.append(a)
    for sized in (False, True):
        frames = G.compress(reads, go, sized=sized)
        back = G.decompress(frames, [a.nbytes for a in reads], go, sized=sized)
        for a, f, b in zip(reads, frames, back):
            assert not isinstance(f, int) and not isinstance(b, int)
            assert b.tobytes() == a.tobytes()
            assert O.decompress(f, a.nbytes, oo, sized=sized).tobytes() == a.tobytes()
    # and arbitrary tails behind a valid frame body: whatever the last word says, the verdict is libzstd's
    base = G.compress([O.synth_signal(5, 40, 30000)], _lib.CompressionOptions(True, 2, 1, 1))[0]
    body, _ = _trailers(base)
    variants = []
    for tail in (0xFFFFFFF8, 0xFFFFFFF0, 0x18, 0x20, 0x110, len(body) - 8, len(body) + 8):
        junk = np.concatenate([rng.integers(0, 256, 40, dtype=np.uint8), np.array([tail & 0xFFFFFFFF], "<u4").view(np.uint8)])
        variants.append(np.concatenate([body, junk]))
    got = G.decompress(variants, [60000] * len(variants), _lib.CompressionOptions(True, 2, 1, 1))
    for v, gq in zip(variants, got):
        lz = O.decompress(v, 60000, O.options(True, 2, 1, 1))
        assert isinstance(gq, int) == isinstance(lz, int), (gq if isinstance(gq, int) else "data", lz if isinstance(lz, int) else "data")


def test_last_block_flag_inside_a_large_frame():
    """A frame that says "last block" in the middle ends there for every zstd decoder; the rest is trailing garbage and
    libzstd refuses the buffer.  The span decoder must not read on past it (found by tools/soak_corrupt.py: a span whose
    final block carried the flag still ended where the next span starts)."""
    a = O.synth_signal(5, 2, 1_000_000)
    o = (True, 2, 1, 1)
    go, oo = _lib.CompressionOptions(*o), O.options(*o)
    g = G.compress([a], go)[0]
    body, tr = _trailers(g)
    idx = [t for t in tr if int(t[:4].view("<u4")[0]) == IDX_MAGIC][0]
    ns = int(idx[8:12].view("<u4")[0]) & 0x7FFFFFFF   # (bit 31: spans that begin with a treeless block)
    starts = [int(idx[12 + 8 * j : 16 + 8 * j].view("<u4")[0]) for j in range(ns)]
    # walk the blocks of the frame (3-byte headers: last | type << 1 | size << 3; an RLE block holds one byte)
    pos, blocks = starts[0], []
    while True:
        h = int(body[pos]) | int(body[pos + 1]) << 8 | int(body[pos + 2]) << 16
        blocks.append(pos)
        pos += 3 + (1 if (h >> 1) & 3 == 1 else h >> 3)
        if h & 1:
            break
    assert pos == len(body) and set(starts) <= set(blocks)
    variants = []
    for k in (1, ns // 2, ns - 1):                      # the final block of span k - 1 ...
        d = g.copy()
        d[blocks[blocks.index(starts[k]) - 1]] |= 1
        variants.append(d)
    d = g.copy()
    d[blocks[blocks.index(starts[ns // 2]) + 1]] |= 1   # ... and a block in the middle of a span
    variants.append(d)
    got = G.decompress(variants, [a.nbytes] * len(variants), go)
    for v, gq in zip(variants, got):
        assert isinstance(O.decompress(v, a.nbytes, oo), int)
        assert isinstance(gq, int), "the device decoded a frame that ends in the middle"


def test_aliased_sources_cannot_overflow_the_segment_tables():
    """vbz_gpu.h only asks that src_bytes covers every read: reads may ALIAS their source bytes, and then the sizes add up to
    more than the arena -- which is what the host sizes the segment tables (and grids) of the large-read path by.  Three
    reads over the same 2 MB buffer: every read either comes out right or is refused with OUT_OF_MEMORY (the first one must
    come out right); nothing is written behind the tables (round 2 did that)."""
    c = G.codec()
    dev = c.device
    a = O.synth_signal(5, 11, 1_048_576)                # 2 MB of int16
    raw = torch.from_numpy(np.frombuffer(a.tobytes(), np.uint8).copy()).to(dev)
    src = torch.zeros(raw.numel() + 64, dtype=torch.uint8, device=dev)
    src[: raw.numel()] = raw
    n = 3
    for level in (0, 1):
        go, oo = _lib.CompressionOptions(True, 2, level, 1), O.options(True, 2, level, 1)
        cap = _lib.load().vbz_max_compressed_size(a.nbytes, ctypes.byref(go))
        doff, dtotal = batch.layout([cap + 32] * n, 64)
        dst = torch.zeros(dtotal + 64, dtype=torch.uint8, device=dev)
        res = torch.full((n,), -8, dtype=torch.int32, device=dev)
        c.compress(src[: raw.numel()], torch.zeros(n, dtype=torch.int64, device=dev), torch.full((n,), a.nbytes, dtype=torch.int32, device=dev), dst,
                   doff.to(dev), torch.full((n,), cap, dtype=torch.int32, device=dev), res, go)
        torch.cuda.synchronize()
        r = [int(x) & 0xFFFFFFFF for x in res.cpu().tolist()]
        host = dst.cpu().numpy()
        good = 0
        for i in range(n):
            if r[i] == 0xFFFFFFF9:
                continue
            assert r[i] < 0xFFFFFFF0, hex(r[i])
            f = host[int(doff[i]) : int(doff[i]) + r[i]]
            assert O.decompress(f, a.nbytes, oo).tobytes() == a.tobytes(), (level, i)
            good += 1
        assert good >= 1 and r[0] < 0xFFFFFFF0, [hex(x) for x in r]
        # the other direction: three reads over ONE compressed buffer decode into three slots (the destination arena covers
        # them, so nothing is refused there)
        f = O.compress(a, oo)
        fsrc = torch.zeros(len(f) + 64, dtype=torch.uint8, device=dev)
        fsrc[: len(f)] = torch.from_numpy(f.copy()).to(dev)
        ooff, ototal = batch.layout([a.nbytes + 32] * n, 64)
        out = torch.zeros(ototal + 64, dtype=torch.uint8, device=dev)
        res = torch.full((n,), -8, dtype=torch.int32, device=dev)
        c.decompress(fsrc[: len(f)], torch.zeros(n, dtype=torch.int64, device=dev), torch.full((n,), len(f), dtype=torch.int32, device=dev), out,
                     ooff.to(dev), torch.full((n,), a.nbytes, dtype=torch.int32, device=dev), res, go)
        torch.cuda.synchronize()
        host = out.cpu().numpy()
        for i in range(n):
            assert int(res[i]) == a.nbytes
            assert host[int(ooff[i]) : int(ooff[i]) + a.nbytes].tobytes() == a.tobytes()


def test_long_reads_among_many_short_ones_are_routed():
    """The reference treats every buffer alike (vbz/vbz.cpp:116-208); ultra-long nanopore reads exist (the shipped file already
    spans 9 885 ... 505 057 samples).  A batch of 4096 reads of ~100 k samples with three reads of 5 M samples among them: the
    batch's AVERAGE says "one workgroup / one wavefront per read", which for a 5 M-sample read means tens of milliseconds on
    one wavefront -- the library routes such reads to the large-read path on the device.  Bit-exact both ways (the frames of
    the long reads carry a span index and decode through the reference's decoder), the other reads' frames are byte for byte
    what they are without the long reads, and the batch takes < 1.3 x the time of the same batch without them."""
    import time

    c = G.codec()
    dev = c.device
    L = _lib.load()
    go, oo = _lib.CompressionOptions(True, 2, 1, 1), O.options(True, 2, 1, 1)
    n_short, long_at = 4096, (7, 2000, 4098)
    short = [O.synth_signal(5, i % 64, 90000 + 311 * (i % 64)) for i in range(64)]
    longs = [O.synth_signal(5, 100 + k, 5_000_000) for k in range(3)]

    def build(with_long):
        reads = [short[i % 64] for i in range(n_short)]
        if with_long:
            for pos, a in zip(long_at, longs):
                reads.insert(pos, a)
        sizes = [a.nbytes for a in reads]
        off, total = batch.layout(sizes, 64)
        raw = torch.zeros(total, dtype=torch.uint8, device=dev)
        uniq = {}
        for a, o_ in zip(reads, off.tolist()):
            if id(a) not in uniq:
                uniq[id(a)] = torch.from_numpy(np.frombuffer(a.tobytes(), np.uint8).copy()).to(dev)
            raw[o_ : o_ + a.nbytes] = uniq[id(a)]
        caps = [L.vbz_max_compressed_size(sz, ctypes.byref(go)) for sz in sizes]
        coff, ctotal = batch.layout(caps, 64)
        return dict(reads=reads, raw=raw, off=off.to(dev), size=torch.tensor(sizes, dtype=torch.int32, device=dev), coff=coff.to(dev), coff_h=coff,
                    cap=torch.tensor(caps, dtype=torch.int32, device=dev), comp=torch.zeros(ctotal, dtype=torch.uint8, device=dev),
                    csize=torch.zeros(len(reads), dtype=torch.int32, device=dev), back=torch.zeros(total, dtype=torch.uint8, device=dev),
                    res=torch.zeros(len(reads), dtype=torch.int32, device=dev))

    def run(B, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            c.compress(B["raw"], B["off"], B["size"], B["comp"], B["coff"], B["cap"], B["csize"], go)
            c.decompress(B["comp"], B["coff"], B["csize"], B["back"], B["off"], B["size"], B["res"], go)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    A, B = build(False), build(True)
    run(A, 1)
    run(B, 1)
    assert bool((B["res"] == B["size"]).all()) and torch.equal(B["raw"], B["back"])
    assert bool((A["res"] == A["size"]).all()) and torch.equal(A["raw"], A["back"])
    comp = B["comp"].cpu().numpy()
    cs = B["csize"].cpu().numpy()
    compA, csA = A["comp"].cpu().numpy(), A["csize"].cpu().numpy()
    for pos in long_at:   # the reference's decoder reads the long reads' frames; they came down the large-read path (span index)
        f = comp[int(B["coff_h"][pos]) : int(B["coff_h"][pos]) + int(cs[pos])]
        assert O.decompress(f, longs[0].nbytes, oo).tobytes() == B["reads"][pos].tobytes()
        body, trailers = _trailers(f)
        assert any(int(t[:4].view("<u4")[0]) == IDX_MAGIC for t in trailers)
    ia = 0
    for ib in range(len(B["reads"])):   # everybody else: the same bytes as without the long reads
        if ib in long_at:
            continue
        if ia % 97 == 0:
            fa = compA[int(A["coff_h"][ia]) : int(A["coff_h"][ia]) + int(csA[ia])]
            fb = comp[int(B["coff_h"][ib]) : int(B["coff_h"][ib]) + int(cs[ib])]
            assert fa.tobytes() == fb.tobytes(), (ia, ib)
        ia += 1
    tA = min(run(A, 3) for _ in range(2))
    tB = min(run(B, 3) for _ in range(2))
    raw_ratio = float(B["size"].to(torch.int64).sum()) / float(A["size"].to(torch.int64).sum())
    print("4096 short reads: %.2f ms; with three 5 M-sample reads (%.3f x the bytes): %.2f ms = %.2f x" % (tA * 1e3, raw_ratio, tB * 1e3, tB / tA))
    assert tB < 1.3 * tA, (tA, tB)


def test_aliased_sources_beyond_the_scratch_arena():
    """The same contract on the one-workgroup-per-read path: 3000 descriptors over ONE 180 KB source.  The library sizes its
    scratch arena by src_bytes, so with the entropy stage on only the first few reads get a slot; the slot plan runs on several workgroups (one per 1024
    reads, each adding up the slots in front of its own while the others write their verdicts) and must give every read
    the same verdict on every run: coded right, or refused with OUT_OF_MEMORY -- the first read always fits."""
    dev = torch.device("cuda", 0)
    a = O.synth_signal(5, 12, 90_000)
    raw = torch.from_numpy(np.frombuffer(a.tobytes(), np.uint8).copy()).to(dev)
    n = 3000
    for level in (0, 1):
        go, oo = _lib.CompressionOptions(True, 2, level, 1), O.options(True, 2, level, 1)
        cap = _lib.load().vbz_max_compressed_size(a.nbytes, ctypes.byref(go))
        doff, dtotal = batch.layout([cap + 32] * n, 64)
        runs = []
        for attempt in range(2):
            c = batch.GpuCodec(0)   # a fresh context: its scratch arena is as small as this batch asks for
            with torch.cuda.stream(c.stream):
                src = torch.zeros(raw.numel() + 64, dtype=torch.uint8, device=dev)
                src[: raw.numel()] = raw
                dst = torch.zeros(dtotal + 64, dtype=torch.uint8, device=dev)
                res = torch.full((n,), -8, dtype=torch.int32, device=dev)
                c.compress(src[: raw.numel()], torch.zeros(n, dtype=torch.int64, device=dev), torch.full((n,), a.nbytes, dtype=torch.int32, device=dev),
                           dst, doff.to(dev), torch.full((n,), cap, dtype=torch.int32, device=dev), res, go)
            c.synchronize()
            r = [int(x) & 0xFFFFFFFF for x in res.cpu().tolist()]
            host = dst.cpu().numpy()
            good = [i for i in range(n) if r[i] != 0xFFFFFFF9]
            # (without the entropy stage the svb stream goes straight to the destination: no scratch, every read fits)
            assert good and good[0] == 0 and (len(good) < n if level else len(good) == n), (len(good), hex(r[0]))
            assert good == list(range(len(good))), "the reads that fit are the first ones"
            want = O.compress(a, oo) if level == 0 else None
            for i in good:
                assert r[i] < 0xFFFFFFF0, hex(r[i])
                f = host[int(doff[i]) : int(doff[i]) + r[i]]
                if want is not None:
                    assert f.tobytes() == want.tobytes()
                else:
                    assert O.decompress(f, a.nbytes, oo).tobytes() == a.tobytes(), (level, i)
            runs.append(r)
            c.close() if hasattr(c, "close") else None
        assert runs[0] == runs[1], "the slot plan gave different verdicts on two runs"


def test_small_batches_take_the_large_read_path():
    """The shape rule's second clause: a call whose reads cannot fill the device with one wavefront each -- one read of 100 k
    samples, say: the single-buffer API, the HDF5 filter -- is coded as spans (its frame carries the span index), a batch of
    thousands of such reads is not (one wavefront per read: checkpoints only), and reads of less than 32 k samples stay on
    one wavefront however few they are.  Either way the reference's decoder reads the frames and the device reads them
    back; and a read that cycles a template gets its matches on both paths."""
    go, oo = _lib.CompressionOptions(True, 2, 1, 1), O.options(True, 2, 1, 1)

    def kinds(frames):
        return [sorted({int(t[:4].view("<u4")[0]) for t in _trailers(f)[1]}) for f in frames]

    one = [O.synth_signal(5, 40, 100_000)]
    few = [O.synth_signal(5, 41 + i, 70_000 + 1000 * i) for i in range(5)]
    short = [O.synth_signal(5, 50 + i, 20_000) for i in range(4)]
    many = [O.synth_signal(5, 60 + (i % 16), 60_000) for i in range(1200)]   # 144 MB: beyond what counts as a small batch
    cyc = [np.resize(O.synth_signal(5, 9, 15643), 120_000).astype(np.int16)]
    for reads, want_idx in ((one, True), (few, True), (short, False), (many, False), (cyc, False)):
        comp = G.compress(reads, go)
        k = kinds(comp[:8])
        assert all((IDX_MAGIC in x) == want_idx for x in k), (len(reads), k)
        back = G.decompress(comp, [a.nbytes for a in reads], go)
        for a, f, b in list(zip(reads, comp, back))[:: max(1, len(reads) // 40)]:
            assert not isinstance(b, int) and b.tobytes() == a.tobytes()
            assert O.decompress(f, a.nbytes, oo).tobytes() == a.tobytes()
    # the cycled read went to the one-wavefront matcher (no span index) and compresses like libzstd's
    f = G.compress(cyc, go)[0]
    assert len(f) <= 1.25 * len(O.compress(cyc[0], oo)), (len(f), len(O.compress(cyc[0], oo)))
