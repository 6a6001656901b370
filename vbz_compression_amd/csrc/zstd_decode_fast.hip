// zstd_decode_fast.hip -- the batched decoder for the frames zstd_encode.hip writes (and any frame of the same shape).
//
// zstd_decode.hip decodes ANY conforming frame with one wavefront per frame; on the device's own frames it spends a third of its
// time in the serial parts of a frame (block headers found one after the other, two tree descriptions, the sequence chain) as
// wave-uniform scalar code with 63 lanes idle, and the rest in a stream phase that sits on the memory system's ceiling for mixed
// scattered 64-byte accesses (profiles/r04_stream_scaling.md).  Here the same work is cut by what bounds it:
//
//   fast_scan_kernel     one LANE per frame: frame header, trailer, every block header; checks the shape (below) and leaves one
//                        task per Huffman stream: where it starts, how long it is, where its bytes go, which table it uses
//   fast_weights_kernel  one LANE per Huffman tree description: the FSE-coded weights (RFC 8878 4.2.1.1) -> 256 weight bytes
//   fast_streams_kernel  one wavefront per frame, nothing but the streams: 64 lanes, one stream each, whole aligned 128-byte lines
//                        requested per lane (a 64-dword ring per lane in LDS), 128 bytes stored per lane and burst
//   fast_runs_kernel     one wavefront per frame: the zero-run block (state chains from the encoder's checkpoints, placement from
//                        prefix sums: zstd_runs.h)
//
// The shape: a single frame with a content size and no dictionary, compressed blocks only, every literals section four Huffman
// streams, at most two trees (one of at most 9-bit codes, none above 11), no sequences except zero-run sequences (predefined LL / ML
// tables, offsets = repeat offset 1) in the FIRST block, at most 64 streams.  Nothing is trusted: a frame that is not of this shape,
// or that fails any check on the way (a stream that does not end on its first bit, a chain that does not regenerate what the later
// blocks leave of the content size ...), is marked in redo[] and decoded by zstd_decode_kernel in a last launch gated by that
// array -- which is also what decides every error verdict.  A frame decoded here gets exactly the bytes the one-wavefront decoder
// writes: same tables, same streams, same placement code.
// Replaces, like zstd_decode.hip, the reference's ZSTD_decompress call (vbz/vbz.cpp:236-273).
#include <algorithm>
#include <cstdlib>

#include "vbz_kernels.h"
#include "zstd_runs.h"

namespace vbzhip {

namespace {

constexpr uint32_t FAST_TASKS = 64;
constexpr uint32_t CP_MAGIC = 0x184D2A5Bu, IDX_MAGIC = 0x184D2A5Cu;
constexpr uint32_t TASK_REL = 1u << 31, TASK_TAB = 1u << 30, TASK_CNT = TASK_TAB - 1u;

struct FastFrame  // 128 bytes per read
{
    uint32_t fcs, ntask, ntree, block_max;
    uint32_t tree_off[2], tree_len[2];
    uint32_t tlog[2], nw[2];          // fast_weights_kernel
    uint32_t seq_off, seq_len, nseq;  // the zero-run block's bit stream (behind count, modes and the RLE symbol); nseq 0: none
    uint32_t b0_regen, base_out;      // its literals; the bytes it regenerates = where the later blocks' content starts
    uint32_t ws_lit, ws_pairs;        // staging in the destination slot (literals, length pairs)
    uint32_t cp_off, cp_count, cp_spacing;
    uint32_t pad[10];
};
static_assert(sizeof(FastFrame) == 128, "FastFrame");

struct FastTask
{
    uint32_t src, size;  // the stream in the read's source
    uint32_t out;        // where its bytes go: offset in the destination slot (TASK_REL: behind the zero-run block)
    uint32_t cnt;        // symbols | TASK_TAB (second tree) | TASK_REL
};

typedef __attribute__((address_space(1))) const uint8_t gcu8;
typedef __attribute__((address_space(1))) uint8_t gu8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ld32(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// one lane: the frame header (RFC 8878 3.1.1.1) of a frame of at least 16 bytes, in the terms of zstd_decode_kernel; false: not for the
// batched kernels (F->fcs, F->block_max, *pos = where the first block begins)
__device__ __forceinline__ bool scan_frame_header(const uint8_t* src, uint32_t cap, FastFrame* F, uint32_t* pos_out, uint32_t* has_checksum)
{
    const uint64_t h0 = ld64(src), h1 = ld64(src + 8);
    auto hb = [&](uint32_t i) -> uint32_t { return (uint32_t)((i < 8 ? h0 >> (8 * i) : h1 >> (8 * (i - 8))) & 0xFF); };
    const uint32_t fhd = hb(4);
    if ((uint32_t)h0 != 0xFD2FB528u || (fhd & 0x08) || (fhd & 3)) return false;  // (a Dictionary_ID field: the careful decoder)
    const uint32_t single = (fhd >> 5) & 1, fcs_flag = fhd >> 6;
    uint32_t pos = 5;
    uint64_t window = 0;
    if (!single) {
        const uint32_t wd = hb(pos++);
        const uint32_t wlog = 10 + (wd >> 3);
        if (wlog > 31) return false;
        window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7);
    }
    const uint32_t fsz = fcs_flag == 0 ? (single ? 1u : 0u) : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
    if (fsz == 0) return false;
    uint64_t fcs = 0;
    for (uint32_t i = 0; i < fsz; ++i) fcs |= (uint64_t)hb(pos + i) << (8 * i);
    if (fsz == 2) fcs += 256;
    pos += fsz;
    if (fcs > cap || fcs >= (1u << 30)) return false;
    if (single) window = fcs;
    F->fcs = (uint32_t)fcs;
    F->block_max = (uint32_t)(window < BLOCK_MAX ? window : BLOCK_MAX);
    *has_checksum = (fhd >> 2) & 1;
    *pos_out = pos;
    return true;
}

// ---- one lane per frame ------------------------------------------------------------------------------------------------------------
// Everything zstd_decode_kernel checks on its way through a frame of this shape is checked here (or in the kernels behind), in the
// same terms; whatever is not of this shape returns early and leaves redo[r] = 1.  (Reads of up to 16 bytes past the read are inside
// the arena's slack, as in zstd_decode_kernel.)
__global__ __launch_bounds__(256) void fast_scan_kernel(ReadBatch b, FastFrame* frames, FastTask* tasks, uint32_t* redo)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= b.n_reads) return;
    redo[r] = 1;
    if (b.gate && b.gate[r] >= GATE_SKIP) return;
    const uint32_t n = b.src_size[r];
    if (n >= E_FIRST || n < 32) return;
    const uint8_t* src = b.src + b.src_off[r];
    const uint32_t cap = b.dst_cap[r];
    FastFrame F = {};
    uint32_t pos;
    uint32_t has_checksum;
    if (!scan_frame_header(src, cap, &F, &pos, &has_checksum)) return;
    // the encoder's checkpoint trailer (see zero_run_chain_segments); an index trailer may follow it
    if (n >= 64) {
        uint32_t tb = ld32(src + n - 4), ne = n;
        if (tb >= 24 && tb <= n - 16 && (tb & 7u) == 0) {
            if (ld32(src + n - tb) == IDX_MAGIC && ld32(src + n - tb + 4) == tb - 8) {
                ne = n - tb;
                tb = ld32(src + ne - 4);
            }
        }
        if (tb >= 20 && tb <= 8 + 4 + 4 * 63 + 4 && tb + 16 <= ne) {
            const uint32_t m0 = ld32(src + ne - tb), m1 = ld32(src + ne - tb + 4), m2 = ld32(src + ne - tb + 8);
            const uint32_t cnt = m2 >> 16;
            if (m0 == CP_MAGIC && m1 == tb - 8 && tb == 16 + 4 * cnt && cnt >= 1) {
                F.cp_off = ne - tb + 12;
                F.cp_count = cnt;
                F.cp_spacing = m2 & 0xFFFFu;
            }
        }
    }
    FastTask* T = tasks + (size_t)r * FAST_TASKS;
    uint32_t ntask = 0, ntree = 0;
    uint64_t rel = 0;  // content of the blocks behind the zero-run block (or of all blocks)
    int cur_tab = -1;
    for (;;) {
        if (pos + 3 > n) return;
        const uint64_t a0 = ld64(src + pos), a1 = ld64(src + pos + 8);
        auto win = [&](uint32_t o) -> uint64_t { return o == 0 ? a0 : ((a0 >> (8 * o)) | (a1 << (64 - 8 * o))); };  // 8 bytes from offset o <= 8
        const uint32_t bh = (uint32_t)a0 & 0xFFFFFFu;
        const uint32_t last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype != 2 || bsize < 5 || bsize >= BLOCK_MAX || (uint64_t)pos + 3 + bsize > n) return;
        const uint32_t blk = pos + 3;
        uint32_t lh, regen, csize;
        uint32_t ltype;
        {
            const uint64_t v = win(3);
            const uint32_t h0 = (uint32_t)v & 0xFF, fmt = (h0 >> 2) & 3;
            ltype = h0 & 3;
            if (ltype < 2 || fmt == 0) return;
            if (fmt == 1) { lh = 3; regen = (uint32_t)(v >> 4) & 0x3FF; csize = (uint32_t)(v >> 14) & 0x3FF; }
            else if (fmt == 2) { lh = 4; regen = (uint32_t)(v >> 4) & 0x3FFF; csize = (uint32_t)(v >> 18) & 0x3FFF; }
            else { lh = 5; regen = (uint32_t)(v >> 4) & 0x3FFFF; csize = (uint32_t)(v >> 22) & 0x3FFFF; }
        }
        if (regen == 0 || csize == 0 || regen > BLOCK_MAX || lh + csize >= bsize) return;
        uint32_t tree_used = 0;
        if (ltype == 2) {
            if (ntree == 2) return;
            const uint32_t hb = src[blk + lh];
            if (hb >= 128) tree_used = 1 + ((hb - 127) + 1) / 2;
            else if (hb == 0) return;
            else tree_used = 1 + hb;
            if (tree_used > csize) return;
            F.tree_off[ntree] = blk + lh;
            F.tree_len[ntree] = tree_used;
            cur_tab = (int)ntree++;
        } else if (cur_tab < 0) {
            return;
        }
        uint32_t q = blk + lh + tree_used, qn = csize - tree_used;
        if (qn < 10) return;
        const uint64_t j = ltype == 3 ? win(3 + lh) : ld64(src + q);
        const uint32_t s1 = (uint32_t)j & 0xFFFFu, s2 = (uint32_t)(j >> 16) & 0xFFFFu, s3 = (uint32_t)(j >> 32) & 0xFFFFu;
        q += 6;
        qn -= 6;
        if (s1 + s2 + s3 > qn) return;
        const uint32_t seg = (regen + 3) >> 2;
        if (seg * 3 > regen) return;
        const uint32_t sq = blk + lh + csize, sqn = bsize - (lh + csize);
        const uint64_t sq8 = ld64(src + sq);  // (bytes past the block are never looked at)
        const uint32_t nseq0 = (uint32_t)sq8 & 0xFF;
        uint32_t out0, flags = cur_tab ? TASK_TAB : 0u;
        if (nseq0 == 0) {
            if (sqn != 1 || regen > F.block_max) return;
            out0 = (uint32_t)rel;
            flags |= TASK_REL;
            rel += regen;
            if (rel > F.fcs) return;
        } else {
            // zero-run sequences: the first block only, predefined LL / ML tables, OF = RLE of code 0 (repeat offset 1 is 1 there)
            if (ntask != 0 || sqn < 4) return;
            const uint32_t used0 = nseq0 < 128 ? 1u : (nseq0 < 255 ? 2u : 3u);
            const uint32_t b1 = (uint32_t)(sq8 >> 8) & 0xFF, b2 = (uint32_t)(sq8 >> 16) & 0xFF;
            const uint32_t ns0 = nseq0 < 128 ? nseq0 : (nseq0 < 255 ? ((nseq0 - 128) << 8) + b1 : b1 + (b2 << 8) + 0x7F00);
            if (used0 + 2 >= sqn || ns0 == 0) return;
            if (((uint32_t)(sq8 >> (8 * used0)) & 0xFF) != 0x10u || ((uint32_t)(sq8 >> (8 * (used0 + 1))) & 0xFF) != 0u) return;
            const uint64_t ws_lit64 = ((uint64_t)F.fcs + 15u) & ~15ull;
            if (ws_lit64 + BLOCK_MAX + 16 >= 0xFFFFFFF0ull) return;
            F.ws_lit = (uint32_t)ws_lit64;
            F.ws_pairs = F.ws_lit + ((regen + 7u) & ~7u);
            if ((uint64_t)F.ws_pairs + 8ull * ns0 + 8 > cap) return;
            F.seq_off = sq + used0 + 2;
            F.seq_len = sqn - (used0 + 2);
            F.nseq = ns0;
            F.b0_regen = regen;
            out0 = F.ws_lit;
        }
        if (ntask + 4 > FAST_TASKS) return;
        {
            const uint32_t so[4] = { 0u, s1, s1 + s2, s1 + s2 + s3 };
            const uint32_t sz[4] = { s1, s2, s3, qn - s1 - s2 - s3 };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                FastTask t;
                t.src = q + so[k];
                t.size = sz[k];
                t.out = out0 + (uint32_t)k * seg;
                t.cnt = (k < 3 ? seg : regen - 3 * seg) | flags;
                T[ntask + k] = t;
            }
        }
        ntask += 4;
        pos = blk + bsize;
        if (last) break;
    }
    if (has_checksum) {
        if (pos + 4 > n) return;
        pos += 4;  // xxh64 of the content: not verified (like zstd_decode_kernel)
    }
    while (n - pos >= 8) {  // skippable frames behind the frame
        const uint32_t m0 = ld32(src + pos), m1 = ld32(src + pos + 4);
        if ((m0 & 0xFFFFFFF0u) != 0x184D2A50u || (uint64_t)pos + 8 + m1 > n) break;
        pos += 8 + m1;
    }
    if (pos != n) return;
    if (F.nseq) {
        F.base_out = F.fcs - (uint32_t)rel;  // what the zero-run block must regenerate (fast_runs_kernel holds it to that)
        if (F.base_out > F.block_max) return;
    } else if (rel != F.fcs) {
        return;
    }
    F.ntask = ntask;
    F.ntree = ntree;
    frames[r] = F;
    redo[r] = 0;
}

// ---- one lane per tree description ---------------------------------------------------------------------------------------------------
// The serial reader of zstd_decode.hip (huf_read_weights: read_ncount, fse_build, two interleaved FSE states over a backward bit
// stream, the checks on the weights) with one description per LANE: what it indexes lives in LDS columns of its own ([index][lane]:
// no bank conflicts), the weights go to memory.  Descriptions it cannot hold (a
// 12-bit code) are left to the careful decoder like every failure.
constexpr int WMAXS = 11;   // the weights' alphabet: 0 .. HUF_TABLELOG_MAX - 1 (libzstd >= 1.4.7 refuses a description that lists more)
struct WeightsLds
{
    uint32_t desc[34][WAVE];  // the description behind its header byte, zero beyond
    uint32_t tab[64][WAVE];   // FSE decoding table of the weights: symbol | nbBits << 8 | base << 16
    int16_t norm[WMAXS + 1][WAVE];
    uint16_t symnext[WMAXS + 1][WAVE];
};

// (nrec: frames[] / weights / redo[] hold nrec records, record i for read i % n_reads -- one per read, or one per block of the frames
// the reference wrote: ref_lit_scan_kernel)
__global__ __launch_bounds__(WAVE) void fast_weights_kernel(ReadBatch b, FastFrame* frames, uint8_t* weights, uint32_t* redo, uint32_t nrec)
{
    __shared__ WeightsLds S;
    const int lane = threadIdx.x;
    const uint32_t t = blockIdx.x * (uint32_t)WAVE + (uint32_t)lane;
    const uint32_t r = t >> 1, k = t & 1;
    if (r >= nrec || redo[r]) return;
    FastFrame* F = frames + r;
    if (k >= F->ntree) return;
    const uint8_t* g = b.src + b.src_off[r % b.n_reads] + F->tree_off[k];
    const uint32_t used = F->tree_len[k];
    uint8_t* W = weights + ((size_t)r * 2 + k) * 256;
    const uint32_t hb = g[0];
    uint32_t nw = 0;
#define WFAIL()       \
    do {              \
        redo[r] = 1;  \
        return;       \
    } while (0)
    // (the checks on the weights -- 4.2.1 -- are made as they are produced: reading them back took one memory round trip per weight,
    // 0.18 ms per launch whatever the number of trees, because every load stood behind the branch on the one before)
    uint32_t total = 0;
    int r1 = 0;
    bool wide = false;
    auto put = [&](uint32_t wt) {
        W[nw++] = (uint8_t)wt;
        wide |= wt >= 12u;
        total += (wt != 0u && wt < 12u) ? (1u << (wt - 1u)) : 0u;
        r1 += (wt == 1u);
    };
    if (hb >= 128) {  // direct representation: 4 bits per weight
        const uint32_t cnt = hb - 127;
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t by = g[1 + i / 2];
            put((i & 1) ? (by & 0xF) : (by >> 4));
        }
    } else {
        for (uint32_t j = 0; j < 34; ++j) {
            const uint32_t off = 1 + 4 * j;
            uint32_t v = 0;
            if (off < used) {
                v = ld32(g + off);
                if (off + 4 > used) v &= (1u << (8 * (used - off))) - 1u;
            }
            S.desc[j][lane] = v;
        }
        auto bits = [&](uint32_t bitpos, uint32_t nb) -> uint32_t {  // nb <= 16 bits at bit position bitpos of the description
            const uint32_t idx = bitpos >> 5;
            const uint64_t v = (uint64_t)S.desc[idx < 33 ? idx : 33][lane] | ((uint64_t)S.desc[idx < 32 ? idx + 1 : 33][lane] << 32);
            return (uint32_t)(v >> (bitpos & 31)) & ((1u << nb) - 1u);
        };
        // ---- probabilities (RFC 8878 4.1.1): read_ncount(p + 1, hb, 255, 6) of zstd_decode.hip
        const int log = (int)bits(0, 4) + 5;
        if (log > 6) WFAIL();
        uint32_t bitpos = 4;
        int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
        bool prev0 = false;
        while (remaining > 1 && sym <= WMAXS) {
            if (bitpos > 8u * 128u) WFAIL();
            if (prev0) {
                for (;;) {
                    const uint32_t rr = bits(bitpos, 2);
                    bitpos += 2;
                    for (uint32_t i = 0; i < rr; ++i) {
                        if (sym > WMAXS) WFAIL();
                        S.norm[sym++][lane] = 0;
                    }
                    if (rr != 3) break;
                    if (bitpos > 8u * 128u) WFAIL();
                }
                prev0 = false;
                if (sym > WMAXS) break;
                continue;
            }
            const int max = (2 * threshold - 1) - remaining;
            const uint32_t v = bits(bitpos, (uint32_t)nbits);
            int count;
            if ((int)(v & (uint32_t)(threshold - 1)) < max) {
                count = (int)(v & (uint32_t)(threshold - 1));
                bitpos += (uint32_t)(nbits - 1);
            } else {
                count = (int)(v & (uint32_t)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                bitpos += (uint32_t)nbits;
            }
            count--;
            remaining -= count < 0 ? -count : count;
            S.norm[sym++][lane] = (int16_t)count;
            prev0 = (count == 0);
            while (remaining < threshold) {
                nbits--;
                threshold >>= 1;
            }
        }
        if (remaining != 1 || sym > WMAXS + 1) WFAIL();
        const uint32_t hdr = (bitpos + 7) >> 3;
        if (hdr > hb) WFAIL();
        const int nsym = sym;
        // ---- decoding table: fse_build
        const int size = 1 << log;
        {
            int high = size - 1;
            for (int s = 0; s < nsym; ++s) {
                const int c = S.norm[s][lane];
                if (c == -1) {
                    S.tab[high--][lane] = (uint32_t)s;
                    S.symnext[s][lane] = 1;
                } else {
                    S.symnext[s][lane] = (uint16_t)c;
                }
            }
            const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
            int p = 0;
            for (int s = 0; s < nsym; ++s) {
                const int c = S.norm[s][lane];
                for (int i = 0; i < c; ++i) {
                    S.tab[p][lane] = (uint32_t)s;
                    do {
                        p = (p + step) & mask;
                    } while (p > high);
                }
            }
            if (p != 0) WFAIL();
            for (int u = 0; u < size; ++u) {
                const uint32_t s = S.tab[u][lane] & 0xFF;
                const uint32_t ns = S.symnext[s][lane];
                S.symnext[s][lane] = (uint16_t)(ns + 1);
                const int nb = log - hbit(ns);
                S.tab[u][lane] = s | ((uint32_t)nb << 8) | ((((ns << nb) - (uint32_t)size) & 0xFFFFu) << 16);
            }
        }
        // ---- two interleaved states over the backward bit stream behind the probabilities
        const int qn = (int)hb - (int)hdr;
        auto qbyte = [&](int i) -> uint32_t {
            const uint32_t o = hdr + (uint32_t)i;
            return (S.desc[o >> 2][lane] >> (8 * (o & 3))) & 0xFF;
        };
        if (qn < 1 || qbyte(qn - 1) == 0) WFAIL();
        const int top = hbit(qbyte(qn - 1));
        int left = (qn - 1) * 8 + top;
        uint64_t buf = top ? ((uint64_t)(qbyte(qn - 1) & ((1u << top) - 1u)) << (64 - top)) : 0ull;
        int avail = top, nextb = qn - 1;
        auto rd = [&](int nb) -> uint32_t {
            while (avail <= 56 && nextb > 0) {
                --nextb;
                buf |= (uint64_t)qbyte(nextb) << (56 - avail);
                avail += 8;
            }
            const uint32_t v = nb ? (uint32_t)(buf >> (64 - nb)) : 0u;
            buf <<= nb;
            avail = avail > nb ? avail - nb : 0;
            left -= nb;
            return v;
        };
        uint32_t s1 = rd(log), s2 = rd(log);
        if (left < 0) WFAIL();
        for (;;) {
            if (nw > 253) WFAIL();
            uint32_t e = S.tab[s1][lane];
            put(e & 0xFFu);
            s1 = (e >> 16) + rd((int)((e >> 8) & 0xFF));
            if (left < 0) {
                put(S.tab[s2][lane] & 0xFFu);
                break;
            }
            if (nw > 253) WFAIL();
            e = S.tab[s2][lane];
            put(e & 0xFFu);
            s2 = (e >> 16) + rd((int)((e >> 8) & 0xFF));
            if (left < 0) {
                put(S.tab[s1][lane] & 0xFFu);
                break;
            }
        }
    }
    // ---- the weights must describe a complete code (4.2.1): the last weight follows from the others
    if (wide || total == 0) WFAIL();
    const int tlog = hbit(total) + 1;
    if (tlog > 11) WFAIL();  // (12-bit codes are legal: the careful decoder has the table for them)
    const uint32_t rest = (1u << tlog) - total;
    if (rest & (rest - 1)) WFAIL();
    const uint32_t lastw = (uint32_t)hbit(rest) + 1;
    W[nw++] = (uint8_t)lastw;
    r1 += (lastw == 1);
    if (r1 < 2 || (r1 & 1)) WFAIL();
    F->tlog[k] = (uint32_t)tlog;
    F->nw[k] = nw;
#undef WFAIL
}

// ---- one wavefront per frame: the streams ---------------------------------------------------------------------------------------------
// Table of a tree from its weight bytes: like huf_fill_table of zstd_decode.hip (cells by increasing weight, then symbol value; starts
// from ballots), the cells per weight counted with ballots too.
__device__ __forceinline__ void fast_fill_table(uint16_t* T, const uint8_t* W, uint32_t nw, uint32_t tlog, int lane)
{
    uint32_t wt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (uint32_t)lane + 64u * j;
        wt[j] = s < nw ? W[s] : 0u;
    }
    uint32_t base[13];
    {
        uint32_t acc = 0;
#pragma unroll
        for (int v = 1; v <= 12; ++v) {
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) c += (uint32_t)__popcll(__ballot(wt[j] == (uint32_t)v));
            base[v] = acc;
            acc += c << (v - 1);
        }
        base[0] = 0;
    }
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t st[4], len[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st[j] = 0;
        len[j] = wt[j] ? 1u << (wt[j] - 1) : 0u;
#pragma unroll
        for (int v = 1; v <= 12; ++v) {
            const uint64_t m = __ballot(wt[j] == (uint32_t)v);
            if (wt[j] == (uint32_t)v) st[j] = base[v] + ((uint32_t)__popcll(m & below) << (v - 1));
            base[v] += (uint32_t)__popcll(m) << (v - 1);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (uint32_t)lane + 64u * j;
        const uint16_t ent = (uint16_t)(s | ((tlog + 1 - wt[j]) << 8));
        if (len[j] && len[j] < 64)
            for (uint32_t i = 0; i < len[j]; ++i) T[st[j] + i] = ent;
        uint64_t big = __ballot(len[j] >= 64);
        while (big) {
            const int src_lane = __ffsll((long long)big) - 1;
            big &= big - 1;
            const uint32_t bst = (uint32_t)__builtin_amdgcn_readlane((int)st[j], src_lane);
            const uint32_t blen = (uint32_t)__builtin_amdgcn_readlane((int)len[j], src_lane);
            const uint32_t bent = (uint32_t)__builtin_amdgcn_readlane((int)ent, src_lane);
            for (uint32_t i = lane; i < blen; i += WAVE) T[bst + i] = (uint16_t)bent;
        }
    }
}

// Round 6: a 32-dword ring, 64-byte requests, a look at the ring's room every 16 symbols -- 13.5 KB of LDS per wavefront instead of 21.8:
// eleven wavefronts per CU instead of seven.  The kernel is a chain of dependent LDS look-ups (two per symbol pair, ~220 cycles per symbol
// and lane at seven wavefronts, VALU 25 - 44 % busy, LDS pipe 35 %): wavefronts in flight are what its throughput follows.  Looking at the
// room twice as often costs 0.85 ms per 65 536 frames by itself (VBZ_FS_RING=64 VBZ_FS_BATCH=32 VBZ_FS_PERIOD=16: 7.6 against 6.75 ms for the
// whole stage), the four wavefronts more return it and a little more: 6.76 ms alone, and 608 against 597 GB/s for the step with the two
// halves of a call in flight (round 5's 48-dword ring paid the first and got two wavefronts: it lost).
#ifndef VBZ_FS_RING
#define VBZ_FS_RING 32
#define VBZ_FS_BATCH 16
#define VBZ_FS_PERIOD 16
#endif
constexpr int RING = VBZ_FS_RING;      // dwords per lane in the LDS ring
constexpr int BATCH = VBZ_FS_BATCH;    // one request: a whole aligned line of 4 * BATCH bytes (128)
constexpr int PERIOD = VBZ_FS_PERIOD;  // symbols between two requests (at most 11 bits each: PERIOD * 11 / 32 dwords)
constexpr int BURST = 128;        // bytes a lane stores together
constexpr unsigned long long LINE = 4ull * BATCH;
// The ring cannot run dry between a request and its commit (one period later): a request is refused while more than RING - BATCH dwords
// are unread, so a period starts with at least RING - BATCH + 1 - (a period's worst case) unread dwords, which must cover a period
static_assert((RING & (RING - 1)) == 0 && RING - BATCH + 1 - (PERIOD * 11 + 31) / 32 >= (PERIOD * 11 + 31) / 32, "the ring would run dry");
static_assert(BURST % PERIOD == 0 && PERIOD % 16 == 0 && BATCH % 4 == 0, "burst = whole periods, period = whole groups of 16 symbols");
constexpr uint32_t TBL_BIG = 2048, TBL_SMALL = 512;  // entries: one table of up to 11-bit codes, one of up to 9-bit codes

// The decoder of zstd_decode.hip's flush_tasks_ring (no bit buffer: the next 32 unread bits are one v_alignbit of two ring dwords held
// in registers, two symbols per 32 fresh bits, the ring upside down with a mirror slot) around a different memory side:
//  * a lane requests whole aligned 128-byte LINES, from the one that holds the stream's last byte down to the one that holds its
//    first: never an address outside the lines the stream itself touches -- which may begin up to 127 bytes below the arena's first
//    input byte and end up to 127 bytes behind its last one: vbz_gpu.h asks callers for memory that is readable to those line
//    boundaries (any arena that is its own allocation is) --, and a line is fetched once.  Bit positions count from the top of the first line; the bytes between the stream's end and that top are consumed
//    before the first symbol;
//  * request and commit of a batch are one stretch of straight-line code (request, decode a period, commit): the wait in front of
//    the commit is a counted vmcnt, not the vmcnt(0) a loop-carried batch costs;
//  * 128 decoded bytes per lane leave together.
__global__ __launch_bounds__(WAVE) void fast_streams_kernel(ReadBatch b, const FastFrame* frames, const FastTask* tasks, const uint8_t* weights, uint32_t* redo)
{
    __shared__ __attribute__((aligned(16))) uint16_t T[TBL_BIG + TBL_SMALL];
    __shared__ uint32_t ringbuf[RING + 1][WAVE];
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x;
    if (redo[r]) return;
    const FastFrame* F = frames + r;
    const uint32_t ntree = F->ntree, ntask = F->ntask;
    uint32_t tlog0 = F->tlog[0], tlog1 = ntree > 1 ? F->tlog[1] : 0u;
    uint32_t tb0 = 0, tb1 = TBL_BIG;
    if (ntree > 1 && tlog1 > tlog0) {  // the wider table takes the large slot
        tb0 = TBL_BIG;
        tb1 = 0;
    }
    {
        const uint32_t big = tlog0 > tlog1 ? tlog0 : tlog1, small = tlog0 > tlog1 ? tlog1 : tlog0;
        if (big > 11 || (ntree > 1 && small > 9)) {
            if (lane == 0) redo[r] = 1;
            return;
        }
    }
    fast_fill_table(T + tb0, weights + ((size_t)r * 2) * 256, F->nw[0], tlog0, lane);
    if (ntree > 1) fast_fill_table(T + tb1, weights + ((size_t)r * 2 + 1) * 256, F->nw[1], tlog1, lane);
    const uint8_t* src = b.src + b.src_off[r];
    uint8_t* dst = b.dst + b.dst_off[r];
    const bool mine = (uint32_t)lane < ntask;
    FastTask t = { 0, 0, 0, 0 };
    if (mine) t = tasks[(size_t)r * FAST_TASKS + lane];
    uint32_t cnt = t.cnt & TASK_CNT;
    const bool second = (t.cnt & TASK_TAB) != 0;
    const uint32_t sL = 32u - (second ? tlog1 : tlog0);
    const uint16_t* Tl = T + (second ? tb1 : tb0);
    gu8* o = (gu8*)dst + t.out + ((t.cnt & TASK_REL) ? F->base_out : 0u);
    const uint32_t nbytes = t.size;
    uint32_t* ring = &ringbuf[0][0] + lane;
    wave_lds_sync();

    bool bad = false;
    int32_t n = -1;
    uint32_t widx = 0, end_bits = 0;
    uint64_t nextline = 0, lowline = 0;  // the next request reads [nextline - 128, nextline); no line below lowline is read
    if (mine) {
        const uint8_t* p = src + t.src;
        const uint32_t last = nbytes ? p[nbytes - 1] : 0u;
        if (last == 0) {
            bad = true;
            cnt = 0;
        } else {
            const uint64_t e = (uint64_t)(p + nbytes);
            nextline = (e + (LINE - 1ull)) & ~(LINE - 1ull);
            lowline = (uint64_t)p & ~(LINE - 1ull);
            const uint32_t pad = (uint32_t)(nextline - e);
            n = -(int32_t)(8u * pad + 8u - (uint32_t)(31 - __clz((int)last)));  // bytes above the stream, padding bits, end mark
            end_bits = 8u * (nbytes + pad);
        }
    }
    typedef __attribute__((address_space(1))) const u32x4 gq4;
#define FETCH(pend)                                                         \
    do {                                                                    \
        gcu8* q__ = (gcu8*)(nextline - LINE);                               \
        _Pragma("unroll") for (int v = 0; v < BATCH / 4; ++v) {             \
            const u32x4 x__ = *(gq4*)(q__ + 16 * (BATCH / 4 - 1 - v));      \
            pend[4 * v + 0] = x__.w;                                        \
            pend[4 * v + 1] = x__.z;                                        \
            pend[4 * v + 2] = x__.y;                                        \
            pend[4 * v + 3] = x__.x;                                        \
        }                                                                   \
        nextline -= LINE;                                                   \
    } while (0)
#define RING_PUT(pend)                                                                                 \
    do {                                                                                               \
        const uint32_t wb__ = (uint32_t)RING - (widx & (uint32_t)(RING - 1));                          \
        _Pragma("unroll") for (int k = 0; k < BATCH; ++k) ring[(wb__ - (uint32_t)k) * WAVE] = pend[k]; \
        if (wb__ == (uint32_t)RING) ring[0] = pend[0];                                                 \
        widx += BATCH;                                                                                 \
    } while (0)
#define MORE() (nextline > lowline)
    for (int f = 0; f < 2; ++f) {
        uint32_t pend0[BATCH];
        if (cnt > 0 && MORE()) {
            FETCH(pend0);
            RING_PUT(pend0);
        }
    }
    int32_t tprev = n >> 5;
    uint32_t w0 = ring[((((uint32_t)tprev) & (uint32_t)(RING - 1)) + 1u) * WAVE];
    uint32_t w1 = ring[(((uint32_t)tprev) & (uint32_t)(RING - 1)) * WAVE];
    uint32_t w2 = ring[(((uint32_t)tprev - 1u) & (uint32_t)(RING - 1)) * WAVE];
#define HUF_PAIR(e1, e2)                                                     \
    do {                                                                     \
        const int32_t t__ = n >> 5;                                          \
        const bool adv__ = t__ != tprev;                                     \
        const uint32_t a__ = adv__ ? w1 : w0, b__ = adv__ ? w2 : w1;         \
        w0 = a__;                                                            \
        w1 = b__;                                                            \
        tprev = t__;                                                         \
        w2 = ring[(((uint32_t)t__ - 1u) & (uint32_t)(RING - 1)) * WAVE];     \
        uint32_t x__ = __builtin_amdgcn_alignbit(a__, b__, (uint32_t)n);     \
        e1 = Tl[x__ >> sL];                                                  \
        x__ <<= (e1 >> 8);                                                   \
        e2 = Tl[x__ >> sL];                                                  \
    } while (0)
#define QUAD(dstword)                                                                             \
    do {                                                                                          \
        uint32_t e1, e2, e3, e4;                                                                  \
        HUF_PAIR(e1, e2);                                                                         \
        n -= (int32_t)((e1 >> 8) + (e2 >> 8));                                                    \
        HUF_PAIR(e3, e4);                                                                         \
        n -= (int32_t)((e3 >> 8) + (e4 >> 8));                                                    \
        dstword = (e1 & 0xFFu) | ((e2 & 0xFFu) << 8) | ((e3 & 0xFFu) << 16) | (e4 << 24);         \
    } while (0)
    // room for a batch: at most RING - BATCH dwords of the ring are still unread
#define ROOM() (widx - (((uint32_t)~n) >> 5) <= (uint32_t)(RING - BATCH))
    typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
    // symbols one at a time / in groups of 16 (one 16-byte store), as many as the lane asks for
    auto singles = [&](uint32_t k) {
        while (__any(k > 0)) {
            uint32_t pend[BATCH];
            const bool issue = k > 0 && ROOM() && MORE();
            if (issue) FETCH(pend);
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                if (k > 0) {
                    uint32_t e1, e2;
                    HUF_PAIR(e1, e2);
                    (void)e2;  // only the first symbol of the pair is taken
                    n -= (int32_t)(e1 >> 8);
                    *o = (uint8_t)e1;
                    ++o;
                    --k;
                    --cnt;
                }
            }
            if (issue) RING_PUT(pend);
        }
    };
    auto groups = [&](uint32_t g) {
        while (__any(g > 0)) {
            uint32_t pend[BATCH];
            const bool issue = g > 0 && ROOM() && MORE();
            if (issue) FETCH(pend);
            if (g > 0) {
                uint32_t ow[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) QUAD(ow[q]);
                const u32x4 ov = { ow[0], ow[1], ow[2], ow[3] };
                *(gs4*)o = ov;
                o += 16;
                cnt -= 16;
                --g;
            }
            if (issue) RING_PUT(pend);
        }
    };
    // ---- head: up to the next 128-byte line of the output, so that every burst below writes whole aligned lines (the memory system
    // makes 1.45 x more of whole lines than of 64-byte pieces, and unaligned bursts cost 0.8 ms per 65 536 frames:
    // profiles/r04_stream_scaling.md)
    {
        const uint32_t to16 = (uint32_t)(-(int64_t)(uint64_t)o) & 15u;
        singles(cnt < to16 ? cnt : to16);
        const uint32_t to128 = ((uint32_t)(-(int64_t)(uint64_t)o) & 127u) >> 4;
        groups((cnt >> 4) < to128 ? (cnt >> 4) : to128);
    }
    // ---- whole bursts
    while (__any(cnt >= (uint32_t)BURST)) {
        if (cnt >= (uint32_t)BURST) {
            uint32_t ow[BURST / 4];
#pragma unroll
            for (int h = 0; h < BURST / PERIOD; ++h) {
                uint32_t pend[BATCH];
                const bool issue = ROOM() && MORE();
                if (issue) FETCH(pend);
#pragma unroll
                for (int q = h * PERIOD / 4; q < (h + 1) * PERIOD / 4; ++q) QUAD(ow[q]);
                if (h == BURST / PERIOD - 1) {  // the stores go out in front of the last commit: its wait is vmcnt(stores)
#pragma unroll
                    for (int q = 0; q < BURST / 16; ++q) {
                        const u32x4 ov = { ow[4 * q], ow[4 * q + 1], ow[4 * q + 2], ow[4 * q + 3] };
                        *(gs4*)(o + 16 * q) = ov;
                    }
                }
                if (issue) RING_PUT(pend);
            }
            o += BURST;
            cnt -= BURST;
        }
    }
    // ---- the rest of a stream
    groups(cnt >> 4);
    singles(cnt);
#undef HUF_PAIR
#undef QUAD
#undef ROOM
#undef MORE
#undef RING_PUT
#undef FETCH
    if (mine && !bad && n != -(int32_t)end_bits) bad = true;  // every bit of the stream must be consumed, none beyond
    if (__any(bad)) {
        if (lane == 0) redo[r] = 1;
    } else if (F->nseq == 0 && lane == 0) {
        b.result[r] = F->fcs;
    }
}

// ---- frames the reference wrote: the literals of the first block, beside the chain walk (RefLits, vbz_kernels.h) ---------------------
// libzstd gives a read one or two 128 KB blocks whose literals are FOUR Huffman streams of up to 32 KB: a chain of ~ 28 000 dependent
// symbols per stream, seven tenths of what such a frame costs the one-wavefront decoder -- and nothing of it needs the sequence chains
// that ref_chain_kernel walks meanwhile on a few hundred wavefronts.  Huffman codes resynchronise, so (as zstd_decode.hip's
// huf_split_plan) every stream is cut into 16 pieces of equal bit length, one lane each; but a piece is walked ONCE here:
//   run-up  a lane starts REF_RUNUP bits before its piece (dry) and walks to the first code boundary inside it: its presumed start;
//   walk    from there to the first boundary inside the next piece, and what it decodes is KEPT -- in a stripe of its own in the free
//           part of the destination slot, behind where the block's literals belong;
//   check   a piece's end must be the next one's presumed start, the last one's the first bit of the stream, and the counts must add
//           up to the stream's regenerated size;
//   hand-over  the counts say which literals a stripe holds; zstd_decode_kernel reads them where they stand.
// Nothing is repaired and no verdict is given here: a frame for which anything fails is left as it was -- the one-wavefront decoder
// decodes its streams itself and is the one to say what is wrong with it.
constexpr uint32_t REF_TASKS = 4;
constexpr uint32_t REF_PIECES_LOG = 4, REF_PIECES = 1u << REF_PIECES_LOG;   // pieces a stream, at most (8 or 4 for shorter streams: idle lanes)
constexpr uint32_t REF_PIECES_LOG_MIN = 2;
constexpr uint32_t REF_RUNUP = 768;                          // bits
constexpr uint32_t REF_HALO = 256;                           // literals of the next stripe repeated behind a stripe (= zstd_decode.hip's LANE_COPY_MAX)
constexpr uint32_t REF_MIN_PIECE = 384;                      // bytes: a piece of 384 bytes holds at least 256 symbols of up to 11 bits + the run-up's share
#ifndef VBZ_REF_RING
#define VBZ_REF_RING 32
#define VBZ_REF_BATCH 16
#endif
static_assert(REF_TASKS * REF_PIECES == (uint32_t)WAVE && REF_PIECES_LOG_MIN >= 2, "one lane per piece; the move goes by eight stripes");

// One lane per frame with only[r] != 0: which of its blocks are of the shape -- up to REF_UNITS of them, a record ("unit") each: where the
// tree (its own, or the one of an earlier unit for a treeless block) and the four streams are, how many pieces a stream, which part of the
// slot's free space takes its stripes.  skip[k * n_reads + r] = 0 for the units there are.  Whatever the walk through the block headers does
// not understand ends it: the units found so far stand, the rest of the frame is the decoder's.
constexpr uint32_t REF_UNITS_LOG = 2, REF_UNITS = 1u << REF_UNITS_LOG;   // (= REF_MAXBLK of the chain walk: it hands over at most four blocks)
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
static_assert(REF_UNITS == REF_MAXBLK, "the decoder looks a block's record up among REF_MAXBLK per read");
__global__ __launch_bounds__(256) void ref_lit_scan_kernel(ReadBatch b, const uint32_t* only, FastFrame* frames, FastTask* tasks, uint32_t* skip, uint32_t max_units)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= b.n_reads) return;
    // unit k of read r is record k * n_reads + r: the wavefronts that have work -- most frames have one or two units -- are neighbours in
    // the grid (with a read's units side by side three of four workgroups were empty, and workgroups go round the XCDs in order: two of the
    // eight got all the work, the kernel took 7 ms instead of 2.9)
    const uint32_t nr = b.n_reads;
    for (uint32_t k = 0; k < max_units; ++k) skip[(size_t)k * nr + r] = 1;
    if (!only[r]) return;
    if (b.gate && b.gate[r] >= GATE_SKIP) return;
    const uint32_t n = b.src_size[r];
    if (n >= E_FIRST || n < 32) return;
    const uint8_t* src = b.src + b.src_off[r];
    const uint32_t cap = b.dst_cap[r];
    FastFrame H = {};
    uint32_t pos, has_checksum;
    if (!scan_frame_header(src, cap, &H, &pos, &has_checksum)) return;
    // where the decoder stages the literals of a block whose chains are walked ahead (zstd_decode_kernel: ws_plit, par)
    const uint64_t at = ((uint64_t)H.fcs + 15u) & ~15ull;
    uint32_t nunit = 0, tree_unit = REF_NONE, regens[REF_UNITS] = { 0u, 0u, 0u, 0u };
    uint64_t sum_regen = 0;
    bool more_blocks = false;
    for (uint32_t bidx = 0; nunit < max_units; ++bidx) {
        if ((uint64_t)pos + 3 + 16 > n) break;
        const uint64_t a0 = ld64(src + pos), a1 = ld64(src + pos + 8);
        const uint32_t bh = (uint32_t)a0 & 0xFFFFFFu;
        const uint32_t last = bh & 1u, btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3) break;
        const uint64_t next = (uint64_t)pos + 3 + (btype == 1 ? 1u : bsize);
        if (next > n) break;
        if (bidx != 0 || !last) more_blocks = true;
        const uint32_t blk = pos + 3;
        bool unit = false;
        if (btype == 2 && bsize >= 5 && bsize < BLOCK_MAX) {
            FastFrame F = H;
            uint32_t lh = 0, regen = 0, csize = 0;
            const uint64_t v = (a0 >> 24) | (a1 << 40);
            const uint32_t h0 = (uint32_t)v & 0xFF, fmt = (h0 >> 2) & 3, ltype = h0 & 3;
            bool ok = ltype >= 2 && fmt != 0;    // compressed literals in four streams, under a tree of their own or the one before
            if (ok) {
                if (fmt == 1) { lh = 3; regen = (uint32_t)(v >> 4) & 0x3FF; csize = (uint32_t)(v >> 14) & 0x3FF; }
                else if (fmt == 2) { lh = 4; regen = (uint32_t)(v >> 4) & 0x3FFF; csize = (uint32_t)(v >> 18) & 0x3FFF; }
                else { lh = 5; regen = (uint32_t)(v >> 4) & 0x3FFFF; csize = (uint32_t)(v >> 22) & 0x3FFFF; }
                ok = regen != 0 && csize != 0 && regen <= BLOCK_MAX && lh + csize < bsize && regen <= H.fcs;
            }
            uint32_t tree_used = 0;
            if (ok && ltype == 2) {
                const uint32_t hb = src[blk + lh];
                if (hb >= 128) tree_used = 1 + ((hb - 127) + 1) / 2;
                else if (hb == 0) ok = false;
                else tree_used = 1 + hb;
            } else if (ok) {
                ok = tree_unit != REF_NONE;      // a treeless block under a tree this kernel has no record of
            }
            // (no sequences -- every block but the first of a long read, whose later blocks are data bytes only --: the literals ARE the
            // block's content, their place is the block's; the pieces' tail placement is all there is to do)
            const bool noseq = ok && src[blk + lh + csize] == 0;
            ok = ok && (!noseq || lh + csize + 1 == bsize);
            ok = ok && tree_used + 10 <= csize && at + ((regen + 15u) & ~15u) + 16 <= cap;
            if (ok) {
                uint32_t q = blk + lh + tree_used;
                const uint32_t qn = csize - tree_used - 6;
                const uint64_t j = ld64(src + q);
                const uint32_t s1 = (uint32_t)j & 0xFFFFu, s2 = (uint32_t)(j >> 16) & 0xFFFFu, s3 = (uint32_t)(j >> 32) & 0xFFFFu;
                q += 6;
                const uint32_t seg = (regen + 3) >> 2;
                if (s1 + s2 + s3 <= qn && seg * 3 <= regen) {
                    const uint32_t so[4] = { 0u, s1, s1 + s2, s1 + s2 + s3 };
                    const uint32_t sz[4] = { s1, s2, s3, qn - s1 - s2 - s3 };
                    // 16 pieces a stream where every stream has 16 x 384 bytes (blocks from ~ 30 000 samples on), else 8, else 4 (~ 8 000 samples)
                    const uint32_t szmin = min(min(sz[0], sz[1]), min(sz[2], sz[3]));
                    uint32_t gs = REF_PIECES_LOG;
                    while (gs > REF_PIECES_LOG_MIN && szmin < (REF_MIN_PIECE << gs)) --gs;
                    if (szmin >= (REF_MIN_PIECE << gs)) {
                        const uint32_t u = nunit * nr + r;
                        F.ntask = REF_TASKS;
                        F.ntree = ltype == 2 ? 1u : 0u;
                        F.tree_off[0] = blk + lh;
                        F.tree_len[0] = tree_used;
                        F.b0_regen = regen;
                        F.ws_lit = (uint32_t)at;
                        F.pad[0] = pos;      // the block header
                        F.pad[1] = csize;
                        F.pad[2] = last;
                        F.pad[3] = gs;
                        F.pad[4] = ltype == 2 ? u : tree_unit;   // whose weights
                        F.pad[7] = bidx;
                        F.pad[8] = noseq ? 1u : 0u;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            FastTask t;
                            t.src = q + so[k];
                            t.size = sz[k];
                            t.out = (uint32_t)at + (uint32_t)k * seg;
                            t.cnt = k < 3 ? seg : regen - 3 * seg;
                            tasks[(size_t)u * REF_TASKS + k] = t;
                        }
                        frames[u] = F;
                        regens[nunit] = regen;
                        sum_regen += regen;
                        if (ltype == 2) tree_unit = u;
                        ++nunit;
                        unit = true;
                    }
                }
            }
            // (a block that brings a tree but is no unit: later treeless blocks are under a tree without a record)
            if (!unit && ltype == 2) tree_unit = REF_NONE;
        }
        pos = (uint32_t)next;
        if (last) break;
    }
    if (nunit == 0) return;
    // the slot's free space behind the literals' staging place (a whole block's worth if the frame has more blocks than one: a block that
    // is not a unit has its literals staged there by the decoder while later units' stripes wait), shared out by the units' literals
    const uint64_t stage = more_blocks || nunit > 1 ? (uint64_t)BLOCK_MAX : (((uint64_t)regens[0] + 15u) & ~15ull);
    const uint64_t s0 = (at + stage + 16u + 127u) & ~127ull;
    if (s0 + 4096u > cap) return;
    const uint64_t room = cap - s0;
    uint64_t off = s0;
    for (uint32_t k = 0; k < nunit; ++k) {
        const uint64_t share = (room * regens[k] / sum_regen) & ~127ull;
        frames[(size_t)k * nr + r].pad[5] = (uint32_t)off;
        frames[(size_t)k * nr + r].pad[6] = (uint32_t)share;
        off += share;
        skip[(size_t)k * nr + r] = 0;
    }
}

// one wavefront per unit (block of a frame) that the scan and the weights kernel have passed: the table, the 64 pieces, the check, halo and tail
__global__ __launch_bounds__(WAVE, 3) void ref_pieces_kernel(ReadBatch b, const FastFrame* frames, const FastTask* tasks, const uint8_t* weights,
                                                            const uint32_t* skip, RefLits* lits, uint32_t* pos)
{
    // (The ring's size is this kernel's own.  It runs beside ref_chain_kernel, whose wavefront holds 60 KB of a CU's LDS: eight of these fit
    // beside it, twelve would with 16 dwords a lane -- but a 16-dword ring in requests of 16 bytes has no slack for the dword that is
    // requested a pair ahead (w2): measured, 5 % of the frames came out with wrong symbols of the RIGHT lengths, which no check of positions
    // and counts can see.  Hence the two dwords of margin in the assertion, and 32 / 16, fast_streams_kernel's proven pair.)
    constexpr int RING = VBZ_REF_RING, BATCH = VBZ_REF_BATCH, PERIOD = 16;
    constexpr unsigned long long LINE = 4ull * BATCH;
    static_assert((RING & (RING - 1)) == 0 && RING - BATCH + 1 - (PERIOD * 11 + 31) / 32 >= (PERIOD * 11 + 31) / 32 + 2, "the ring would run dry");
    static_assert(BURST % PERIOD == 0 && BATCH % 4 == 0, "burst = whole periods");
    __shared__ __attribute__((aligned(16))) uint16_t T[TBL_BIG];
    __shared__ uint32_t ringbuf[RING + 1][WAVE];
    const int lane = threadIdx.x;
    const uint32_t u = blockIdx.x, r = u % b.n_reads;   // (unit k of read r: k * n_reads + r)
    RefLits res = {};
#define LEAVE()                           \
    do {                                  \
        if (lane == 0) lits[u] = res;     \
        return;                           \
    } while (0)
    if (skip[u]) LEAVE();
    const FastFrame* F = frames + u;
    const uint32_t tu = F->pad[4];   // the unit whose tree this block is coded under (itself, or an earlier one of the frame)
    if (tu > u || tu % b.n_reads != r || skip[tu]) LEAVE();
    const uint32_t tlog = frames[tu].tlog[0];
    if (tlog > 11 || tlog == 0) LEAVE();
    fast_fill_table(T, weights + ((size_t)tu * 2) * 256, frames[tu].nw[0], tlog, lane);
    const uint8_t* src = b.src + b.src_off[r];
    uint8_t* dst = b.dst + b.dst_off[r];
    // G = 16, 8 or 4 pieces a stream (the scan's choice): lanes [0, 4 G) each walk a piece, the others idle along (their piece is empty)
    const uint32_t gs = F->pad[3], G = 1u << gs, nact = REF_TASKS << gs;
    if (gs < REF_PIECES_LOG_MIN || gs > REF_PIECES_LOG) LEAVE();
    const bool act = (uint32_t)lane < nact;
    const uint32_t st = act ? (uint32_t)lane >> gs : 0u, j = (uint32_t)lane & (G - 1u);
    const FastTask tk = tasks[(size_t)u * REF_TASKS + st];
    const uint32_t nbytes = tk.size, cnt = tk.cnt;
    const uint8_t* p = src + tk.src;
    const uint32_t last = p[nbytes - 1];
    if (__any(last == 0)) LEAVE();
    const uint32_t pad = 8u - (uint32_t)hbit(last), B = 8u * nbytes;
    const uint32_t seg = (B - pad + G - 1u) >> gs;
    const uint32_t c_lo = act ? pad + j * seg : pad, c_hi = act ? (j == G - 1u ? B : pad + (j + 1u) * seg) : pad;
    // the stripes: what the slot has behind the literals' place, 128-byte aligned, a 64th each; a piece holds its share of the
    // stream's symbols give or take a few per cent -- a stripe must have room for a quarter more, else the frame is not done here
    const uint32_t regen = F->b0_regen;
    const uint64_t tb = ((uint64_t)(dst + F->pad[5]) + 127ull) & ~127ull, te = (uint64_t)(dst + F->pad[5]) + F->pad[6];   // (the unit's share: the scan)
    const uint32_t pcap = te > tb ? (uint32_t)(((te - tb) >> (gs + 2u)) < 0x10000ull ? ((te - tb) >> (gs + 2u)) : 0x10000ull) & ~127u : 0u;
    if (__any((cnt >> gs) + (cnt >> (gs + 2u)) + 256u > pcap)) LEAVE();
    gu8* o = (gu8*)(tb + (uint64_t)lane * pcap);

    // ---- the walker: fast_streams_kernel's (aligned lines into a ring per lane, two symbols per 32 fresh bits), started in mid-stream:
    // bit positions count from the top of the first line a lane requests (rel = abs + off; abs: bits consumed from the stream's end)
    const uint32_t sL = 32u - tlog;
    const uint16_t* Tl = T;
    uint32_t* ring = &ringbuf[0][0] + lane;
    const uint32_t from = (!act || j == 0 || c_lo - pad <= REF_RUNUP) ? pad : c_lo - REF_RUNUP;
    const uint64_t e = (uint64_t)(p + nbytes);
    uint64_t nextline = (e - (uint64_t)(from >> 3) + (LINE - 1ull)) & ~(LINE - 1ull);
    const uint64_t lowline = (uint64_t)p & ~(LINE - 1ull);
    const int32_t off = (int32_t)(8ll * (int64_t)(nextline - e));
    int32_t n = -((int32_t)from + off);
    uint32_t widx = 0;
    wave_lds_sync();
    typedef __attribute__((address_space(1))) const u32x4 gq4;
#define FETCH(pend)                                                         \
    do {                                                                    \
        gcu8* q__ = (gcu8*)(nextline - LINE);                               \
        _Pragma("unroll") for (int v = 0; v < BATCH / 4; ++v) {             \
            const u32x4 x__ = *(gq4*)(q__ + 16 * (BATCH / 4 - 1 - v));      \
            pend[4 * v + 0] = x__.w;                                        \
            pend[4 * v + 1] = x__.z;                                        \
            pend[4 * v + 2] = x__.y;                                        \
            pend[4 * v + 3] = x__.x;                                        \
        }                                                                   \
        nextline -= LINE;                                                   \
    } while (0)
#define RING_PUT(pend)                                                                                 \
    do {                                                                                               \
        const uint32_t wb__ = (uint32_t)RING - (widx & (uint32_t)(RING - 1));                          \
        _Pragma("unroll") for (int k = 0; k < BATCH; ++k) ring[(wb__ - (uint32_t)k) * WAVE] = pend[k]; \
        if (wb__ == (uint32_t)RING) ring[0] = pend[0];                                                 \
        widx += BATCH;                                                                                 \
    } while (0)
#define MORE() (nextline > lowline)
    for (int f = 0; f < 2; ++f) {
        uint32_t pend0[BATCH];
        if (MORE()) {
            FETCH(pend0);
            RING_PUT(pend0);
        }
    }
    int32_t tprev = n >> 5;
    uint32_t w0 = ring[((((uint32_t)tprev) & (uint32_t)(RING - 1)) + 1u) * WAVE];
    uint32_t w1 = ring[(((uint32_t)tprev) & (uint32_t)(RING - 1)) * WAVE];
    uint32_t w2 = ring[(((uint32_t)tprev - 1u) & (uint32_t)(RING - 1)) * WAVE];
#define HUF_PAIR(e1, e2)                                                     \
    do {                                                                     \
        const int32_t t__ = n >> 5;                                          \
        const bool adv__ = t__ != tprev;                                     \
        const uint32_t a__ = adv__ ? w1 : w0, b__ = adv__ ? w2 : w1;         \
        w0 = a__;                                                            \
        w1 = b__;                                                            \
        tprev = t__;                                                         \
        w2 = ring[(((uint32_t)t__ - 1u) & (uint32_t)(RING - 1)) * WAVE];     \
        uint32_t x__ = __builtin_amdgcn_alignbit(a__, b__, (uint32_t)n);     \
        e1 = Tl[x__ >> sL];                                                  \
        x__ <<= (e1 >> 8);                                                   \
        e2 = Tl[x__ >> sL];                                                  \
    } while (0)
#define QUAD(dstword)                                                                             \
    do {                                                                                          \
        uint32_t e1, e2, e3, e4;                                                                  \
        HUF_PAIR(e1, e2);                                                                         \
        n -= (int32_t)((e1 >> 8) + (e2 >> 8));                                                    \
        HUF_PAIR(e3, e4);                                                                         \
        n -= (int32_t)((e3 >> 8) + (e4 >> 8));                                                    \
        dstword = (e1 & 0xFFu) | ((e2 & 0xFFu) << 8) | ((e3 & 0xFFu) << 16) | (e4 << 24);         \
    } while (0)
#define ROOM() (widx - (((uint32_t)~n) >> 5) <= (uint32_t)(RING - BATCH))
    typedef __attribute__((address_space(1), aligned(16))) u32x4 gs4;
    uint32_t m = 0;
    bool spill = false;
    // to the first code boundary at or beyond `stop` (rel).  A symbol is at most 11 bits: 128 / 16 symbols are decoded unlooked-at while
    // they cannot cross it, the last few one by one.  keep: the symbols are stored (a lane that runs out of room stops, and says so).
    auto walk_to = [&](int32_t stop, bool keep) {
        if (keep) {
            while (__any(!spill && stop + n >= (int32_t)(BURST * 11))) {
                if (!spill && stop + n >= (int32_t)(BURST * 11)) {
                    if (m + (uint32_t)BURST > pcap) {
                        spill = true;
                    } else {
                        uint32_t ow[BURST / 4];
#pragma unroll
                        for (int h = 0; h < BURST / PERIOD; ++h) {
                            uint32_t pend[BATCH];
                            const bool issue = ROOM() && MORE();
                            if (issue) FETCH(pend);
#pragma unroll
                            for (int q = h * PERIOD / 4; q < (h + 1) * PERIOD / 4; ++q) QUAD(ow[q]);
                            if (h == BURST / PERIOD - 1) {
#pragma unroll
                                for (int q = 0; q < BURST / 16; ++q) {
                                    const u32x4 ov = { ow[4 * q], ow[4 * q + 1], ow[4 * q + 2], ow[4 * q + 3] };
                                    *(gs4*)(o + 16 * q) = ov;
                                }
                            }
                            if (issue) RING_PUT(pend);
                        }
                        o += BURST;
                        m += BURST;
                    }
                }
            }
        }
        while (__any(!spill && stop + n >= 16 * 11)) {
            uint32_t pend[BATCH];
            const bool go = !spill && stop + n >= 16 * 11;
            const bool issue = go && ROOM() && MORE();
            if (issue) FETCH(pend);
            if (go) {
                uint32_t ow[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) QUAD(ow[q]);
                if (keep) {
                    if (m + 16u > pcap) {
                        spill = true;
                    } else {
                        const u32x4 ov = { ow[0], ow[1], ow[2], ow[3] };
                        *(gs4*)o = ov;
                        o += 16;
                        m += 16;
                    }
                }
            }
            if (issue) RING_PUT(pend);
        }
        while (__any(!spill && stop + n > 0)) {
            uint32_t pend[BATCH];
            const bool issue = !spill && stop + n > 0 && ROOM() && MORE();
            if (issue) FETCH(pend);
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                if (!spill && stop + n > 0) {
                    uint32_t e1, e2;
                    HUF_PAIR(e1, e2);
                    (void)e2;  // only the first symbol of the pair is taken
                    n -= (int32_t)(e1 >> 8);
                    if (keep) {
                        if (m + 1u > pcap) {
                            spill = true;
                        } else {
                            *o = (uint8_t)e1;
                            ++o;
                            ++m;
                        }
                    }
                }
            }
            if (issue) RING_PUT(pend);
        }
    };
    walk_to((int32_t)c_lo + off, false);           // run-up (lanes that start at a known boundary are there already)
    const uint32_t s_bit = (uint32_t)(-n - off);   // where this piece presumably starts
    walk_to((int32_t)c_hi + off, true);
    const uint32_t e_bit = (uint32_t)(-n - off);
#undef HUF_PAIR
#undef QUAD
#undef ROOM
#undef MORE
#undef RING_PUT
#undef FETCH
    if (__any(spill)) LEAVE();
    // ---- the pieces must chain from the end mark to the first bit of the stream, symbol for symbol
    const uint32_t e_prev = (uint32_t)__shfl_up((int)e_bit, 1, 64);
    const bool holds = !act || ((j == 0 ? s_bit == pad : s_bit == e_prev) && (j != G - 1u || e_bit == B));
    const uint32_t incl = wave_incl_scan_u32(m);
    const uint32_t before_ = (uint32_t)__shfl((int)incl, st ? (int)(st * G) - 1 : 0, 64);   // (every lane takes part: the source must be active)
    const uint32_t before = st ? before_ : 0u;
    const uint32_t total = (uint32_t)__shfl((int)incl, (int)(st * G + G - 1u), 64) - before;
    if (__any(!holds || (act && total != cnt))) LEAVE();
    // The stripes stay where they are: the decoder reads the literals out of them (RefLits: tb, pcap; pos[]: the index of every piece's first
    // literal) -- moving them to one place first cost 0.54 of this kernel's 2.1 ms per 16 384 frames and 3 GB of traffic under the walks.
    // What is moved is a HALO: behind its last literal a stripe gets the next stripe's first REF_HALO, so that a run of up to REF_HALO
    // literals that begins in a stripe is read in one piece from it (the decoder's lanes place runs of up to 256 bytes each; a run that
    // had to be fetched in two parts cost every lane of the wavefront a second memory round trip, trip after trip).
    if (__any(act && (m < REF_HALO || m + REF_HALO > pcap))) LEAVE();
    __syncthreads();   // (the stripes are in memory)
    if ((uint32_t)lane + 1u < nact) {
        typedef __attribute__((address_space(1), aligned(16))) const u32x4 gl4;
        typedef __attribute__((address_space(1), aligned(1))) u32x4 gh4;
        gcu8* f = (gcu8*)(tb + (uint64_t)(lane + 1) * pcap);
        gu8* h = (gu8*)(tb + (uint64_t)lane * pcap) + m;
        u32x4 v[REF_HALO / 16];
#pragma unroll
        for (int k = 0; k < (int)(REF_HALO / 16); ++k) v[k] = *(gl4*)(f + 16 * k);
#pragma unroll
        for (int k = 0; k < (int)(REF_HALO / 16); ++k) *(gh4*)(h + 16 * k) = v[k];
    }
    const uint32_t lit0 = tk.out - F->ws_lit + (incl - m - before);   // the index of this stripe's first literal
    pos[(size_t)u * WAVE + lane] = act ? lit0 : 0xFFFFFFFFu;   // (no stripe: behind every literal)
    // The literals behind a block's last sequence -- for a read they are most of its data bytes, nine tenths of all literals -- are the last
    // bytes the block regenerates.  If the block ends at E (the frame's last: E = fcs), literal x of them belongs at E - regen + x whatever the
    // sequences are: the stripes go there now, whole (what lands below the first such literal is overwritten by the decoder's output later, which
    // never reads it), and the decoder finds its longest copy done (RefLits.tail).  Piece after piece by the whole wavefront -- an
    // instruction is a kilobyte of whole lines --, eight pieces' loads in flight before their stores.
    // (A block that is not the frame's last: libzstd cuts its input into blocks of BLOCK_MAX bytes, so the k-th such block ends at k x
    // block_max -- a guess the decoder checks like everything else: it skips its copy only if the literals' place comes out where they were put.)
    const uint64_t full = ((uint64_t)F->pad[7] + 1u) * F->block_max;   // (the block's ordinal in the frame)
    const uint32_t tail_end = F->pad[2] ? F->fcs : (full < F->fcs ? (uint32_t)full : F->fcs);
    if (tail_end >= regen) {
        typedef __attribute__((address_space(1), aligned(16))) const u32x4 gl4;
        typedef __attribute__((address_space(1), aligned(1))) u32x4 gst4;
        const uint32_t dpos = tail_end - regen + lit0;
        const uint32_t k0 = 16u * (uint32_t)lane;
        constexpr int MG = 8;
        for (int g = 0; g < (int)nact; g += MG) {
            u32x4 v[MG][3];
            uint32_t tailb[MG];
#pragma unroll
            for (int u = 0; u < MG; ++u) {
                const uint32_t mq = (uint32_t)__builtin_amdgcn_readlane((int)m, g + u);
                gcu8* f = (gcu8*)(tb + (uint64_t)(g + u) * pcap);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (k0 + 1024u * c + 16u <= mq) v[u][c] = *(gl4*)(f + k0 + 1024u * c);
                const uint32_t rb = mq & 15u;
                tailb[u] = (uint32_t)lane < rb ? f[mq - rb + (uint32_t)lane] : 0u;
            }
#pragma unroll
            for (int u = 0; u < MG; ++u) {
                const uint32_t mq = (uint32_t)__builtin_amdgcn_readlane((int)m, g + u);
                gu8* d = (gu8*)dst + (uint32_t)__builtin_amdgcn_readlane((int)dpos, g + u);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (k0 + 1024u * c + 16u <= mq) *(gst4*)(d + k0 + 1024u * c) = v[u][c];
                const uint32_t rb = mq & 15u;
                if ((uint32_t)lane < rb) d[mq - rb + (uint32_t)lane] = (uint8_t)tailb[u];
                if (mq > 3072u + 16u) {   // (a stripe of more than 3 KB: only where the slot is far larger than the frame)
                    gcu8* f = (gcu8*)(tb + (uint64_t)(g + u) * pcap);
                    for (uint32_t k = k0 + 3072u; k + 16u <= mq; k += 1024u) *(gst4*)(d + k) = *(gl4*)(f + k);
                }
            }
        }
        res.tail = tail_end;
    } else if (F->pad[8]) {
        LEAVE();   // (a block without sequences whose literals could not be put in place: nothing to hand over)
    }
    res.blk = F->pad[0];
    res.regen = regen;
    res.csize = F->pad[1];
    res.at = F->ws_lit;
    res.tb = (uint32_t)(tb - (uint64_t)dst);
    res.pcap = pcap;
    LEAVE();
#undef LEAVE
}

// ---- one wavefront per frame: the zero-run block ----------------------------------------------------------------------------------------
constexpr uint32_t RUNS_LDS = 8704;
__global__ __launch_bounds__(WAVE) void fast_runs_kernel(ReadBatch b, const FastFrame* frames, const SeqDTables* dtabs, uint32_t* redo)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[RUNS_LDS / 4];
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x;
    if (redo[r]) return;
    const FastFrame* F = frames + r;
    const uint32_t nseq = F->nseq;
    if (nseq == 0) return;
    const uint8_t* src = b.src + b.src_off[r];
    uint8_t* dst = b.dst + b.dst_off[r];
    const uint2 llt = dtabs->ll[lane], mlt = dtabs->ml[lane];
    uint2* pairs = reinterpret_cast<uint2*>(dst + F->ws_pairs);
    const uint32_t regen = F->b0_regen, base_out = F->base_out;
    uint32_t total = 0, ok = 3;
    if (F->cp_count)
        ok = zero_run_chain_segments(src + F->seq_off, F->seq_len, pairs, nseq, llt, mlt, regen, src + F->cp_off, F->cp_count, F->cp_spacing, lane, &total, lds,
                                     RUNS_LDS);
    if (ok == 3) ok = zero_run_chain(src + F->seq_off, F->seq_len, pairs, nseq, llt, mlt, 6, 6, regen, lane, &total);
    if (ok != 1 || total != base_out) {  // corrupt, not a pure zero-run block, or not what the other blocks leave of the content size
        if (lane == 0) redo[r] = 1;
        return;
    }
    __syncthreads();  // the pairs are in memory
    const uint32_t end = place_zero_runs(dst, pairs, nseq, dst + F->ws_lit, 2u, regen, 0u, F->fcs, F->block_max, reinterpret_cast<uint8_t*>(lds), RUNS_LDS - 8u, lane);
    if (lane == 0) {
        if (end != base_out) redo[r] = 1;
        else b.result[r] = F->fcs;
    }
}

}  // namespace

// VBZ_HIP_REF_LITERALS=0: no literals ahead of the decoder (measurements, tests of the other path)
static uint32_t ref_units_max()
{
    static const uint32_t v = [] {
        const char* e = getenv("VBZ_HIP_REF_UNITS");   // (measurements: blocks per frame whose literals are decoded beside the walk)
        const int k = e ? atoi(e) : (int)4;
        return (uint32_t)(k < 1 ? 1 : (k > 4 ? 4 : k));
    }();
    return v;
}

static int ref_lits_ahead()
{
    static const int on = [] {
        const char* e = getenv("VBZ_HIP_REF_LITERALS");
        return e ? atoi(e) : 1;
    }();
    return on;
}

size_t zstd_fast_meta_bytes(uint32_t n_reads)
{
    return (size_t)n_reads * (sizeof(FastFrame) + FAST_TASKS * sizeof(FastTask) + 512 + 4 + 4) + 1024;
}

// what the batched decoder keeps per read in the call's scratch (zstd_fast_meta_bytes)
struct FastMeta
{
    FastFrame* frames;
    FastTask* tasks;
    uint8_t* weights;
    uint32_t* redo;
    uint32_t* scanned;   // redo[] as the scan left it: what the chain walk goes by
};
// ... and what the literals of the reference's frames take (ref_lit_scan / ref_pieces_kernel), REF_UNITS records per read: a buffer of its
// own (zstd_ref_lit_meta_bytes), only there when chains are walked
struct RefLitMeta
{
    FastFrame* frames;
    FastTask* tasks;
    uint8_t* weights;
    RefLits* lits;
    uint32_t* skip;
    uint32_t* pos;      // the stripes' first literals: WAVE per unit
};
static RefLitMeta ref_lit_meta(void* meta, uint32_t n)
{
    RefLitMeta M;
    const size_t nu = (size_t)n * REF_UNITS;
    uint8_t* m = reinterpret_cast<uint8_t*>(meta);
    M.frames = reinterpret_cast<FastFrame*>(m);
    m += nu * sizeof(FastFrame);
    M.tasks = reinterpret_cast<FastTask*>(m);
    m += nu * REF_TASKS * sizeof(FastTask);
    M.weights = m;
    m += nu * 512;
    M.lits = reinterpret_cast<RefLits*>(m);
    m += nu * sizeof(RefLits);
    M.skip = reinterpret_cast<uint32_t*>(m);
    m += nu * 4;
    m += (256 - (reinterpret_cast<uintptr_t>(m) & 255)) & 255;
    M.pos = reinterpret_cast<uint32_t*>(m);
    return M;
}
size_t zstd_ref_lit_meta_bytes(uint32_t n_reads)
{
    return (size_t)n_reads * REF_UNITS * (sizeof(FastFrame) + REF_TASKS * sizeof(FastTask) + 512 + sizeof(RefLits) + 4 + 4 * WAVE) + 1024;
}
uint32_t zstd_ref_lit_units() { return REF_UNITS; }
static FastMeta fast_meta(void* meta, uint32_t n)
{
    FastMeta M;
    uint8_t* m = reinterpret_cast<uint8_t*>(meta);
    M.frames = reinterpret_cast<FastFrame*>(m);
    m += (size_t)n * sizeof(FastFrame);
    M.tasks = reinterpret_cast<FastTask*>(m);
    m += (size_t)n * FAST_TASKS * sizeof(FastTask);
    M.weights = m;
    m += (size_t)n * 512;
    M.redo = reinterpret_cast<uint32_t*>(m);
    m += (size_t)n * 4;
    M.scanned = reinterpret_cast<uint32_t*>(m);
    m += (size_t)n * 4;
    return M;
}

const RefLits* zstd_ref_lits(const void* lit_meta, uint32_t n_reads) { return ref_lit_meta(const_cast<void*>(lit_meta), n_reads).lits; }
const uint32_t* zstd_ref_lit_skip(const void* lit_meta, uint32_t n_reads) { return ref_lit_meta(const_cast<void*>(lit_meta), n_reads).skip; }
bool zstd_ref_literals_enabled() { return ref_lits_ahead() != 0; }

const uint32_t* zstd_fast_redo(const void* meta, uint32_t n_reads) { return fast_meta(const_cast<void*>(meta), n_reads).redo; }

hipError_t launch_zstd_decode_fast(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, void* meta, void* ref_pre, void* ref_tables,
                                   void* ref_recs, uint64_t ref_recs_cap, void* ref_lits, uint32_t ref_units, unsigned long long* dbg, FastSide side,
                                   hipStream_t s)
{
    const uint32_t n = b.n_reads;
    if (n == 0) return hipSuccess;
    const FastMeta M = fast_meta(meta, n);
    const RefLitMeta R = ref_lit_meta(ref_lits, n);
    FastFrame* const frames = M.frames;
    FastTask* const tasks = M.tasks;
    uint8_t* const weights = M.weights;
    uint32_t* const redo = M.redo;
    uint32_t* const scanned = M.scanned;
    hipError_t e = hipSuccess;
    if (dbg) {  // phase timing: every frame to the one-wavefront decoder
        e = hipMemsetAsync(redo, 1, 4ull * n, s);
        if (e != hipSuccess) return e;
    } else {
        hipLaunchKernelGGL(fast_scan_kernel, dim3((n + 255) / 256), dim3(256), 0, s, b, frames, tasks, redo);
    }
    // What the scan does not recognise may be a frame the reference wrote: its sequence chains, one lane per frame -- beside the
    // launches for this library's own frames when the caller has a second stream for it (the walk is a few hundred wavefronts
    // bound by latency; a batch of both kinds of frame would otherwise wait for it).  From the fork on nothing returns before
    // the join is queued.
    RefChains chains;
    const bool beside = ref_pre && side.stream && !dbg;
    const bool lits = ref_pre && ref_lits && ref_lits_ahead();
    // (units per read that get a record -- and a workgroup each of the pieces' kernel, most of them empty in a call of short reads: the
    // caller's word, by the call's average content)
    const uint32_t upr = std::min(std::max(ref_units, 1u), std::min(ref_units_max(), REF_UNITS));
    const uint32_t nu = n * upr;
    if (ref_pre) {
        e = hipMemcpyAsync(scanned, redo, 4ull * n, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
        if (lits) {
            // which blocks, and their trees: IN FRONT of the fork -- a tenth of a millisecond by themselves, but beside the walk the tree
            // reader (one lane per tree, a chain of LDS look-ups) took 0.66 ms for the two trees a frame of long reads, and the pieces wait for it
            hipLaunchKernelGGL(ref_lit_scan_kernel, dim3((n + 255) / 256), dim3(256), 0, s, b, scanned, R.frames, R.tasks, R.skip, upr);
            hipLaunchKernelGGL(fast_weights_kernel, dim3((2 * nu + WAVE - 1) / WAVE), dim3(WAVE), 0, s, b, R.frames, R.weights, R.skip, nu);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        if (beside) {
            e = hipEventRecord(side.fork, s);
            if (e != hipSuccess) return e;
            e = hipStreamWaitEvent(side.stream, side.fork, 0);
            if (e != hipSuccess) return e;
            e = hipEventRecord(side.join, side.stream);   // "the second stream has come as far as the walk's launch": see below
        }
        const hipError_t e1 = launch_zstd_ref_chain(b, scanned, ref_pre, ref_tables, ref_recs, ref_recs_cap, &chains, beside ? side.stream : s);
        if (e == hipSuccess) e = e1;
    }
    if (!dbg && e == hipSuccess) {
        hipLaunchKernelGGL(fast_weights_kernel, dim3((2 * n + WAVE - 1) / WAVE), dim3(WAVE), 0, s, b, frames, weights, redo, n);
        hipLaunchKernelGGL(fast_streams_kernel, dim3(n), dim3(WAVE), 0, s, b, frames, tasks, weights, redo);
        hipLaunchKernelGGL(fast_runs_kernel, dim3(n), dim3(WAVE), 0, s, b, frames, reinterpret_cast<const SeqDTables*>(seq_dtables), redo);
        e = hipGetLastError();
    }
    // the literals of the frames being walked, meanwhile
    if (lits && e == hipSuccess) {
        // The pieces' 16 384 wavefronts must not be on the device before the walk's 256: a walk wavefront holds 60 KB of LDS and finds no CU
        // with that much free once twelve of these sit on each -- it then starts a round of pieces late, and the call ends 0.8 ms later
        // (measured: calls of 4.6 and of 5.4 ms, by which queue was served first).  So this stream waits until the other has reached the walk.
        if (beside) {
            const hipError_t e1 = hipStreamWaitEvent(s, side.join, 0);
            if (e == hipSuccess) e = e1;
        }
        hipLaunchKernelGGL(ref_pieces_kernel, dim3(nu), dim3(WAVE), 0, s, b, R.frames, R.tasks, R.weights, R.skip, R.lits, R.pos);
        chains.lits_pos = R.pos;
        e = hipGetLastError();
        chains.lits = R.lits;
        chains.lits_units = upr;
    }
    if (beside) {   // (the join proper: the event again, now behind the walk)
        const hipError_t e1 = hipEventRecord(side.join, side.stream);
        const hipError_t e2 = hipStreamWaitEvent(s, side.join, 0);
        if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
    }
    if (e != hipSuccess) return e;
    return launch_zstd_decode_only(b, toosmall_code, seq_dtables, redo, chains, dbg, s);  // whatever is not of the shape, and every error verdict
}

}  // namespace vbzhip
