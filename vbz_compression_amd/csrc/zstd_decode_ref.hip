// zstd_decode_ref.hip -- frames the REFERENCE wrote (vbz/vbz.cpp:176-189: ZSTD_compress of the svb stream with libzstd, level 1 by
// default): the sequence chains of a batch of such frames, walked one LANE per frame before the one-wavefront decoder runs.
//
// A libzstd frame of a read is one or two 128 KB blocks, each with four Huffman literal streams and ~1 100 sequences whose literal
// lengths, match lengths and offsets come from three FSE state machines reading ONE backward bit stream (RFC 8878 3.1.1.3.2): a
// dependent chain of ~1 100 table look-ups that no wavefront can spread over its lanes.  zstd_decode_kernel walks it on three lanes of
// its one wavefront (~760 cycles per sequence with the other 61 lanes idle: 0.84 M of the 2.2 M cycles it spends on such a frame).
// Here the chain of a frame is ONE LANE's work, and a wavefront walks 64 frames' chains at once:
//
//   ref_chain_kernel   lane per frame: frame header, every block header, the literals section header (only its sizes), the sequences
//                      section header, the three table descriptions (RFC 8878 4.1.1) built into the lane's own tables in memory, the
//                      chain -> 16-byte records {literal length, match length, offset} in a workspace the call owns.
//
// zstd_decode_kernel then finds `RefPre.ok` for the frame and takes the records instead of building tables and walking the chain
// itself (its phases B and C -- literals from prefix sums, matches in dependency rounds -- and every other check stay where they are).
// Nothing is trusted and nothing is decided here: a frame that fails ANY check on the way -- or whose shape this kernel does not
// handle (more than REF_MAXBLK blocks with sequences, the zero-run blocks of zstd_encode.hip, a sequences
// header longer than REF_HDR bytes, no room in the workspace) -- is left with ok = 0 and the one-wavefront decoder treats it exactly
// as before, which is also what produces every error verdict.  The accept conditions are those of general_sequence_records
// (zstd_decode.hip) and of read_ncount / fse_build / seq_table there, restated per lane.
#include "vbz_kernels.h"
#include "zstd_runs.h"

namespace vbzhip {

namespace {

constexpr int REF_HDR = 192;  // staged bytes of a sequences section header (count, modes, three table descriptions)
constexpr uint32_t T_LL = 0, T_OF = 512, T_ML = 768, T_ALL = 1280;  // the lane's tables: entries of {baseline, base | nb << 16 | extra bits << 24}

__device__ __forceinline__ uint32_t rld32(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t rld64(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

template <int FPW>
struct RefLds  // [index][lane]: a wavefront's lanes reading the same index of their own arrays hit FPW different banks
{
    int16_t norm[64][FPW];
    uint16_t symnext[64][FPW];
    uint32_t hdr[REF_HDR / 4 + 2][FPW];
};

// bits [bitpos, bitpos + k) of the lane's staged header bytes, k <= 17 (little-endian bit order: RFC 8878 4.1.1)
template <int FPW>
__device__ __forceinline__ uint32_t hdr_bits(const RefLds<FPW>& S, int l, uint32_t bitpos, int k)
{
    const uint32_t w = bitpos >> 5;
    const uint64_t two = (uint64_t)S.hdr[w][l] | ((uint64_t)S.hdr[w + 1][l] << 32);
    return (uint32_t)(two >> (bitpos & 31u)) & ((1u << k) - 1u);
}
template <int FPW>
__device__ __forceinline__ uint32_t hdr_byte(const RefLds<FPW>& S, int l, uint32_t i)
{
    return (S.hdr[i >> 2][l] >> (8u * (i & 3u))) & 0xFFu;
}

// read_ncount of zstd_decode.hip for one lane: the description starts at byte `at` of the staged header, n bytes are there.
// Returns bytes consumed or -1; fills the lane's norm[0..nsym).
template <int FPW>
__device__ int ref_read_ncount(RefLds<FPW>& S, int l, uint32_t at, int n, int max_symbol, int max_log, int* out_log, int* out_nsym)
{
    if (n < 1) return -1;
    auto bits = [&](uint32_t bitpos, int k) -> uint32_t {  // bytes beyond n read as zero
        const uint32_t avail = 8u * (uint32_t)n;
        if (bitpos >= avail) return 0u;
        uint32_t v = hdr_bits(S, l, 8u * at + bitpos, k);
        if (bitpos + (uint32_t)k > avail) v &= (1u << (avail - bitpos)) - 1u;
        return v;
    };
    const int log = (int)(hdr_byte(S, l, at) & 0xF) + 5;
    if (log > max_log) return -1;
    uint32_t bitpos = 4;
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
    bool prev0 = false;
    while (remaining > 1 && sym <= max_symbol) {
        if (bitpos > 8u * (uint32_t)n + 32u) return -1;  // (far beyond the description: stop; the byte count below fails anyway)
        if (prev0) {
            for (;;) {
                const uint32_t rr = bits(bitpos, 2);
                bitpos += 2;
                for (uint32_t k = 0; k < rr; ++k) {
                    if (sym > max_symbol) return -1;
                    S.norm[sym++][l] = 0;
                }
                if (rr != 3) break;
                if (bitpos > 8u * (uint32_t)n + 32u) return -1;
            }
            prev0 = false;
            if (sym > max_symbol) break;
            continue;
        }
        const int max = (2 * threshold - 1) - remaining;
        const uint32_t v = bits(bitpos, nbits);
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
        S.norm[sym++][l] = (int16_t)count;
        prev0 = (count == 0);
        while (remaining < threshold) {
            nbits--;
            threshold >>= 1;
        }
    }
    if (remaining != 1) return -1;
    if (sym > max_symbol + 1) return -1;
    const int used = (int)((bitpos + 7) >> 3);
    if (used > n) return -1;
    *out_log = log;
    *out_nsym = sym;
    return used;
}

// what a code stands for: kind 0 literal lengths, 1 offsets, 2 match lengths -> {baseline, extra bits}
__device__ __forceinline__ uint2 code_value(int kind, uint32_t code)
{
    if (kind == 1) return make_uint2(1u << (code & 31u), code);
    if (kind == 0) return make_uint2(LL_BASE[code], LL_BITS[code]);
    return make_uint2(ML_BASE[code], ML_BITS[code]);
}

// fse_build of zstd_decode.hip for one lane, into the lane's table in memory (RFC 8878 4.1.1); entries as described at T_LL
template <int FPW>
__device__ bool ref_fse_build(RefLds<FPW>& S, int l, uint2* tab, int nsym, int log, int kind)
{
    const int size = 1 << log;
    int high = size - 1;
    for (int s = 0; s < nsym; ++s) {
        const int c = S.norm[s][l];
        if (c == -1) {
            tab[high--].x = (uint32_t)s;
            S.symnext[s][l] = 1;
        } else {
            S.symnext[s][l] = (uint16_t)c;
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; ++s) {
        const int c = S.norm[s][l];
        for (int i = 0; i < c; ++i) {
            tab[pos].x = (uint32_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    }
    if (pos != 0) return false;
    for (int u = 0; u < size; ++u) {
        const uint32_t s = tab[u].x & 0xFFu;
        const uint32_t ns = S.symnext[s][l];
        S.symnext[s][l] = (uint16_t)(ns + 1);
        const int nb = log - hbit(ns);
        const uint2 cv = code_value(kind, s);
        tab[u] = make_uint2(cv.x, ((((ns << nb) - (uint32_t)size) & 0xFFFFu)) | ((uint32_t)nb << 16) | (cv.y << 24));
    }
    return true;
}

// seq_table of zstd_decode.hip for one lane.  Returns bytes consumed from the staged header at `at`, or -1.
template <int FPW>
__device__ int ref_seq_table(RefLds<FPW>& S, int l, uint2* tab, int* log_io, bool* have, int mode, uint32_t at, int n, int kind)
{
    const int16_t* def = kind == 0 ? LL_DEFAULT : (kind == 1 ? OF_DEFAULT : ML_DEFAULT);
    const int def_n = kind == 0 ? 36 : (kind == 1 ? 29 : 53), def_log = kind == 1 ? 5 : 6;
    const int max_sym = kind == 0 ? 35 : (kind == 1 ? 31 : 52), max_log = kind == 1 ? 8 : 9;
    if (mode == 0) {
        for (int i = 0; i < def_n; ++i) S.norm[i][l] = def[i];
        if (!ref_fse_build(S, l, tab, def_n, def_log, kind)) return -1;
        *log_io = def_log;
        *have = true;
        return 0;
    }
    if (mode == 1) {
        if (n < 1) return -1;
        const uint32_t code = hdr_byte(S, l, at);
        if (code > (uint32_t)max_sym) return -1;
        const uint2 cv = code_value(kind, code);
        tab[0] = make_uint2(cv.x, cv.y << 24);
        *log_io = 0;
        *have = true;
        return 1;
    }
    if (mode == 2) {
        int log, nsym;
        const int used = ref_read_ncount(S, l, at, n, max_sym, max_log, &log, &nsym);
        if (used < 0) return -1;
        if (!ref_fse_build(S, l, tab, nsym, log, kind)) return -1;
        *log_io = log;
        *have = true;
        return used;
    }
    return *have ? 0 : -1;
}

// The 64 bits that follow bit s (counted from the top) of the 128-bit value hi:lo, s <= 127
__device__ __forceinline__ uint64_t top64(uint64_t hi, uint64_t lo, uint32_t s)
{
    const uint64_t a = (s & 64u) ? lo : hi, c = (s & 64u) ? 0ull : lo;
    const uint32_t t = s & 63u;
    return (a << t) | (t ? c >> (64u - t) : 0ull);
}

// The chain of one sequences section for one lane: general_sequence_records of zstd_decode.hip, the same accept conditions.
// Returns 0, or why not (a diagnostic).  bs / bsn: the bit stream; arena_lo: how many bytes below bs may be read (they belong to the arena).  rep: in and out.
// *out_end: output position behind the block's last match (the literals behind it are the caller's).
__device__ uint32_t ref_chain(const uint8_t* bs, uint32_t bsn, uint64_t arena_lo, const uint2* tab, uint32_t log_ll, uint32_t log_of, uint32_t log_ml,
                          uint4* rec, uint32_t nseq, uint32_t regen, uint32_t opos0, uint32_t fcs, uint32_t (&rep)[3], uint64_t* out_end)
{
    if (bsn == 0) return 101u;
    const uint32_t top = bs[bsn - 1];
    if (top == 0) return 102u;
    const uint32_t total_bits = 8u * bsn;
    uint32_t pos = 8u - (uint32_t)hbit(top);  // bits consumed, counted from the end of the stream
    // the 16 bytes that end with the byte the next unread bit is in: hi = the upper eight (little-endian), lo = the lower eight
    auto window = [&](uint32_t p, uint64_t& hi, uint64_t& lo) {
        const uint32_t e = bsn - (p >> 3);  // bytes [e - 16, e) of the stream; p <= total_bits: e >= 0
        if ((uint64_t)e + arena_lo >= 16) {
            lo = rld64(bs + (int64_t)e - 16);
            hi = rld64(bs + (int64_t)e - 8);
        } else {  // the first bytes of the arena: assemble what is there
            hi = lo = 0;
            for (uint32_t k = 0; k < 16; ++k) {
                const int64_t at = (int64_t)e - 1 - (int64_t)k;  // byte k from the top
                if (at + (int64_t)arena_lo < 0) break;
                const uint64_t v = bs[at];
                if (k < 8) hi |= v << (56 - 8 * k);
                else lo |= v << (56 - 8 * (k - 8));
            }
        }
    };
    uint64_t hi, lo;
    window(pos, hi, lo);
    uint32_t sl, so, sm;
    {
        const uint32_t sh = pos & 7u;
        const uint64_t w = top64(hi, lo, sh);  // log_ll + log_of + log_ml <= 26 bits
        const uint32_t all = log_ll + log_of + log_ml;
        const uint32_t v = all ? (uint32_t)(w >> (64u - all)) : 0u;
        sl = log_ll ? v >> (log_of + log_ml) : 0u;
        so = (v >> log_ml) & ((1u << log_of) - 1u);
        sm = v & ((1u << log_ml) - 1u);
        pos += all;
    }
    if (pos > total_bits) return 103u;
    uint32_t rep0 = rep[0], rep1 = rep[1], rep2 = rep[2];
    uint64_t sum_ll = 0, outp = opos0;
    bool astray = false;
    for (uint32_t i = 0; i < nseq; ++i) {
        const uint2 el = tab[T_LL + sl], eo = tab[T_OF + so], em = tab[T_ML + sm];
        window(pos, hi, lo);
        const uint32_t xl = el.y >> 24, xo = eo.y >> 24, xm = em.y >> 24;
        const uint32_t xt = xo + xm + xl;  // <= 31 + 16 + 16: offset bits first, then match length, then literal length
        const uint32_t sh = pos & 7u;
        const uint64_t wx = top64(hi, lo, sh);
        const uint64_t X = xt ? wx >> (64u - xt) : 0ull;
        const uint32_t ofv = eo.x + (uint32_t)(X >> (xm + xl));
        const uint32_t mlen = em.x + ((uint32_t)(X >> xl) & ((1u << xm) - 1u));
        const uint32_t llen = el.x + ((uint32_t)X & ((1u << xl) - 1u));
        const bool lastseq = i + 1 == nseq;
        const uint32_t nl = lastseq ? 0u : (el.y >> 16) & 0xFFu, nm = lastseq ? 0u : (em.y >> 16) & 0xFFu, no = lastseq ? 0u : (eo.y >> 16) & 0xFFu;
        const uint32_t nt = nl + nm + no;  // <= 26: the states move on in the order LL, ML, OF
        const uint64_t ws = top64(hi, lo, sh + xt);
        const uint32_t v = nt ? (uint32_t)(ws >> (64u - nt)) : 0u;
        sl = (el.y & 0xFFFFu) + (v >> (nm + no));
        sm = (em.y & 0xFFFFu) + ((v >> no) & ((1u << nm) - 1u));
        so = (eo.y & 0xFFFFu) + (v & ((1u << no) - 1u));
        pos += xt + nt;
        if (pos > total_bits) return 104u;
        // repeat offsets (RFC 8878 3.1.1.5): idx 0 = rep0 as it is, 1 = rep1, 2 = rep2, 3 = rep0 - 1
        const bool isrep = ofv <= 3;
        const uint32_t idx = ofv - 1 + (llen == 0 ? 1u : 0u);
        uint32_t cand = idx == 1 ? rep1 : (idx == 2 ? rep2 : rep0 - (idx == 3 ? 1u : 0u));
        cand = cand ? cand : 1u;  // libzstd forces an invalid 0 to 1
        const uint32_t offset = isrep ? cand : ofv - 3;
        rep2 = (isrep && idx <= 1) ? rep2 : rep1;
        rep1 = (isrep && idx == 0) ? rep1 : rep0;
        rep0 = offset;
        sum_ll += llen;
        outp += llen;
        astray = astray || offset > outp;
        outp += mlen;
        rec[i] = make_uint4(llen, mlen, offset, 0u);
    }
    if (astray) return 105u;
    if (sum_ll > regen || outp > fcs || outp - opos0 > BLOCK_MAX) return 106u;
    if (pos != total_bits) return 107u;
    rep[0] = rep0;
    rep[1] = rep1;
    rep[2] = rep2;
    *out_end = outp + (regen - sum_ll);
    return 0u;
}

// One frame, one lane.  Returns true when every block with sequences of the frame has its records in the workspace.
// (why a frame was left alone: RefPre.pad[0], a diagnostic that vbz_gpu_decode_paths prints under VBZ_HIP_TRACE)
#define BAIL(k) do { P->pad[0] = (k); return false; } while (0)
template <int FPW>
__device__ bool ref_frame(RefLds<FPW>& S, int l, const ReadBatch& b, uint32_t r, RefPre* P, uint2* tab, uint4* recs, uint64_t recs_cap,
                          unsigned long long* recs_used)
{
    if (b.gate && b.gate[r] >= GATE_SKIP) BAIL(1);
    const uint32_t n = b.src_size[r];
    if (n >= E_FIRST || n < 32) BAIL(2);
    const uint64_t src_off = b.src_off[r];
    const uint8_t* src = b.src + src_off;
    const uint32_t cap = b.dst_cap[r];
    uint32_t pos, fcs, block_max;
    {
        const uint64_t h0 = rld64(src), h1 = rld64(src + 8);
        auto hb = [&](uint32_t i) -> uint32_t { return (uint32_t)((i < 8 ? h0 >> (8 * i) : h1 >> (8 * (i - 8))) & 0xFF); };
        const uint32_t fhd = hb(4);
        if ((uint32_t)h0 != 0xFD2FB528u || (fhd & 0x08) || (fhd & 3)) BAIL(3);  // (a Dictionary_ID field: the careful decoder)
        const uint32_t single = (fhd >> 5) & 1, fcs_flag = fhd >> 6;
        pos = 5;
        uint64_t window = 0;
        if (!single) {
            const uint32_t wd = hb(pos++);
            const uint32_t wlog = 10 + (wd >> 3);
            if (wlog > 31) BAIL(4);
            window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7);
        }
        const uint32_t fsz = fcs_flag == 0 ? (single ? 1u : 0u) : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
        if (fsz == 0) BAIL(5);
        uint64_t f = 0;
        for (uint32_t i = 0; i < fsz; ++i) f |= (uint64_t)hb(pos + i) << (8 * i);
        if (fsz == 2) f += 256;
        pos += fsz;
        if (f > cap || f >= (1u << 30)) BAIL(6);
        if (single) window = f;
        fcs = (uint32_t)f;
        block_max = (uint32_t)(window < BLOCK_MAX ? window : BLOCK_MAX);
    }
    uint32_t nblk = 0, opos = 0;
    uint64_t frame_ns = 0;
    uint32_t rep[3] = { 1, 4, 8 };
    bool have_ll = false, have_of = false, have_ml = false;
    int log_ll = 0, log_of = 0, log_ml = 0;
    for (;;) {
        if (pos + 3 > n) BAIL(7);
        const uint32_t bh = rld32(src + pos) & 0xFFFFFFu;
        const uint32_t block_at = pos;
        pos += 3;
        const uint32_t last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3) BAIL(8);
        if (btype == 0 || btype == 1) {
            if (bsize > block_max || (uint64_t)pos + (btype == 0 ? bsize : 1u) > n || (uint64_t)opos + bsize > fcs) BAIL(9);
            opos += bsize;
            pos += btype == 0 ? bsize : 1u;
        } else {
            if (bsize >= BLOCK_MAX || (uint64_t)pos + bsize > n || bsize < 2) BAIL(10);
            const uint8_t* blk = src + pos;
            uint32_t lh, regen, csize;
            {
                const uint64_t v = rld64(blk);
                const uint32_t h0 = (uint32_t)v & 0xFF, fmt = (h0 >> 2) & 3, ltype = h0 & 3;
                if (ltype < 2) {
                    if (fmt == 0 || fmt == 2) { lh = 1; regen = h0 >> 3; }
                    else if (fmt == 1) { lh = 2; regen = ((uint32_t)v & 0xFFFFu) >> 4; }
                    else { lh = 3; regen = ((uint32_t)v & 0xFFFFFFu) >> 4; }
                    if (lh > bsize) BAIL(11);
                    csize = ltype == 0 ? regen : 1;
                } else {
                    if (bsize < 5) BAIL(12);
                    if (fmt < 2) { lh = 3; regen = (uint32_t)(v >> 4) & 0x3FF; csize = (uint32_t)(v >> 14) & 0x3FF; }
                    else if (fmt == 2) { lh = 4; regen = (uint32_t)(v >> 4) & 0x3FFF; csize = (uint32_t)(v >> 18) & 0x3FFF; }
                    else { lh = 5; regen = (uint32_t)(v >> 4) & 0x3FFFF; csize = (uint32_t)(v >> 22) & 0x3FFFF; }
                    if (regen == 0 || csize == 0) BAIL(13);
                }
                if (regen > BLOCK_MAX || (uint64_t)lh + csize >= bsize) BAIL(14);
            }
            const uint32_t lit_end = lh + csize;
            const uint8_t* sq = blk + lit_end;
            const uint32_t sqn = bsize - lit_end;
            if ((uint64_t)opos + regen > fcs) BAIL(15);
            if (sq[0] == 0) {  // no sequences: the block is its literals
                if (sqn != 1 || regen > block_max) BAIL(16);
                opos += regen;
            } else {
                if (nblk == REF_MAXBLK) BAIL(17);
                // stage the header of the sequences section: 16 bytes per load, all of them in flight at once
                const uint32_t hn = sqn < (uint32_t)REF_HDR ? sqn : (uint32_t)REF_HDR;
                for (uint32_t k = 0; 16 * k < hn; ++k) {
                    const uint64_t a = rld64(sq + 16 * k), c = rld64(sq + 16 * k + 8);
                    S.hdr[4 * k][l] = (uint32_t)a;
                    S.hdr[4 * k + 1][l] = (uint32_t)(a >> 32);
                    S.hdr[4 * k + 2][l] = (uint32_t)c;
                    S.hdr[4 * k + 3][l] = (uint32_t)(c >> 32);
                }
                uint32_t ns = hdr_byte(S, l, 0), used = 1;
                if (ns >= 128) {
                    if (ns == 255) {
                        if (hn < 3) BAIL(18);
                        ns = hdr_byte(S, l, 1) + (hdr_byte(S, l, 2) << 8) + 0x7F00;
                        used = 3;
                    } else {
                        if (hn < 2) BAIL(19);
                        ns = ((ns - 128) << 8) + hdr_byte(S, l, 1);
                        used = 2;
                    }
                }
                if (used >= hn) BAIL(20);
                const uint32_t modes = hdr_byte(S, l, used++);
                if (modes & 3) BAIL(21);
                int u = ref_seq_table(S, l, tab + T_LL, &log_ll, &have_ll, (modes >> 6) & 3, used, (int)(hn - used), 0);
                if (u < 0) BAIL(23);
                used += (uint32_t)u;
                // predefined length tables and offsets that are all "repeat offset 1" (an RLE table of code 0): what zstd_encode.hip
                // writes for its zero-run blocks (a frame of this library that the batched decoder did not take: raw blocks, say),
                // with checkpoints behind the frame that let the one-wavefront decoder walk the chain in parallel segments; left to it.
                // (libzstd's frames of nanopore signal have the same offsets -- its matches are the zero runs of the control bytes --
                // but FSE-coded length tables of 9 bits, which that path does not take: they are walked here)
                if (modes == 0x10u && used < hn && hdr_byte(S, l, used) == 0) BAIL(22);
                u = ref_seq_table(S, l, tab + T_OF, &log_of, &have_of, (modes >> 4) & 3, used, (int)(hn - used), 1);
                if (u < 0) BAIL(24);
                used += (uint32_t)u;
                u = ref_seq_table(S, l, tab + T_ML, &log_ml, &have_ml, (modes >> 2) & 3, used, (int)(hn - used), 2);
                if (u < 0) BAIL(25);
                used += (uint32_t)u;
                if (used >= hn) BAIL(26);  // (hn < sqn: a header longer than the staged bytes is left to the careful decoder)
                // a frame may claim 16 bytes of records per 16 bytes of its content (libzstd on nanopore signal: one sequence per ~110
                // bytes): the claims of a call then fit a workspace of the size of the call's content whatever the frames are
                frame_ns += ns;
                if (16ull * frame_ns > fcs) BAIL(31);
                const unsigned long long first = atomicAdd(recs_used, (unsigned long long)ns);
                if (first + ns > recs_cap) BAIL(27);  // (a workspace capped below the content: nothing behind this claim fits either)
                uint64_t end = 0;
                const uint32_t why = ref_chain(sq + used, sqn - used, src_off + pos + lit_end + used, tab, (uint32_t)log_ll, (uint32_t)log_of,
                                               (uint32_t)log_ml, recs + first, ns, regen, opos, fcs, rep, &end);
                if (why) BAIL(why);
                if (end > fcs || end - opos > BLOCK_MAX || end - opos > block_max) BAIL(29);
                RefBlock& B = P->blk[nblk++];
                B.pos = block_at;
                B.nseq = ns;
                B.rec_lo = (uint32_t)first;
                B.rec_hi = (uint32_t)(first >> 32);
                B.rep[0] = rep[0];
                B.rep[1] = rep[1];
                B.rep[2] = rep[2];
                B.end = (uint32_t)end;
                opos = (uint32_t)end;
            }
            pos += bsize;
        }
        if (last) break;
    }
    if (nblk == 0) BAIL(30);  // nothing to hand over
    P->nblk = nblk;
    return true;
}

#undef BAIL

// REF_SLOTS lanes in all (FPW of a wavefront's 64 carry a frame: fewer frames per wavefront = fewer scattered requests per
// instruction, more wavefronts); lane `slot` takes frames slot, slot + REF_SLOTS, ... and owns table area `slot`.
template <int FPW>
__global__ __launch_bounds__(WAVE) void ref_chain_kernel(ReadBatch b, const uint32_t* redo, RefPre* pre, uint2* tables, uint4* recs,
                                                         uint64_t recs_cap, unsigned long long* recs_used)
{
    __shared__ RefLds<FPW> S;
    const int l = threadIdx.x;
    if (l >= FPW) return;
    const uint32_t slot = blockIdx.x * (uint32_t)FPW + (uint32_t)l;
    uint2* tab = tables + (size_t)slot * T_ALL;
    for (uint32_t r = slot; r < b.n_reads; r += REF_SLOTS) {
        RefPre* P = pre + r;
        bool ok = false;
        if (redo[r]) ok = ref_frame<FPW>(S, l, b, r, P, tab, recs, recs_cap, recs_used);
        P->ok = ok ? 1u : 0u;
    }
}

}  // namespace

size_t zstd_ref_pre_bytes(uint32_t n_reads) { return (size_t)n_reads * sizeof(RefPre) + 64; }
const RefPre* zstd_ref_pre(const void* pre_meta) { return reinterpret_cast<const RefPre*>(reinterpret_cast<const uint8_t*>(pre_meta) + 64); }
size_t zstd_ref_table_bytes() { return (size_t)REF_SLOTS * T_ALL * sizeof(uint2); }

#ifndef VBZ_REF_FPW
#define VBZ_REF_FPW 64
#endif

hipError_t launch_zstd_ref_chain(const ReadBatch& b, const uint32_t* redo, void* pre_meta, void* tables, void* recs, uint64_t recs_cap,
                                 RefChains* out, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    uint8_t* m = reinterpret_cast<uint8_t*>(pre_meta);
    unsigned long long* used = reinterpret_cast<unsigned long long*>(m);
    RefPre* pre = reinterpret_cast<RefPre*>(m + 64);
    out->pre = pre;
    out->recs = recs;
    hipError_t e = hipMemsetAsync(used, 0, 8, s);
    if (e != hipSuccess) return e;
    constexpr int FPW = VBZ_REF_FPW;
    hipLaunchKernelGGL(ref_chain_kernel<FPW>, dim3(REF_SLOTS / FPW), dim3(WAVE), 0, s, b, redo, pre, reinterpret_cast<uint2*>(tables),
                       reinterpret_cast<uint4*>(recs), recs_cap, used);
    return hipGetLastError();
}

}  // namespace vbzhip
