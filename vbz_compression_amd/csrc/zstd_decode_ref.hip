// zstd_decode_ref.hip -- frames the REFERENCE wrote (vbz/vbz.cpp:176-189: ZSTD_compress of the svb stream with libzstd, level 1 by
// default): the sequence chains of a batch of such frames, walked one LANE per frame before the one-wavefront decoder runs.
//
// A libzstd frame of a read is one or two 128 KB blocks, each with four Huffman literal streams and ~1 100 sequences whose literal
// lengths, match lengths and offsets come from three FSE state machines reading ONE backward bit stream (RFC 8878 3.1.1.3.2): a
// dependent chain of ~1 100 table look-ups that no wavefront can spread over its lanes.  zstd_decode_kernel walks it on three lanes of
// its one wavefront (~760 cycles per sequence with the other 61 lanes idle: 0.84 M of the 2.2 M cycles it spends on such a frame).
// Here the chain of a frame is ONE LANE's work, and what bounds a lane is the round trip to its tables, so the tables live in LDS:
//
//   ref_chain_kernel   lane per frame, REF_FPW frames per wavefront, four wavefronts per CU (the LDS holds 36 frames' tables):
//                      frame header, every block header, the literals section header (only its sizes), the sequences section
//                      header, the three table descriptions (RFC 8878 4.1.1) built into the lane's tables in LDS (3 bytes per state),
//                      the bit stream through a 256-byte ring per lane that is refilled half a ring ahead of its use, the chain ->
//                      16-byte records {literal length, match length, offset} in a workspace the call owns.
//                      (The first version kept the tables in memory, 10 KB per lane: ~1 us per sequence -- a memory round trip --
//                      whatever the number of frames per wavefront; LDS: see DESIGN.md 4.4.)
//
// zstd_decode_kernel then finds `RefPre.ok` for the frame and takes the records instead of building tables and walking the chain
// itself (its phases B and C -- literals from prefix sums, matches in dependency rounds -- and every other check stay where they are).
// Nothing is trusted and nothing is decided here: a frame that fails ANY check on the way -- or whose shape this kernel does not
// handle (more than REF_MAXBLK blocks with sequences, the zero-run blocks of zstd_encode.hip, a sequences header longer than REF_HDR
// bytes, more sequences than its share of the workspace) -- is left with ok = 0 and the one-wavefront decoder treats it exactly as
// before, which is also what produces every error verdict.  The accept conditions are those of general_sequence_records
// (zstd_decode.hip) and of read_ncount / fse_build / seq_table there, restated per lane.
#include <cstdlib>

#include "vbz_kernels.h"
#include "zstd_runs.h"

namespace vbzhip {

namespace {

constexpr int REF_FPW = 9;     // tables in LDS: frames per wavefront
constexpr int REF_GRID = 1024; //   wavefronts: four per CU, as many as the LDS takes
constexpr int REF_FPW_MEM = 64;              // tables in memory: frames per wavefront
constexpr uint32_t REF_MEM_WAVES_MAX = 1024; //   at most 65 536 frames in flight (a lane takes several frames beyond that)
constexpr uint32_t REF_TAB_BYTES = 3840;     //   a lane's tables: 1 280 states x (2 + 1) bytes
constexpr int REF_HDR = 160;   // staged bytes of a sequences section header (count, modes, three table descriptions)
constexpr int REF_RING = 64;   // dwords of bit stream per lane
constexpr uint32_t T_LL = 0, T_OF = 512, T_ML = 768, T_ALL = 1280;  // the lane's three tables, one entry per state

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

// [index][lane]: a wavefront's lanes reading the same index of their own arrays hit different banks.
// A state's entry is 3 bytes: bn = new-state base (9 bits) | state bits to read << 9 (4) | extra bits, low 3 << 13;
// cx = code (6 bits) | extra bits, high 2 << 6.  The tables live in LDS (INLDS: 36 frames per CU, a sequence costs an LDS round
// trip) or in memory (any number of frames in flight, a sequence costs a memory round trip; LDS then only holds the symbols of the
// table under construction -- building it in memory would be a dependent memory round trip per state).
template <int FPW, bool INLDS>
struct RefTabs;
template <int FPW>
struct RefTabs<FPW, true>
{
    uint16_t bn[T_ALL][FPW];
    uint8_t cx[T_ALL][FPW];
};
template <int FPW>
struct RefTabs<FPW, false>
{
    uint8_t spread[512][FPW];
};
struct RefGlobal  // the lane's tables in memory (!INLDS)
{
    __attribute__((address_space(1))) uint16_t* bn;
    __attribute__((address_space(1))) uint8_t* cx;
};
template <int FPW, bool INLDS>
struct RefLds
{
    RefTabs<FPW, INLDS> t;
    union
    {
        uint32_t ring[REF_RING + 4][FPW];  // the chain: dword j of the stream (counted from its end) in slot j % 64; slots 64.. mirror 0..
        struct                              // the tables' descriptions
        {
            uint32_t hdr[REF_HDR / 4 + 2][FPW];
            int16_t norm[64][FPW];
            uint16_t symnext[64][FPW];
        } p;
    } u;
    uint32_t llb[36], mlb[53];  // code -> baseline
};
template <int FPW, bool INLDS>
__device__ __forceinline__ uint32_t get_bn(const RefLds<FPW, INLDS>& S, RefGlobal G, int l, uint32_t at)
{
    if constexpr (INLDS) return S.t.bn[at][l];
    else return G.bn[at];
}
template <int FPW, bool INLDS>
__device__ __forceinline__ uint32_t get_cx(const RefLds<FPW, INLDS>& S, RefGlobal G, int l, uint32_t at)
{
    if constexpr (INLDS) return S.t.cx[at][l];
    else return G.cx[at];
}
// the symbol of state u of the table under construction at t0
template <int FPW, bool INLDS>
__device__ __forceinline__ void sp_put(RefLds<FPW, INLDS>& S, int l, uint32_t t0, uint32_t u, uint32_t sym)
{
    if constexpr (INLDS) S.t.cx[t0 + u][l] = (uint8_t)sym;
    else S.t.spread[u][l] = (uint8_t)sym;
}
template <int FPW, bool INLDS>
__device__ __forceinline__ uint32_t sp_get(const RefLds<FPW, INLDS>& S, int l, uint32_t t0, uint32_t u)
{
    if constexpr (INLDS) return S.t.cx[t0 + u][l];
    else return S.t.spread[u][l];
}

// bits [bitpos, bitpos + k) of the lane's staged header bytes, k <= 17 (little-endian bit order: RFC 8878 4.1.1)
template <int FPW, bool INLDS>
__device__ __forceinline__ uint32_t hdr_bits(const RefLds<FPW, INLDS>& S, int l, uint32_t bitpos, int k)
{
    const uint32_t w = bitpos >> 5;
    const uint64_t two = (uint64_t)S.u.p.hdr[w][l] | ((uint64_t)S.u.p.hdr[w + 1][l] << 32);
    return (uint32_t)(two >> (bitpos & 31u)) & ((1u << k) - 1u);
}
template <int FPW, bool INLDS>
__device__ __forceinline__ uint32_t hdr_byte(const RefLds<FPW, INLDS>& S, int l, uint32_t i)
{
    return (S.u.p.hdr[i >> 2][l] >> (8u * (i & 3u))) & 0xFFu;
}

// read_ncount of zstd_decode.hip for one lane: the description starts at byte `at` of the staged header, n bytes are there.
// Returns bytes consumed or -1; fills the lane's norm[0..nsym).
template <int FPW, bool INLDS>
__device__ int ref_read_ncount(RefLds<FPW, INLDS>& S, int l, uint32_t at, int n, int max_symbol, int max_log, int* out_log, int* out_nsym)
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
                    S.u.p.norm[sym++][l] = 0;
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
        S.u.p.norm[sym++][l] = (int16_t)count;
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

// extra bits of a code: kind 0 literal lengths, 1 offsets, 2 match lengths
__device__ __forceinline__ uint32_t code_bits(int kind, uint32_t code) { return kind == 1 ? code : (kind == 0 ? LL_BITS[code] : ML_BITS[code]); }

template <int FPW, bool INLDS>
__device__ __forceinline__ void put_entry(RefLds<FPW, INLDS>& S, RefGlobal G, int l, uint32_t at, uint32_t code, uint32_t nb, uint32_t base, int kind)
{
    const uint32_t x = code_bits(kind, code);
    const uint16_t bn = (uint16_t)(base | (nb << 9) | ((x & 7u) << 13));
    const uint8_t cx = (uint8_t)(code | ((x >> 3) << 6));
    if constexpr (INLDS) {
        S.t.bn[at][l] = bn;
        S.t.cx[at][l] = cx;
    } else {
        G.bn[at] = bn;
        G.cx[at] = cx;
    }
}

// fse_build of zstd_decode.hip for one lane (RFC 8878 4.1.1), in place in the lane's table at `t0`
template <int FPW, bool INLDS>
__device__ bool ref_fse_build(RefLds<FPW, INLDS>& S, RefGlobal G, int l, uint32_t t0, int nsym, int log, int kind)
{
    const int size = 1 << log;
    int high = size - 1;
    for (int s = 0; s < nsym; ++s) {
        const int c = S.u.p.norm[s][l];
        if (c == -1) {
            sp_put(S, l, t0, (uint32_t)high--, (uint32_t)s);
            S.u.p.symnext[s][l] = 1;
        } else {
            S.u.p.symnext[s][l] = (uint16_t)c;
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; ++s) {
        const int c = S.u.p.norm[s][l];
        for (int i = 0; i < c; ++i) {
            sp_put(S, l, t0, (uint32_t)pos, (uint32_t)s);
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    }
    if (pos != 0) return false;
    for (int u = 0; u < size; ++u) {
        const uint32_t s = sp_get(S, l, t0, (uint32_t)u);
        const uint32_t ns = S.u.p.symnext[s][l];
        S.u.p.symnext[s][l] = (uint16_t)(ns + 1);
        const int nb = log - hbit(ns);
        put_entry(S, G, l, t0 + (uint32_t)u, s, (uint32_t)nb, ((ns << nb) - (uint32_t)size) & 0x1FFu, kind);
    }
    return true;
}

// seq_table of zstd_decode.hip for one lane.  Returns bytes consumed from the staged header at `at`, or -1.
template <int FPW, bool INLDS>
__device__ int ref_seq_table(RefLds<FPW, INLDS>& S, RefGlobal G, int l, uint32_t t0, int* log_io, bool* have, int mode, uint32_t at, int n, int kind)
{
    const int16_t* def = kind == 0 ? LL_DEFAULT : (kind == 1 ? OF_DEFAULT : ML_DEFAULT);
    const int def_n = kind == 0 ? 36 : (kind == 1 ? 29 : 53), def_log = kind == 1 ? 5 : 6;
    const int max_sym = kind == 0 ? 35 : (kind == 1 ? 31 : 52), max_log = kind == 1 ? 8 : 9;
    if (mode == 0) {
        for (int i = 0; i < def_n; ++i) S.u.p.norm[i][l] = def[i];
        if (!ref_fse_build(S, G, l, t0, def_n, def_log, kind)) return -1;
        *log_io = def_log;
        *have = true;
        return 0;
    }
    if (mode == 1) {
        if (n < 1) return -1;
        const uint32_t code = hdr_byte(S, l, at);
        if (code > (uint32_t)max_sym) return -1;
        put_entry(S, G, l, t0, code, 0u, 0u, kind);
        *log_io = 0;
        *have = true;
        return 1;
    }
    if (mode == 2) {
        int log, nsym;
        const int used = ref_read_ncount(S, l, at, n, max_sym, max_log, &log, &nsym);
        if (used < 0) return -1;
        if (!ref_fse_build(S, G, l, t0, nsym, log, kind)) return -1;
        *log_io = log;
        *have = true;
        return used;
    }
    return *have ? 0 : -1;
}

// The 64 bits that follow bit s (counted from the top, s <= 127) of the 128-bit value hi:lo
__device__ __forceinline__ uint64_t top64(uint64_t hi, uint64_t lo, uint32_t s)
{
    const uint64_t a = (s & 64u) ? lo : hi, c = (s & 64u) ? 0ull : lo;
    const uint32_t t = s & 63u;
    return (a << t) | (t ? c >> (64u - t) : 0ull);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// (explicitly global: a FLAT access counts on the LDS counter as well, and the chain waits on that counter once per sequence)
typedef __attribute__((address_space(1), aligned(1))) const u32x4 ref_gld16;
typedef __attribute__((address_space(1))) u32x4 ref_gst16;
typedef __attribute__((address_space(1))) const uint8_t ref_gcu8;

// The chain of one sequences section for one lane: general_sequence_records of zstd_decode.hip, the same accept conditions.
// Returns 0, or why not (a diagnostic).  bs / bsn: the bit stream; arena_lo: how many bytes below bs may be read (they belong to
// the arena).  rep: in and out.  *out_end: output position behind the block (its trailing literals included).
// Per sequence ONE dependent LDS round trip: the entries of the next states and the four dwords of bit stream at the next position
// are requested as soon as both are known; the repeat-offset rules and the record's store run behind the request.
template <int FPW, bool INLDS>
__device__ uint32_t ref_chain(RefLds<FPW, INLDS>& S, RefGlobal G, int l, const uint8_t* bs, uint32_t bsn, uint64_t arena_lo, uint32_t log_ll, uint32_t log_of,
                              uint32_t log_ml, uint4* rec, uint32_t nseq, uint32_t regen, uint32_t opos0, uint32_t fcs, uint32_t (&rep)[3],
                              uint64_t* out_end)
{
    if (bsn == 0) return 101u;
    const uint32_t top = ((ref_gcu8*)bs)[bsn - 1];
    if (top == 0) return 102u;
    ref_gst16* grec = (ref_gst16*)rec;
    const uint32_t total_bits = 8u * bsn;
    // dwords k .. k + 3 of the stream, counted from its end: dword k = bytes [bsn - 4 (k + 1), bsn - 4 k), little-endian, so that its
    // top bit is the first of them to be read (.w = dword k).  Bytes below the stream's first belong to the arena (or, below the
    // arena, read as zero) and are never consumed: a walk that reaches them has left the stream and stops.
    auto quad = [&](uint32_t k) -> u32x4 {
        const int64_t at = (int64_t)bsn - 4 * (int64_t)k - 16;
        u32x4 v = { 0u, 0u, 0u, 0u };
        if (at + (int64_t)arena_lo >= 0) {
            v = *(ref_gld16*)(bs + at);
        } else {  // the first bytes of the arena: whatever of the 16 bytes is there
            for (int j = 0; j < 16; ++j) {
                if (at + j + (int64_t)arena_lo < 0) continue;
                const uint32_t byte = (uint32_t)((ref_gcu8*)bs)[at + j] << (8 * (j & 3));
                v.x |= (j >> 2) == 0 ? byte : 0u;
                v.y |= (j >> 2) == 1 ? byte : 0u;
                v.z |= (j >> 2) == 2 ? byte : 0u;
                v.w |= (j >> 2) == 3 ? byte : 0u;
            }
        }
        return v;
    };
    auto ring_put = [&](uint32_t k, u32x4 v) {  // dwords k .. k + 3 (k % 4 == 0) into their slots
        const uint32_t s0 = k & (uint32_t)(REF_RING - 1);
        S.u.ring[s0][l] = v.w;
        S.u.ring[s0 + 1][l] = v.z;
        S.u.ring[s0 + 2][l] = v.y;
        S.u.ring[s0 + 3][l] = v.x;
        if (s0 == 0) {  // (a read of four consecutive slots may start at slot 63)
            S.u.ring[REF_RING][l] = v.w;
            S.u.ring[REF_RING + 1][l] = v.z;
            S.u.ring[REF_RING + 2][l] = v.y;
            S.u.ring[REF_RING + 3][l] = v.x;
        }
    };
    // the ring holds dwords [base, base + 64); pend = dwords [base + 64, base + 96), on their way
    u32x4 pend[8];
    for (uint32_t h = 0; h < 64u; h += 32u) {
#pragma unroll
        for (int q = 0; q < 8; ++q) pend[q] = quad(h + 4u * (uint32_t)q);
#pragma unroll
        for (int q = 0; q < 8; ++q) ring_put(h + 4u * (uint32_t)q, pend[q]);
    }
    uint32_t base = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) pend[q] = quad(64u + 4u * (uint32_t)q);
    uint32_t pos = 8u - (uint32_t)hbit(top);  // bits of the stream that are consumed
    uint32_t d0 = S.u.ring[0][l], d1 = S.u.ring[1][l], d2 = S.u.ring[2][l], d3 = S.u.ring[3][l];
    uint32_t sl, so, sm;
    {
        const uint64_t w = top64(((uint64_t)d0 << 32) | d1, ((uint64_t)d2 << 32) | d3, pos);  // log_ll + log_of + log_ml <= 26 bits
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
    uint32_t bl = get_bn(S, G, l, T_LL + sl), bo = get_bn(S, G, l, T_OF + so), bm = get_bn(S, G, l, T_ML + sm);
    uint32_t cl = get_cx(S, G, l, T_LL + sl), co = get_cx(S, G, l, T_OF + so), cm = get_cx(S, G, l, T_ML + sm);
    {
        const uint32_t j = pos >> 5;
        d0 = S.u.ring[j][l];
        d1 = S.u.ring[j + 1][l];
        d2 = S.u.ring[j + 2][l];
        d3 = S.u.ring[j + 3][l];
    }
    for (uint32_t i = 0; i < nseq; ++i) {
        const uint32_t xl = (bl >> 13) | ((cl >> 6) << 3), xo = (bo >> 13) | ((co >> 6) << 3), xm = (bm >> 13) | ((cm >> 6) << 3);
        const uint32_t xt = xo + xm + xl;  // <= 31 + 16 + 16: offset bits first, then match length, then literal length
        const bool lastseq = i + 1 == nseq;
        const uint32_t nl = lastseq ? 0u : (bl >> 9) & 15u, nm = lastseq ? 0u : (bm >> 9) & 15u, no = lastseq ? 0u : (bo >> 9) & 15u;
        const uint32_t nt = nl + nm + no;  // <= 26: the states move on in the order LL, ML, OF
        const uint64_t A = ((uint64_t)d0 << 32) | d1, B = ((uint64_t)d2 << 32) | d3;
        const uint32_t sh = pos & 31u;
        const uint64_t ws = top64(A, B, sh + xt);  // sh + xt < 95
        const uint32_t v = nt ? (uint32_t)(ws >> (64u - nt)) : 0u;
        sl = (bl & 511u) + (v >> (nm + no));
        sm = (bm & 511u) + ((v >> no) & ((1u << nm) - 1u));
        so = (bo & 511u) + (v & ((1u << no) - 1u));
        pos += xt + nt;  // (a walk that leaves the stream goes on over whatever lies below it -- the ring and the loads stay inside
                         // the arena, the records inside the claim -- and is refused behind the loop: pos only grows)
        const uint32_t j = pos >> 5;
        if (j >= base + 32u) {  // half the ring is behind: what was requested when the walk entered the other half takes its place
            const uint32_t k0 = base + 64u;
#pragma unroll
            for (int q = 0; q < 8; ++q) ring_put(k0 + 4u * (uint32_t)q, pend[q]);
            base += 32u;
#pragma unroll
            for (int q = 0; q < 8; ++q) pend[q] = quad(base + 64u + 4u * (uint32_t)q);
        }
        const uint32_t kl = cl & 63u, ko = co & 63u, km = cm & 63u;
        // the next sequence's entries and bits: on their way while this one is finished
        bl = get_bn(S, G, l, T_LL + sl);
        bo = get_bn(S, G, l, T_OF + so);
        bm = get_bn(S, G, l, T_ML + sm);
        cl = get_cx(S, G, l, T_LL + sl);
        co = get_cx(S, G, l, T_OF + so);
        cm = get_cx(S, G, l, T_ML + sm);
        const uint32_t js = j & (uint32_t)(REF_RING - 1);
        d0 = S.u.ring[js][l];
        d1 = S.u.ring[js + 1][l];
        d2 = S.u.ring[js + 2][l];
        d3 = S.u.ring[js + 3][l];
        const uint64_t wx = top64(A, B, sh);
        const uint64_t X = xt ? wx >> (64u - xt) : 0ull;
        const uint32_t ofv = (1u << (ko & 31u)) + (uint32_t)(X >> (xm + xl));
        const uint32_t mlen = S.mlb[km] + ((uint32_t)(X >> xl) & ((1u << xm) - 1u));
        const uint32_t llen = S.llb[kl] + ((uint32_t)X & ((1u << xl) - 1u));
        // repeat offsets (RFC 8878 3.1.1.5): idx 0 = rep0 as it is, 1 = rep1, 2 = rep2, 3 = rep0 - 1
        // (written with masks, not with conditions: as conditions they come out as a chain of branches)
        const uint32_t isrep = 0u - (uint32_t)(ofv <= 3);  // all ones / zero
        const uint32_t idx = ofv - 1 + (uint32_t)(llen == 0);
        const uint32_t is1 = 0u - (uint32_t)(idx == 1), is2 = 0u - (uint32_t)(idx == 2);
        uint32_t cand = ((rep1 & is1) | (rep2 & is2) | (rep0 & ~(is1 | is2))) - (uint32_t)(idx == 3);
        cand += (uint32_t)(cand == 0);  // libzstd forces an invalid 0 to 1
        const uint32_t offset = (cand & isrep) | ((ofv - 3) & ~isrep);
        const uint32_t keep2 = isrep & (0u - (uint32_t)(idx <= 1)), keep1 = isrep & (0u - (uint32_t)(idx == 0));
        rep2 = (rep2 & keep2) | (rep1 & ~keep2);
        rep1 = (rep1 & keep1) | (rep0 & ~keep1);
        rep0 = offset;
        sum_ll += llen;
        outp += llen;
        astray = astray | (offset > outp);
        outp += mlen;
        {
            u32x4 rv = { llen, mlen, offset, 0u };
            grec[i] = rv;
        }
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

// (why a frame was left alone: RefPre.pad[0], a diagnostic that vbz_gpu_decode_paths prints under VBZ_HIP_TRACE)
#define BAIL(k) do { P->pad[0] = (k); return false; } while (0)
template <int FPW, bool INLDS>
__device__ bool ref_frame(RefLds<FPW, INLDS>& S, RefGlobal G, int l, const ReadBatch& b, uint32_t r, RefPre* P, uint4* recs, uint64_t recs_cap,
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
                const unsigned long long t0 = __builtin_readcyclecounter();
                // stage the header of the sequences section: 16 bytes per load, all of them in flight at once
                const uint32_t hn = sqn < (uint32_t)REF_HDR ? sqn : (uint32_t)REF_HDR;
                for (uint32_t k = 0; 16 * k < hn; ++k) {
                    const uint64_t a = rld64(sq + 16 * k), c = rld64(sq + 16 * k + 8);
                    S.u.p.hdr[4 * k][l] = (uint32_t)a;
                    S.u.p.hdr[4 * k + 1][l] = (uint32_t)(a >> 32);
                    S.u.p.hdr[4 * k + 2][l] = (uint32_t)c;
                    S.u.p.hdr[4 * k + 3][l] = (uint32_t)(c >> 32);
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
                int u = ref_seq_table(S, G, l, T_LL, &log_ll, &have_ll, (modes >> 6) & 3, used, (int)(hn - used), 0);
                if (u < 0) BAIL(23);
                used += (uint32_t)u;
                // predefined length tables and offsets that are all "repeat offset 1" (an RLE table of code 0): what zstd_encode.hip
                // writes for its zero-run blocks (a frame of this library that the batched decoder did not take: raw blocks, say),
                // with checkpoints behind the frame that let the one-wavefront decoder walk the chain in parallel segments; left to it.
                // (libzstd's frames of nanopore signal have the same offsets -- its matches are the zero runs of the control bytes --
                // but FSE-coded length tables of 9 bits, which that path does not take: they are walked here)
                if (modes == 0x10u && used < hn && hdr_byte(S, l, used) == 0) BAIL(22);
                u = ref_seq_table(S, G, l, T_OF, &log_of, &have_of, (modes >> 4) & 3, used, (int)(hn - used), 1);
                if (u < 0) BAIL(24);
                used += (uint32_t)u;
                u = ref_seq_table(S, G, l, T_ML, &log_ml, &have_ml, (modes >> 2) & 3, used, (int)(hn - used), 2);
                if (u < 0) BAIL(25);
                used += (uint32_t)u;
                const unsigned long long t1 = __builtin_readcyclecounter();
                if (used >= hn) BAIL(26);  // (hn < sqn: a header longer than the staged bytes is left to the careful decoder)
                // a frame may claim 16 bytes of records per 16 bytes of its content (libzstd on nanopore signal: one sequence per ~110
                // bytes): the claims of a call then fit a workspace of the size of the call's content whatever the frames are
                frame_ns += ns;
                if (16ull * frame_ns > fcs) BAIL(31);
                const unsigned long long first = atomicAdd(recs_used, (unsigned long long)ns);
                if (first + ns > recs_cap) BAIL(27);  // (a workspace capped below the content: nothing behind this claim fits either)
                uint64_t end = 0;
                const uint32_t why = ref_chain(S, G, l, sq + used, sqn - used, src_off + pos + lit_end + used, (uint32_t)log_ll, (uint32_t)log_of,
                                               (uint32_t)log_ml, recs + first, ns, regen, opos, fcs, rep, &end);
                if (why) BAIL(why);
                const unsigned long long t2 = __builtin_readcyclecounter();
                P->pad[1] = (uint32_t)(t1 - t0);   // (diagnostics: cycles of the tables and of the chain of the frame's last block)
                P->pad[2] = (uint32_t)(t2 - t1);
                P->pad[3] = ns;
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

// Lane `slot` takes frames slot, slot + slots, ...; INLDS: REF_GRID wavefronts of REF_FPW frames, tables in LDS; else wavefronts of
// REF_FPW_MEM frames with the lane's tables at tables + slot * REF_TAB_BYTES.
template <int FPW, bool INLDS>
__global__ __launch_bounds__(WAVE) void ref_chain_kernel(ReadBatch b, const uint32_t* redo, RefPre* pre, uint4* recs, uint64_t recs_cap,
                                                         unsigned long long* recs_used, uint8_t* tables)
{
    __shared__ RefLds<FPW, INLDS> S;
    const int l = threadIdx.x;
    // a few hundred wavefronts whose latency the whole call waits for, beside thousands that only need throughput (the literals of the
    // same frames, ref_pieces_kernel): theirs is the issue slot when both want it
#ifndef VBZ_REF_NOPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    if (l < 36) S.llb[l] = LL_BASE[l];
    if (l < 53) S.mlb[l] = ML_BASE[l];
    __syncthreads();
    if (l >= FPW) return;
    const uint32_t slots = gridDim.x * (uint32_t)FPW, slot = blockIdx.x * (uint32_t)FPW + (uint32_t)l;
    RefGlobal G = { nullptr, nullptr };
    if constexpr (!INLDS) {
        uint8_t* t = tables + (size_t)slot * REF_TAB_BYTES;
        G.bn = (__attribute__((address_space(1))) uint16_t*)t;
        G.cx = (__attribute__((address_space(1))) uint8_t*)(t + 2 * T_ALL);
    }
    for (uint32_t r = slot; r < b.n_reads; r += slots) {
        RefPre* P = pre + r;
        bool ok = false;
        if (redo[r]) ok = ref_frame<FPW, INLDS>(S, G, l, b, r, P, recs, recs_cap, recs_used);
        P->ok = ok ? 1u : 0u;
    }
}

// which way a call goes: a batch that fits one round of the LDS kernel takes it (a round = the chain's latency with an LDS round
// trip per sequence); a larger batch keeps its tables in memory, every frame in flight at once
__host__ bool ref_in_lds(uint32_t n_reads)
{
    static const int force = [] {
        const char* e = getenv("VBZ_HIP_REF_TABLES");  // lds | mem (measurements)
        return e ? (e[0] == 'l' ? 1 : (e[0] == 'm' ? 2 : 0)) : 0;
    }();
    return force ? force == 1 : n_reads <= (uint32_t)(REF_GRID * REF_FPW);
}
__host__ uint32_t ref_mem_waves(uint32_t n_reads)
{
    const uint32_t w = (n_reads + REF_FPW_MEM - 1) / REF_FPW_MEM;
    return w < REF_MEM_WAVES_MAX ? w : REF_MEM_WAVES_MAX;
}

}  // namespace

size_t zstd_ref_pre_bytes(uint32_t n_reads) { return (size_t)n_reads * sizeof(RefPre) + 64; }
const RefPre* zstd_ref_pre(const void* pre_meta) { return reinterpret_cast<const RefPre*>(reinterpret_cast<const uint8_t*>(pre_meta) + 64); }
size_t zstd_ref_table_bytes(uint32_t n_reads) { return ref_in_lds(n_reads) ? 0 : (size_t)ref_mem_waves(n_reads) * REF_FPW_MEM * REF_TAB_BYTES; }

hipError_t launch_zstd_ref_chain(const ReadBatch& b, const uint32_t* redo, void* pre_meta, void* tables, void* recs, uint64_t recs_cap, RefChains* out,
                                 hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    uint8_t* m = reinterpret_cast<uint8_t*>(pre_meta);
    unsigned long long* used = reinterpret_cast<unsigned long long*>(m);
    RefPre* pre = reinterpret_cast<RefPre*>(m + 64);
    out->pre = pre;
    out->recs = recs;
    hipError_t e = hipMemsetAsync(used, 0, 8, s);
    if (e != hipSuccess) return e;
    static_assert(sizeof(RefLds<REF_FPW, true>) * 4 <= 160 * 1024, "four wavefronts' tables per CU");
    if (ref_in_lds(b.n_reads))
        hipLaunchKernelGGL((ref_chain_kernel<REF_FPW, true>), dim3(REF_GRID), dim3(WAVE), 0, s, b, redo, pre, reinterpret_cast<uint4*>(recs), recs_cap, used,
                           (uint8_t*)nullptr);
    else
        hipLaunchKernelGGL((ref_chain_kernel<REF_FPW_MEM, false>), dim3(ref_mem_waves(b.n_reads)), dim3(WAVE), 0, s, b, redo, pre,
                           reinterpret_cast<uint4*>(recs), recs_cap, used, reinterpret_cast<uint8_t*>(tables));
    return hipGetLastError();
}

}  // namespace vbzhip
