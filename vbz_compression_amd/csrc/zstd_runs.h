// zstd_runs.h -- the zero-run blocks of a frame (zstd_encode.hip codes runs of a byte as sequences "copy the byte in front", offset =
// repeat offset 1 = 1): the walk over the two FSE state machines (serial, or in parallel segments from the encoder's checkpoints) and
// the placement of literals and runs from prefix sums.  Shared by the one-wavefront decoder (zstd_decode.hip) and the batched
// own-frame decoder (zstd_decode_fast.hip); every function works on LDS it is handed, one wavefront per frame.
#pragma once
#include "vbz_kernels.h"

namespace vbzhip {
namespace {

constexpr int WAVE = 64;
constexpr uint32_t BLOCK_MAX = 128u << 10;

// RFC 8878 3.1.1.3.2.2 / 3.1.1.3.2.1.1: predefined distributions of the sequence codes, code -> baseline and extra bits
__device__ const int16_t LL_DEFAULT[36] = { 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                            2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1 };
__device__ const int16_t ML_DEFAULT[53] = { 1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1 };
__device__ const int16_t OF_DEFAULT[29] = { 1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1 };
__device__ const uint32_t LL_BASE[36] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                          20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536 };
__device__ const uint8_t LL_BITS[36] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                         1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };
__device__ const uint32_t ML_BASE[53] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                          21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                          43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539 };
__device__ const uint8_t ML_BITS[53] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };

__device__ __forceinline__ int hbit(uint32_t v) { return 31 - __clz((int)v); }
// wave-uniform helpers: values the compiler keeps in scalar registers, and reads of one lane of a vector register
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// all lanes.  Second half of the zero-run fast path: (literal length, match length) pairs are known, every
// match copies the byte in front of it.  Positions come from wave prefix sums, so all sequences of a chunk of 64
// are placed at once: a lane copies its literals and fills its run; runs of 64+ bytes are filled by the wave.
// Returns the output position behind the block, or 0xFFFFFFFF if the pairs do not fit the block.
__device__ __noinline__ uint32_t place_zero_runs(uint8_t* dst, const uint2* pairs, uint32_t nseq, const uint8_t* litp, uint32_t ltype,
                                    uint32_t regen, uint32_t opos, uint32_t fcs, uint32_t block_max, uint8_t* lds_lit,
                                    uint32_t lds_cap, int lane)
{
    const uint8_t rle_byte = ltype == 1 ? litp[0] : 0;
    uint32_t lposw = 0, oposw = opos;
    uint2 pnext = (uint32_t)lane < nseq ? pairs[lane] : make_uint2(0u, 0u);  // one chunk of pairs is always in flight
    for (uint32_t base = 0; base < nseq; base += WAVE) {
        const uint32_t ll = pnext.x, ml = pnext.y;
        {
            const uint32_t i1 = base + WAVE + (uint32_t)lane;
            pnext = i1 < nseq ? pairs[i1] : make_uint2(0u, 0u);
        }
        const uint32_t il = wave_incl_scan_u32(ll), it = wave_incl_scan_u32(ll + ml);
        const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)il, 63), tt = (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
        if ((uint64_t)lposw + tl > regen || (uint64_t)oposw + tt > fcs) return 0xFFFFFFFFu;
        // Common case (zero runs, a chunk's literals and output fit the LDS area): build the chunk's output in LDS --
        // zero fill, every lane drops its literals in place, runs of a non-zero byte are written out -- and copy it
        // to memory with 16-byte stores.  Otherwise the lanes write to memory directly.
        const uint32_t lit_room = (tl + 15u) & ~15u;
        if (ltype != 1 && 5u * lit_room + tt + 16u <= lds_cap) {
            // LDS: the chunk's literals, one dword per literal (to become its shift), the chunk's output
            uint32_t* lds_sh = reinterpret_cast<uint32_t*>(lds_lit + lit_room);
            uint8_t* lds_out = lds_lit + 5u * lit_room;
            wave_lds_sync();  // one wave: LDS hand-over only, global accesses stay in flight
            for (uint32_t j = lane; 4 * j < tl; j += WAVE) {
                uint32_t v;
                __builtin_memcpy(&v, litp + lposw + 4 * j, 4);  // may read 3 bytes past the literals (staging slack)
                reinterpret_cast<uint32_t*>(lds_lit)[j] = v;
            }
            for (uint32_t j = 4u * (uint32_t)lane; j < lit_room; j += 4u * WAVE) *reinterpret_cast<uint4*>(lds_sh + j) = make_uint4(0u, 0u, 0u, 0u);
            for (uint32_t j = 16u * (uint32_t)lane; j < tt; j += 16u * WAVE) *reinterpret_cast<uint4*>(lds_out + j) = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_sync();
            // a literal lands (sum of the match lengths in front of it) further down than it sits in the literal
            // stream: mark the first literal of every sequence with the match length before it, prefix-sum the marks
            // over the literals -- every lane then moves the same number of literals, however they are spread
            const uint32_t lo = il - ll, oo = it - (ll + ml);
            const uint32_t mprev = wave_prev_lane_u32(ml);
            if (ll != 0 && lane != 0) lds_sh[lo] = mprev;
            wave_lds_sync();
            uint32_t carry = 0;
            for (uint32_t j0 = 0; j0 < tl; j0 += 8u * WAVE) {
                const uint32_t j = j0 + 8u * (uint32_t)lane;
                uint32_t sh[8], run = 0;
                const uint4 a0 = j < lit_room ? *reinterpret_cast<const uint4*>(lds_sh + j) : make_uint4(0u, 0u, 0u, 0u);
                const uint4 a1 = j + 4 < lit_room ? *reinterpret_cast<const uint4*>(lds_sh + j + 4) : make_uint4(0u, 0u, 0u, 0u);
                const uint32_t dl[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    run += dl[k];
                    sh[k] = run;
                }
                const uint32_t inc = wave_incl_scan_u32(run);
                const uint32_t before = carry + inc - run;
                uint2 lb = make_uint2(0u, 0u);
                if (j < lit_room) lb = *reinterpret_cast<const uint2*>(lds_lit + j);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (j + k < tl) lds_out[j + k + before + sh[k]] = (uint8_t)((k < 4 ? lb.x : lb.y) >> (8 * (k & 3)));
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
            wave_lds_sync();
            if (ll != 0) {  // a run of a non-zero byte (never what zstd_encode.hip writes)
                const uint8_t lastb = lds_lit[il - 1];
                if (lastb != 0)
                    for (uint32_t k = 0; k < ml; ++k) lds_out[oo + ll + k] = lastb;
            }
            wave_lds_sync();
            uint8_t* g = dst + oposw;
            for (uint32_t j = 16u * (uint32_t)lane; j < tt; j += 16u * WAVE) {
                if (j + 16u <= tt) {
                    const uint4 v = *reinterpret_cast<const uint4*>(lds_out + j);
                    __builtin_memcpy(g + j, &v, 16);
                } else {
                    for (uint32_t k = j; k < tt; ++k) g[k] = lds_out[k];
                }
            }
            lposw += tl;
            oposw += tt;
            continue;
        }
        const uint32_t my_lit = lposw + il - ll;
        uint32_t my_out = oposw + it - (ll + ml);
        uint8_t lastb = rle_byte;
        if (ltype != 1 && tl <= lds_cap) {
            // the chunk's literals are contiguous: one coalesced copy into LDS, then short per-lane loops
            __syncthreads();
            for (uint32_t j = lane; 4 * j < tl; j += WAVE) {
                uint32_t v;
                __builtin_memcpy(&v, litp + lposw + 4 * j, 4);  // may read 3 bytes past the literals (staging slack)
                reinterpret_cast<uint32_t*>(lds_lit)[j] = v;
            }
            __syncthreads();
            const uint32_t lo = il - ll;
            for (uint32_t k = 0; k < ll; ++k) {
                lastb = lds_lit[lo + k];
                dst[my_out + k] = lastb;
            }
        } else {
            for (uint32_t k = 0; k < ll; ++k) {
                if (ltype != 1) lastb = litp[my_lit + k];
                dst[my_out + k] = lastb;
            }
        }
        my_out += ll;
        if (ml < 64) {  // unaligned dword stores are fine in global memory
            const uint32_t v4 = (uint32_t)lastb * 0x01010101u;
            uint32_t k = 0;
            for (; k + 4 <= ml; k += 4) __builtin_memcpy(dst + my_out + k, &v4, 4);
            for (; k < ml; ++k) dst[my_out + k] = lastb;
        }
        uint64_t big = __ballot(ml >= 64);
        while (big) {
            const int sl_ = __ffsll((long long)big) - 1;
            big &= big - 1;
            // (the lane comes out of a ballot: a scalar -- v_readlane, not a round trip through the LDS crossbar per value)
            const uint32_t bo = (uint32_t)__builtin_amdgcn_readlane((int)my_out, sl_);
            const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)ml, sl_);
            const uint32_t bv = (uint32_t)__builtin_amdgcn_readlane((int)lastb, sl_);
            for (uint32_t k = lane; k < bl; k += WAVE) dst[bo + k] = (uint8_t)bv;
        }
        lposw += tl;
        oposw += tt;
    }
    const uint32_t rest = regen - lposw;
    if ((uint64_t)oposw + rest > fcs) return 0xFFFFFFFFu;
    if (ltype == 1) {
        for (uint32_t k = lane; k < rest; k += WAVE) dst[oposw + k] = rle_byte;
    } else {
        for (uint32_t k = lane; k < rest; k += WAVE) dst[oposw + k] = litp[lposw + k];  // literals never alias the output here
    }
    oposw += rest;
    if (oposw - opos > BLOCK_MAX || oposw - opos > block_max) return 0xFFFFFFFFu;
    return oposw;
}

// Decoding tables of the predefined LL / ML distributions in the compact form the zero-run chain uses:
// entry = { base value, next-state base | extra bits << 16 | state bits << 24 }, one entry per lane.
struct SeqDTables
{
    uint2 ll[64], ml[64];
};


// all lanes.  First half of the zero-run fast path: walk the LL and ML state machines of a block whose offsets are
// all "repeat offset 1" and write (literal length, match length) pairs to `ws`.
// The walk is one dependent chain, so it is written as wave-uniform code that the compiler keeps on the scalar
// unit, with everything it looks up held across the lanes of vector registers: lane j of llt / mlt is table entry
// j (accuracy logs <= 6), lane j of `win` is the j-th dword of the current 256-byte window of the (backward) bit
// stream, lane (i & 63) of pl / pm collects pair i until 64 of them leave with one store.  No LDS, no memory
// latency on the chain except one window load per 2048 bits.
// Returns 0 = corrupt, 1 = pairs written (*total_out = bytes the block regenerates), 2 = not such a block after
// all (the caller decodes the frame again, in order).
__device__ __noinline__ uint32_t zero_run_chain(const uint8_t* bs_, uint32_t bsn_, uint2* ws, uint32_t nseq_, uint2 llt, uint2 mlt,
                                                uint32_t log_ll_, uint32_t log_ml_, uint32_t regen_, int lane, uint32_t* total_out)
{
    const uint32_t bsn = uni(bsn_), nseq = uni(nseq_), log_ll = uni(log_ll_), log_ml = uni(log_ml_), regen = uni(regen_);
    const uint8_t* bs = reinterpret_cast<const uint8_t*>(((uint64_t)uni((uint32_t)((uint64_t)bs_ >> 32)) << 32) |
                                                         uni((uint32_t)(uint64_t)bs_));
    if (bsn == 0) return 0u;
    uint32_t k0 = 0;  // the window holds dwords k0 .. k0+63, counted from the end of the stream
    auto load_window = [&]() -> uint32_t {
        const int64_t off = (int64_t)bsn - 4 * (int64_t)(k0 + (uint32_t)lane + 1);
        uint32_t v = 0;
        if (off >= 0) {
            __builtin_memcpy(&v, bs + off, 4);
        } else if (off > -4) {  // the first bytes of the stream: fewer than four are left
            for (int k = 0; k < 4 + (int)off; ++k) v |= (uint32_t)bs[k] << (8 * (k - (int)off));
        }
        return v;  // zeros before the start of the stream (bits_left tells real bits from padding)
    };
    uint32_t win = load_window();
    const uint32_t top = lane_get(win, 0) >> 24;  // the last byte carries the end mark
    if (top == 0) return 0u;
    const uint32_t hb = (uint32_t)hbit(top);
    int64_t bits_left = (int64_t)(bsn - 1) * 8 + hb;
    uint64_t buf = 0;      // unread bits, left aligned
    uint32_t have = 0, q = 0;
    auto refill = [&]() {  // afterwards have > 32
        if (have <= 32) {
            if (q - k0 == 64) {
                k0 += 64;
                win = load_window();
            }
            const uint32_t d = lane_get(win, q - k0);
            ++q;
            buf |= (uint64_t)d << (32 - have);
            have += 32;
        }
    };
    auto take = [&](uint32_t nb) -> uint32_t {  // nb <= 32 <= have
        const uint32_t v = (uint32_t)((buf >> 1) >> (63 - nb));
        buf <<= nb;
        have -= nb;
        bits_left -= nb;
        return v;
    };
    refill();
    buf <<= 8 - hb;  // padding and end mark
    have -= 8 - hb;
    refill();
    uint32_t sl = take(log_ll);
    uint32_t sm = take(log_ml);  // the offset state has no bits (RLE table)
    uint64_t sum_ll = 0, sum_all = 0;
    uint32_t pl = 0, pm = 0;
    for (uint32_t si = 0; si < nseq; ++si) {
        const uint32_t elx = lane_get(llt.x, sl), ely = lane_get(llt.y, sl);
        const uint32_t emx = lane_get(mlt.x, sm), emy = lane_get(mlt.y, sm);
        refill();
        const uint32_t mlen = emx + take((emy >> 16) & 0xFF);
        refill();
        const uint32_t llen = elx + take((ely >> 16) & 0xFF);
        if (llen == 0) return 2u;  // repeat-offset semantics change: decode in order
        const bool mine = (uint32_t)lane == (si & 63);
        pl = mine ? llen : pl;
        pm = mine ? mlen : pm;
        sum_ll += llen;
        sum_all += (uint64_t)llen + mlen;
        if (si + 1 < nseq) {
            refill();
            sl = (ely & 0xFFFF) + take(ely >> 24);
            sm = (emy & 0xFFFF) + take(emy >> 24);
        }
        if ((si & 63) == 63) ws[(si & ~63u) + (uint32_t)lane] = make_uint2(pl, pm);
    }
    if ((uint32_t)lane < (nseq & 63)) ws[(nseq & ~63u) + (uint32_t)lane] = make_uint2(pl, pm);
    if (bits_left != 0) return 0u;  // every bit must be consumed, none beyond
    if (sum_ll > regen) return 0u;
    if (sum_all + (regen - sum_ll) > BLOCK_MAX) return 0u;
    *total_out = (uint32_t)(sum_all + (regen - sum_ll));
    return 1u;
}

// all lanes.  The same walk as zero_run_chain, split at the encoder's checkpoints (zstd_encode.hip, CP_MAGIC): lane
// j decodes sequences [j * spacing, (j + 1) * spacing) from (unread bits, LL state, ML state) = checkpoint j - 1
// (lane 0: from the top of the stream).  The bit stream is staged in LDS (`lds`, lds_bytes of it), table entries come from the lanes that hold them (ds_bpermute).  The result is accepted only if every
// segment ends exactly where the next one started and the last one consumes the stream: then it is the serial
// walk.  Returns 1 = pairs written, 2 = not a pure zero-run block, 3 = checkpoints unusable (walk serially).
__device__ __noinline__ uint32_t zero_run_chain_segments(const uint8_t* bs, uint32_t bsn, uint2* ws, uint32_t nseq, uint2 llt, uint2 mlt,
                                                         uint32_t regen, const uint8_t* cp, uint32_t ncp, uint32_t spacing, int lane,
                                                         uint32_t* total_out, uint32_t* lds, uint32_t lds_bytes)
{
    const uint32_t CAP = lds_bytes - 16u;  // the bit stream is staged in `lds`
    if (bsn == 0 || bsn > CAP || ncp + 1 > (uint32_t)WAVE || spacing == 0) return 3u;
    if ((uint64_t)(ncp + 1) * spacing < nseq || (uint64_t)ncp * spacing >= nseq) return 3u;
    __syncthreads();
    for (uint32_t i = lane; 4 * i < bsn + 8; i += WAVE) {
        uint32_t v = 0;
        if (4 * i + 4 <= bsn) __builtin_memcpy(&v, bs + 4 * i, 4);
        else for (uint32_t k = 0; 4 * i + k < bsn && k < 4; ++k) v |= (uint32_t)bs[4 * i + k] << (8 * k);
        lds[i] = v;
    }
    __syncthreads();
    auto extract = [&](uint32_t pos, uint32_t nb) -> uint32_t {  // bits [pos, pos + nb) of the stream, nb <= 32
        const uint32_t d0 = lds[pos >> 5], d1 = lds[(pos >> 5) + 1];
        const uint32_t x = __builtin_amdgcn_alignbit(d1, d0, pos & 31);
        return nb >= 32 ? x : (x & ((1u << nb) - 1u));
    };
    const uint32_t nseg = ncp + 1;
    const bool active = (uint32_t)lane < nseg;
    uint32_t P = 0, sl = 0, sm = 0, bad = 0, impure = 0;
    if (lane == 0) {
        const uint32_t top = lds[(bsn - 1) >> 2] >> (8 * ((bsn - 1) & 3)) & 0xFF;
        if (top == 0) bad = 1;
        else {
            P = (bsn - 1) * 8 + (uint32_t)hbit(top);
            if (P < 12) bad = 1;
            else {
                sl = extract(P - 6, 6);
                sm = extract(P - 12, 6);
                P -= 12;
            }
        }
    } else if (active) {
        uint32_t w;
        __builtin_memcpy(&w, cp + 4 * (lane - 1), 4);
        P = w & 0xFFFFFu;
        sl = (w >> 20) & 63u;
        sm = w >> 26;
        if (P > 8 * bsn) bad = 1;
    }
    if (__any(bad)) return (uint32_t)__builtin_amdgcn_readlane((int)bad, 0) ? 0u : 3u;  // a bad end mark is the frame's fault
    const uint32_t P0 = P, sl0 = sl, sm0 = sm;
    const uint32_t first = (uint32_t)lane * spacing;
    uint32_t sum_ll = 0, sum_all = 0;
    for (uint32_t s = 0; s < spacing; ++s) {
        const uint32_t i = first + s;
        const bool on = active && i < nseq && !bad;
        const uint32_t elx = (uint32_t)__shfl((int)llt.x, (int)sl, 64), ely = (uint32_t)__shfl((int)llt.y, (int)sl, 64);
        const uint32_t emx = (uint32_t)__shfl((int)mlt.x, (int)sm, 64), emy = (uint32_t)__shfl((int)mlt.y, (int)sm, 64);
        if (on) {
            const uint32_t lnb = (ely >> 16) & 0xFF, mnb = (emy >> 16) & 0xFF;
            const uint32_t eb = lnb + mnb;
            const bool more = i + 1 < nseq;
            const uint32_t snl = more ? ely >> 24 : 0u, snm = more ? emy >> 24 : 0u;
            if (eb + snl + snm > P) {
                bad = 1;
            } else {
                const uint32_t v = eb ? extract(P - eb, eb) : 0u;
                P -= eb;
                const uint32_t mlen = emx + (lnb >= 32 ? 0u : (v >> lnb));
                const uint32_t llen = elx + (lnb >= 32 ? v : (v & ((1u << lnb) - 1u)));
                if (llen == 0) impure = 1;
                ws[i] = make_uint2(llen, mlen);
                sum_ll += llen;
                sum_all += llen + mlen;
                if (sum_all > BLOCK_MAX) bad = 1;
                if (more) {
                    const uint32_t sb = snl + snm;
                    const uint32_t v2 = sb ? extract(P - sb, sb) : 0u;
                    P -= sb;
                    sl = (ely & 0xFFFF) + (v2 >> snm);
                    sm = (emy & 0xFFFF) + (v2 & ((1u << snm) - 1u));
                }
            }
        }
    }
    // every segment must end in the state the next one started from; the last one at the start of the stream
    const uint32_t nP = (uint32_t)__shfl_down((int)P0, 1, 64), nsl = (uint32_t)__shfl_down((int)sl0, 1, 64),
                   nsm = (uint32_t)__shfl_down((int)sm0, 1, 64);
    if (active) {
        if ((uint32_t)lane + 1 < nseg) bad |= (P != nP || sl != nsl || sm != nsm) ? 1u : 0u;
        else bad |= P != 0 ? 1u : 0u;
    }
    if (__any(bad)) return 3u;  // the serial walk decides whether the frame or only the trailer is wrong
    if (__any(impure)) return 2u;
    const uint32_t tll = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(active ? sum_ll : 0u), 63);
    const uint32_t tall = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(active ? sum_all : 0u), 63);
    if (tll > regen) return 0u;
    if ((uint64_t)tall + (regen - tll) > BLOCK_MAX) return 0u;
    *total_out = tall + (regen - tll);
    __syncthreads();
    return 1u;
}

}  // namespace
}  // namespace vbzhip
