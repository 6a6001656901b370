// svb_wave.h -- the int16 zig-zag streamvbyte decoder for ONE wavefront, run by zstd_decode.hip behind the entropy stage.
//
// Replaces, like svb_kernels.hip's svb_decode_kernel<2, true, true> (whose results it reproduces bit for bit, error codes
// included): StreamVByteWorkerV0<int16_t,true>::decompress, vbz/v0/vbz_streamvbyte_impl_sse3.h:468-580.
//
// Why here: the wave that has just decoded a read's zstd frame holds nothing any more -- its registers and LDS are dead --
// and the svb stream it wrote (126 KB for a 100 k-sample read) is still in the L2 / Infinity Cache.  A separate svb_decode
// launch over 65 536 reads finds every one of those streams evicted to HBM again: 8.3 GB written and read back per launch
// for nothing.  So the wave decodes its own stream straight away.
//
// A lone wave (two per SIMD in that kernel) cannot hide memory latency by occupancy the way the 256-thread kernel does, so
// the stream is walked in BLOCKS of 4096 values -- every lane 64 consecutive ones -- through a software pipeline:
//   plan(b+1)   the lane's 16 control bytes of the next block (loaded one block earlier) -> its data bytes, ONE wave scan
//               -> where its data starts, and the block's data extent;
//   fetch(b+1)  the loads for exactly that extent are issued (16 bytes per lane and piece, into registers);
//   decode(b)   from LDS: 32 x (two values' bytes picked up, spread, un-zig-zagged), the delta prefix, ONE wave scan, 128
//               bytes of samples stored per lane;
//   commit(b+1) the fetched pieces go to LDS.
// Loads are in flight for a whole block of work.  A block takes this path when it carries one- and two-byte codes only
// (all an encoder of the reference's kind ever writes for int16: sse3.h:454-463), where the reference's SIMD body and
// scalar tail agree; a stream with three- or four-byte codes goes, from the block that has them on, through
// svb_decode_wave_tiles, the tile loop of svb_kernels.hip on 64 lanes.
#pragma once

#include "vbz_kernels.h"

namespace vbzhip {

// Per-read validation of the svb decoders (the one-workgroup kernel, the segmented kernels, the wave decoder): returns true
// when the read is finished with `res`.
template <int ELEM, bool I16ZZ>
__device__ __forceinline__ bool svb_decode_check(uint32_t in_size, uint32_t out_size, uint32_t& res)
{
    if (out_size % ELEM != 0) {  // vbz/v0/vbz_streamvbyte.cpp:75-78
        res = E_DESTINATION_SIZE;
        return true;
    }
    const uint32_t count = out_size / ELEM;
    const uint32_t keyLen = (count + 3u) >> 2;
    if (I16ZZ) {
        if (count == 0) {  // sse3.h:472-476
            res = 0;
            return true;
        }
        if (in_size < keyLen) {  // sse3.h:478-482
            res = E_INPUT_SIZE;
            return true;
        }
    } else {
        // streamvbyte_validate_stream (vbz/v0/vbz_streamvbyte_impl.h:49-51)
        if (in_size == 0 || count == 0) {
            res = (in_size == count) ? 0u : E_STREAM;
            return true;
        }
        if (keyLen > in_size) {
            res = E_STREAM;
            return true;
        }
    }
    return false;
}

namespace svbwave {

constexpr int WAVE = 64, VPL = 8, TILE = WAVE * VPL;        // 512 values, 128 control bytes per tile
constexpr int BLOCK_TILES = 8, BLOCK = TILE * BLOCK_TILES;  // 4096 values, 1 KB of control bytes per block
constexpr int PIECES = 9;                                    // 16-byte pieces per lane that cover a block's data: 9 x 1 KB >= 8192 + 15 + 15
constexpr uint32_t DATABUF = PIECES * 1024;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1), aligned(1))) const u32x4 gld16;
typedef __attribute__((address_space(1), aligned(1))) const uint8_t gld1;
// (every access names global memory: a FLAT store -- what a plain pointer gives -- counts on the LDS counter as well, and
// the wave would sit out every store's round trip at its next LDS read)
typedef __attribute__((address_space(1), aligned(1))) u32x4 gst16;
typedef __attribute__((address_space(1), aligned(1))) uint16_t gst2;

__device__ __forceinline__ uint32_t wave_total_u32(uint32_t incl) { return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63); }

// The tile loop of svb_kernels.hip (svb_decode_range, MODE 0) on one wave: values [first, end) of a stream of `count`
// values, data bytes from data[pos], delta chain at `run`.  Returns false when the stream is shorter than its control
// bytes claim.  stage: TILE * 4 + 48 bytes of LDS.
__device__ __noinline__ bool svb_decode_wave_tiles(const uint8_t* in, const uint8_t* data, uint32_t dataBytes, uint32_t count, uint32_t first,
                                                   uint32_t end, uint64_t& pos_io, uint32_t& run_io, uint8_t* out, uint8_t* stage, int lane)
{
    const bool out_aligned = (((uintptr_t)out) & 15u) == 0;
    const uint32_t* stage32 = reinterpret_cast<const uint32_t*>(stage);
    uint64_t pos = pos_io;
    uint32_t run = run_io;
    bool good = true;
    for (uint32_t t0 = first; t0 < end; t0 += TILE) {
        const uint32_t i0 = t0 + (uint32_t)lane * VPL;
        const int valid = i0 >= end ? 0 : (end - i0 >= (uint32_t)VPL ? VPL : (int)(end - i0));
        uint32_t keybits = 0;
        if (valid > 0) {
            gld1* kp = (gld1*)in + (i0 >> 2);
            keybits = kp[0];
            if (valid > 4) keybits |= (uint32_t)kp[1] << 8;
        }
        const uint32_t kb = valid == VPL ? keybits : (keybits & ((1u << (2 * valid)) - 1u));
        const uint32_t L = (uint32_t)valid + (uint32_t)__popc(kb & 0x5555u) + 2u * (uint32_t)__popc(kb & 0xAAAAu);
        const uint32_t incl = wave_incl_scan_u32(L);
        const uint32_t ex = incl - L, tot = wave_total_u32(incl);
        if (pos + tot > dataBytes) {  // stream shorter than its control bytes claim
            good = false;
            break;
        }
        const uint8_t* g0 = data + pos;
        const uint32_t mis = (uint32_t)((uintptr_t)g0 & 15u);
        const uint8_t* ga = g0 - mis;
        const uint32_t nch = (mis + tot + 15u) >> 4;
        wave_lds_sync();  // the tile before has been read
        for (uint32_t c = lane; c < nch; c += WAVE) *reinterpret_cast<u32x4*>(stage + 16u * c) = *(gld16*)(ga + 16ull * c);
        wave_lds_sync();
        uint32_t o = mis + ex;
        // SIMD body vs scalar tail of the reference (sse3.h:494-540 vs 542-572)
        const bool body = ((i0 >> 3) < (count >> 3)) && (dataBytes - (pos + ex) >= 32u);
        uint32_t s[VPL];
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            uint32_t v = 0;
            if (k < valid) {
                const uint32_t len = ((keybits >> (2 * k)) & 3u) + 1u;
                const uint32_t w0 = stage32[o >> 2];
                const uint32_t w1 = stage32[(o >> 2) + 1];
                v = (uint32_t)(((uint64_t)w1 << 32 | w0) >> (8u * (o & 3u)));
                v &= 0xFFFFFFFFu >> (32u - 8u * len);
                o += len;
            }
            if (body) v &= 0xFFFFu;  // keep the low 16 bits (sse3.h:510-514)
            v = (v >> 1) ^ (0u - (v & 1u));
            acc += v;
            s[k] = acc;
        }
        const uint32_t ainc = wave_incl_scan_u32(acc);
        const uint32_t base = run + ainc - acc;
        run += wave_total_u32(ainc);
        if (valid == VPL && out_aligned) {
            u32x4 w;
            w.x = ((base + s[0]) & 0xFFFFu) | ((base + s[1]) << 16);
            w.y = ((base + s[2]) & 0xFFFFu) | ((base + s[3]) << 16);
            w.z = ((base + s[4]) & 0xFFFFu) | ((base + s[5]) << 16);
            w.w = ((base + s[6]) & 0xFFFFu) | ((base + s[7]) << 16);
            *(gst16*)(out + (size_t)i0 * 2) = w;
        } else {
#pragma unroll
            for (int k = 0; k < VPL; ++k)
                if (k < valid) *(gst2*)(out + (size_t)(i0 + k) * 2) = (uint16_t)(base + s[k]);
        }
        pos += tot;
    }
    pos_io = pos;
    run_io = run;
    return good;
}

// ---- the pipeline's pieces --------------------------------------------------------------------------------------------
// In that kernel a wave shares its SIMD with ONE other wave: every instruction it executes costs wall time (a lone wave
// issues a dependent instruction about every ten cycles), so the block pipeline is written for few instructions and for
// independent chains:
//   * a lane owns 64 CONSECUTIVE values of a block -- exactly the 16 control bytes it loads -- so a block needs ONE wave scan
//     for the data offsets (bytes per lane: 64 + the number of two-byte codes, four population counts) and ONE for the
//     delta chain, not one per 512 values;
//   * two samples per instruction (v_pk_* on 16-bit halves: the arithmetic of this path IS 16-bit, sse3.h:516-538);
//   * the bytes of two values are fetched with ONE unaligned ds_read_b32 and spread by ONE v_perm_b32 whose selector comes
//     from a four-entry table; where a pair starts follows from the control bits alone (a population count), so the 32
//     pairs of a lane are independent of each other up to the final carry chain;
//   * no branches inside a block.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((aligned(1))) const uint32_t lds32u;   // a dword at any LDS byte address: one ds_read_b32 on gfx950

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pk_add_hi_of(uint32_t a, uint32_t b)   // both halves of a += the high half of b
{
    const u16x2 bb = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + bb.yy));
}
__device__ __forceinline__ uint32_t pk_add_lo_of(uint32_t a, uint32_t b)   // both halves of a += the low half of b
{
    const u16x2 bb = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + bb.xx));
}
__device__ __forceinline__ uint32_t pk_unzigzag(uint32_t u)   // (u >> 1) ^ -(u & 1) on both halves
{
    const u16x2 v = __builtin_bit_cast(u16x2, u);
    const u16x2 one = { 1, 1 };
    return __builtin_bit_cast(uint32_t, (u16x2)((v >> one) ^ ((u16x2){ 0, 0 } - (v & one))));
}

// Selectors of v_perm_b32 that spread the bytes of two values -- b0 [b1] of the first, then those of the second -- to two
// 16-bit halves, by (first is two bytes, second is two bytes): the table the pipeline keeps in LDS, indexed by the three
// control bits code_a | 0 | code_b (codes are 0 or 1 on this path).
__device__ __forceinline__ uint32_t spread_selector(uint32_t t)
{
    return t == 0 ? 0x0c010c00u : (t == 1 ? 0x0c020100u : (t == 4 ? 0x02010c00u : 0x03020100u));
}

constexpr uint32_t SELTAB = DATABUF;          // byte offset of the selector table in the wave's LDS
constexpr uint32_t LDS_TOTAL = SELTAB + 32;
constexpr int LANE_VALUES = BLOCK / WAVE;     // 64 values = 16 control bytes = 32 dwords of samples per lane and block

// The 16 NW values of one lane: their control bits (k, NW words), the LDS byte address of the lane's first data byte (a0),
// the selector table (tab, LDS) -> d[8 NW]: the lane's samples two per dword as a prefix of their deltas (mod 2^16, without
// what the chain carries into the lane); the lane's delta total is the high half of the last dword.  nv (PARTIAL only): the
// lane's valid values, deltas behind them read as 0.
template <bool PARTIAL, int NW>
__device__ __forceinline__ void lane_deltas(const uint8_t* lds, const uint8_t* tab, const uint32_t (&k)[NW], uint32_t a0, uint32_t nv, uint32_t (&d)[8 * NW])
{
    // every LDS read of the lane first (their addresses follow from the control bits alone), then the arithmetic: one wait
    // for the lot instead of one per pair
    uint32_t x[8 * NW], sel[8 * NW];
    uint32_t wbase = a0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t kw = k[w];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // the pair starts 2p values and as many two-byte codes as stand below it into the word's bytes
            const uint32_t below = (uint32_t)((1ull << (4 * p)) - 1ull) & 0x55555555u;
            const uint32_t addr = p ? (uint32_t)__popc(kw & below) + (wbase + 2u * (uint32_t)p) : wbase;
            x[8 * w + p] = *(lds32u*)(lds + addr);
            const uint32_t ti = p ? (kw >> (4 * p - 2)) & 0x14u : (kw << 2) & 0x14u;   // 4 x (code_a | code_b << 2)
            sel[8 * w + p] = *reinterpret_cast<const uint32_t*>(tab + ti);
        }
        wbase += 16u + (uint32_t)__popc(kw & 0x55555555u);
    }
    __builtin_amdgcn_sched_barrier(0);   // (keep the reads in front of their uses)
#pragma unroll
    for (int i = 0; i < 8 * NW; ++i) {
        uint32_t v = pk_unzigzag(__builtin_amdgcn_perm(0u, x[i], sel[i]));
        if (PARTIAL) v = 2u * (uint32_t)i + 1u < nv ? v : (2u * (uint32_t)i < nv ? (v & 0xFFFFu) : 0u);
        d[i] = pk_add(v, v << 16);
    }
#pragma unroll
    for (int i = 1; i < 8 * NW; ++i) d[i] = pk_add_hi_of(d[i], d[i - 1]);
}

// all lanes of ONE wave (blockDim.x == 64): the svb stream in[0 .. in_size) -> out_size bytes of int16 samples at out.
// Returns what svb_decode_kernel<2, true, true> would leave in result[]: the byte count or an error code.
// lds: LDS_TOTAL bytes, 16-byte aligned, free for the duration of the call.
__device__ __forceinline__ uint32_t svb_decode_wave_i16zz(const uint8_t* in, uint32_t in_size, uint8_t* out, uint32_t out_size, uint8_t* lds, int lane)
{
    uint32_t res;
    if (svb_decode_check<2, true>(in_size, out_size, res)) return res;
    const uint32_t count = out_size / 2;
    const uint32_t keyLen = (count + 3u) >> 2;
    const uint8_t* data = in + keyLen;
    const uint32_t dataBytes = in_size - keyLen;
    uint8_t* databuf = lds;
    uint64_t pos = 0;
    uint32_t run = 0;
    uint32_t done = 0;  // values decoded so far
    const uint32_t nblocks = (count + BLOCK - 1) / BLOCK;   // the last one may be partial
    if ((((uintptr_t)out) & 15u) == 0) {
        // ---- the pipeline over blocks of 4096 values
        if (lane < 8) reinterpret_cast<uint32_t*>(lds + SELTAB)[lane] = spread_selector((uint32_t)lane);
        // the lane's 16 control bytes of block b, bytes behind the stream's control bytes zeroed (past the last block: zeros)
        auto load_keys = [&](uint32_t b) -> u32x4 {
            u32x4 v = { 0u, 0u, 0u, 0u };
            const uint64_t at = (uint64_t)b * 1024u + 16u * (uint32_t)lane;
            if (at + 16u <= keyLen) v = *(gld16*)(in + at);
            else if (at < keyLen) {
                const uint32_t n = keyLen - (uint32_t)at;
                uint32_t w[4] = { 0u, 0u, 0u, 0u };
                for (uint32_t i = 0; i < n; ++i) w[i >> 2] |= (uint32_t)((gld1*)in)[at + i] << (8u * (i & 3u));
                v = (u32x4){ w[0], w[1], w[2], w[3] };
            }
            return v;
        };
        // valid values of this lane in block b
        auto lane_valid = [&](uint32_t b) -> uint32_t {
            const uint64_t first = (uint64_t)b * BLOCK + (uint32_t)LANE_VALUES * (uint32_t)lane;
            return first >= count ? 0u : (count - (uint32_t)first < (uint32_t)LANE_VALUES ? count - (uint32_t)first : (uint32_t)LANE_VALUES);
        };
        // plan: the lane's data bytes -> its offset among the block's data bytes, the block's extent; wide: a code above 1
        uint32_t n_k[4] = { 0, 0, 0, 0 }, n_start = 0, n_extent = 0, n_mis = 0, n_nv = 0;
        uint32_t c_k[4], c_start = 0, c_mis = 0, c_nv = 0;
        bool n_wide = false;
        auto plan = [&](const u32x4& kv, uint32_t nv) {
            n_k[0] = kv.x;
            n_k[1] = kv.y;
            n_k[2] = kv.z;
            n_k[3] = kv.w;
            // (control bits behind the stream's last value are zero: vbz writers leave them so, and a foreign stream that
            // does not is a wide-code stream for this purpose only if the bits are odd ones -- checked with the rest)
            const uint32_t twos = (uint32_t)__popc(kv.x & 0x55555555u) + (uint32_t)__popc(kv.y & 0x55555555u) + (uint32_t)__popc(kv.z & 0x55555555u) +
                                  (uint32_t)__popc(kv.w & 0x55555555u);
            const uint32_t cnt = nv + twos;
            const uint32_t incl = wave_incl_scan_u32(cnt);
            n_start = incl - cnt;
            n_extent = wave_total_u32(incl);
            n_nv = nv;
            n_wide = __any(((kv.x | kv.y | kv.z | kv.w) & 0xAAAAAAAAu) != 0);
        };
        // the 16-byte pieces that cover the block's data bytes, nine per lane; a piece past the extent is fetched from the
        // last one's address instead (no branch; what it brings is never looked at)
        u32x4 dq[PIECES];
        auto fetch = [&](uint64_t at, uint32_t extent, uint32_t& mis_out) {
            const uint8_t* g0 = data + at;
            const uint32_t mis = (uint32_t)((uintptr_t)g0 & 15u);
            const uint8_t* ga = g0 - mis;
            const uint32_t lastc = (mis + (extent ? extent - 1u : 0u)) >> 4;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const uint32_t c = (uint32_t)q * WAVE + (uint32_t)lane;
                dq[q] = *(gld16*)(ga + 16ull * (c < lastc ? c : lastc));
            }
            mis_out = mis;
        };
        auto commit = [&]() {
#pragma unroll
            for (int q = 0; q < PIECES; ++q) *reinterpret_cast<u32x4*>(databuf + 16u * ((uint32_t)q * WAVE + (uint32_t)lane)) = dq[q];
        };
        // a block's partial lanes: control bits of values that do not exist must not count
        auto mask_keys = [&](u32x4 kv, uint32_t nv) -> u32x4 {
            if (nv >= (uint32_t)LANE_VALUES) return kv;
            uint32_t w[4] = { kv.x, kv.y, kv.z, kv.w };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t lo = 16u * (uint32_t)i;
                w[i] = nv <= lo ? 0u : (nv - lo >= 16u ? w[i] : (w[i] & ((1u << (2u * (nv - lo))) - 1u)));
            }
            return (u32x4){ w[0], w[1], w[2], w[3] };
        };
        // prologue: block 0 planned, fetched and committed; the control bytes of block 1 in flight
        u32x4 kq = load_keys(0);
        plan(mask_keys(kq, lane_valid(0)), lane_valid(0));
        kq = load_keys(1);
        bool go = !n_wide;
        bool overrun = false;
        if (go && n_extent > dataBytes) {
            go = false;
            overrun = true;
        }
        if (go) {
            fetch(0, n_extent, n_mis);
            wave_lds_sync();
            commit();
        }
        for (uint32_t b = 0; go; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c_k[i] = n_k[i];
            c_start = n_start;
            c_mis = n_mis;
            c_nv = n_nv;
            const uint64_t pos_next = pos + n_extent;
            const bool last_block = b + 1 == nblocks;
            bool have_next = !last_block;
            // plan(b + 1) from the control bytes loaded a block ago, those of block b + 2 requested; fetch(b + 1) -- in flight
            // while block b is decoded
            if (have_next) {
                const uint32_t nv1 = lane_valid(b + 1);
                plan(b + 2 == nblocks ? mask_keys(kq, nv1) : kq, nv1);
                kq = load_keys(b + 2);
                if (n_wide) have_next = false;
                else if (pos_next + n_extent > dataBytes) {
                    have_next = false;
                    overrun = true;
                }
            }
            if (have_next) fetch(pos_next, n_extent, n_mis);
            wave_lds_sync();   // block b is in LDS
            // decode(b)
            uint32_t d[32];
            const uint32_t a0 = c_mis + c_start;
            if (last_block) lane_deltas<true, 4>(lds, lds + SELTAB, c_k, a0, c_nv, d);
            else lane_deltas<false, 4>(lds, lds + SELTAB, c_k, a0, c_nv, d);
            const uint32_t tl = d[31] >> 16;
            const uint32_t il = wave_incl_scan_u32(tl);
            const uint32_t base = run + il - tl;
            run += wave_total_u32(il);
#pragma unroll
            for (int i = 0; i < 32; ++i) d[i] = pk_add_lo_of(d[i], base);
            uint8_t* o = out + ((size_t)done + (size_t)LANE_VALUES * (uint32_t)lane) * 2;
            if (!last_block || c_nv == (uint32_t)LANE_VALUES) {
#pragma unroll
                for (int q = 0; q < 8; ++q) *(gst16*)(o + 16 * q) = (u32x4){ d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3] };
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (8u * (uint32_t)q + 8u <= c_nv) *(gst16*)(o + 16 * q) = (u32x4){ d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3] };
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (8u * (uint32_t)q + (uint32_t)e < c_nv) *(gst2*)(o + 16 * q + 2 * e) = (uint16_t)(d[4 * q + (e >> 1)] >> (16 * (e & 1)));
                    }
                }
            }
            done = last_block ? count : done + BLOCK;
            pos = pos_next;
            if (!have_next) break;
            wave_lds_sync();
            commit();   // block b + 1 to LDS
        }
        if (overrun) return E_STREAM;   // the stream is shorter than its control bytes claim
    }
    // ---- whatever is left: a stream with wider codes from the block that has them on, unaligned destinations.
    // (copies: variables whose address goes to a called function live in scratch memory, and the pipeline's loop would
    // read them back from there behind its fetches -- a wait for all of them)
    uint64_t tail_pos = pos;
    uint32_t tail_run = run;
    if (done < count) {
        wave_lds_sync();
        if (!svb_decode_wave_tiles(in, data, dataBytes, count, done, count, tail_pos, tail_run, out, databuf, lane)) return E_STREAM;
    }
    return tail_pos != dataBytes ? E_STREAM : count * 2;
}

}  // namespace svbwave

}  // namespace vbzhip
