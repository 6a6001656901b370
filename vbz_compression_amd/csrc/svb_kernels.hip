// svb_kernels.hip -- delta + zig-zag + streamvbyte stage of the VBZ path as HIP kernels for gfx950.
//
// Replaces (reference file:line):
//   StreamVByteWorkerV0<int16_t,true>::compress / ::decompress   vbz/v0/vbz_streamvbyte_impl_sse3.h:406-466, 468-580
//   StreamVByteWorkerV0<T,ZigZag>::compress / ::decompress       vbz/v0/vbz_streamvbyte_impl.h:18-73
//   (dispatch: vbz/v0/vbz_streamvbyte.cpp:20-108, vbz/v1/vbz_streamvbyte.cpp:22-113)
//
// Design (MI355X-first, not a translation of the SSSE3 loop):
//   * one 256-thread workgroup per read; the read is walked in tiles of 256 lanes x VPL values, every
//     lane loading 16 contiguous bytes (global_load_dwordx4, 1 KiB per wave instruction);
//   * the variable-length byte scatter/gather is turned into a prefix sum: per-lane byte counts ->
//     wave scan (cross-lane shuffles) -> workgroup scan through 4 LDS words -> running carry per read;
//   * data bytes are staged through LDS so that HBM only ever sees 16-byte aligned, coalesced
//     dwordx4 stores/loads, whatever the byte alignment of the data section;
//   * the int16 wrap-around delta chain and its inverse (inclusive prefix sum mod 2^16) use the same
//     scan primitives; the previous sample comes from the neighbouring lane by a shuffle.
// Algorithmic HBM bytes per int16 sample: encode 2 read + ~1.26 written; decode the reverse.
#include "vbz_kernels.h"
#include "svb_wave.h"

namespace vbzhip {

namespace {

constexpr int WG = 256;

template <int ELEM>
struct Vpl
{
    static constexpr int value = (ELEM == 4) ? 4 : 8;  // values per lane per tile
};

__device__ __forceinline__ int32_t load_elem(const uint8_t* p, int elem)
{
    if (elem == 1) return (int8_t)p[0];
    if (elem == 2) {
        uint16_t v;
        __builtin_memcpy(&v, p, 2);
        return (int16_t)v;
    }
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return (int32_t)v;
}

__device__ __forceinline__ void store_elem(uint8_t* p, int elem, uint32_t v)
{
    if (elem == 1) p[0] = (uint8_t)v;
    else if (elem == 2) {
        uint16_t t = (uint16_t)v;
        __builtin_memcpy(p, &t, 2);
    } else __builtin_memcpy(p, &v, 4);
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
// Per-read validation shared by the one-workgroup-per-read kernel and the segmented kernels: 0 = go on, or the error.
template <int ELEM, bool I16ZZ>
__device__ __forceinline__ uint32_t svb_encode_check(uint32_t size, uint32_t cap, uint32_t hdr, uint32_t strict_cap)
{
    if (size % ELEM != 0) return E_INPUT_SIZE;  // vbz/v0/vbz_streamvbyte.cpp:28-31
    const uint32_t n = size / ELEM;
    const uint32_t keyLen = (n + 3u) >> 2;
    // the reference requires the destination to hold the worst case (vbz/vbz.cpp:171-174); the
    // library's own scratch slots (strict_cap == 0) are sized for what this kernel can really emit
    const uint64_t worst = (uint64_t)keyLen + ((I16ZZ && !strict_cap) ? 2ull : 4ull) * n + hdr;
    if (worst > cap) return worst > 0xFFFFFFF0ull ? E_INPUT_SIZE : E_DESTINATION_SIZE;
    return 0;
}

// ---- probe for a long repeat distance in the data bytes (consumed by zstd_encode.hip's long-repeat coder) -------------
// libzstd's match finder (on at every level of the reference: vbz/vbz.cpp:194-207) turns signal that cycles a template --
// the reference's own perf generator does, vbz/perf/test_data_generator.h:61-67 -- into a few long matches: the data bytes
// of such a read repeat at ONE distance D.  Finding D must cost a read without one next to nothing, and this kernel has
// every data byte in LDS once anyway.  The sixteen dwords that start at data bytes p0 .. p0 + 15 are the probes; whatever D
// is, exactly one of them comes to lie, D bytes further down, on the first dword of a 16-byte chunk of the flush loop.  So
// the flush looks up the first dword of every chunk it writes in a small hash table (value -> probe number; an empty cell
// holds a value that does not hash to it, so nothing matches it): a multiply, a shift, one LDS read and a compare per 16
// data bytes.  A hit whose 16 bytes equal the 16 bytes behind the probe proposes D = position - (p0 + probe number); the
// smallest proposal is left in hint[r] (0: none), and the entropy stage checks it before it relies on it.
constexpr uint32_t PROBE_TABLE = 512, PROBE_P0 = 256, PROBE_MIN_D = 64, PROBE_TRIES = 4, PROBE_SHIFT = 80;
struct PeriodProbe
{
    uint32_t val[PROBE_TABLE];
    uint8_t idx[PROBE_TABLE];
    uint32_t ctx[12];   // the 32 + 16 data bytes from p0 on (dword k = bytes p0 + 4k ...)
    uint32_t best;      // smallest distance proposed so far
    uint32_t lost;      // table construction: a probe collided
};
__device__ __forceinline__ uint32_t probe_hash(uint32_t w) { return (w * 0x9E3779B1u) >> 23; }

// ---- the data bytes' histogram on the way (CNT; int16 zig-zag reads into library scratch) -------------------------------------------
// The entropy stage builds the data-byte region's Huffman table from a histogram of one kilobyte in four (region_histogram of
// zstd_encode.hip) -- a pass of its own over bytes this kernel has just had in LDS.  Here exactly that sample (the unaligned ends and
// every fourth stripe of 64 aligned 16-byte chunks; everything, in two parts, for reads so short that the region may have to be counted
// exactly) is counted by LDS atomics while the bytes are flushed, and left in the read's plan (EncPlan::hist_mode): the same counts, so
// the frames are byte for byte what they were, and the entropy stage's planning never reads the data bytes.
// Round 5 also moved the control bytes' TOKENISER in here, twice -- every wavefront deciding its own 128 control bytes per tile with
// wave-uniform scalar arithmetic (a CU has ONE scalar unit: 2.3 x the kernel's time), then tokenise_runs' own arithmetic as "trips" of
// one wavefront over an LDS ring of control bytes (bit-exact, and + 45 % on this kernel for - 11 % on the planning launch: a trip's
// latency, ~5 000 cycles, sits on the workgroup's barrier-synchronised critical path) -- and took it out again: profiles/r05_experiments.md.
constexpr uint32_t CNT_MIN_VALUES = 1640;            // n + ceil(n / 4) >= 2048: the entropy stage then cuts the stream into its two regions
constexpr uint32_t CNT_HIST_SAMPLE_FROM = 32u << 10; // region_histogram samples regions of this size and more (HIST_SAMPLE_FROM)

struct CntLds
{
    uint32_t hA[4][256];      // data bytes of the sample, one copy per wavefront (atomics on a shared bin serialise)
    uint32_t hB[256];         // the other data bytes (hist_mode 2)
};

__device__ __forceinline__ void cnt_count16(uint32_t* h, const uint4& v)
{
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&h[(w[k] >> (8 * j)) & 0xFFu], 1u);
    }
}

// ---- pairs of whole int16 zig-zag tiles (round 6) --------------------------------------------------------------------------------------
// The tile loop of svb_encode_range issues ~185 wave-level VALU instructions per 512 samples of a wavefront, most of them the 4-cycle
// kind (tools/valu_rate.hip: compares, selects, v_bfe, DPP), and four workgroup barriers per tile: the kernel sat on instruction issue
// at 4.6 ms per 65 536 reads where its 21.4 GB would take 3.5 at a copy's rate.  Whole tiles of int16 samples -- all but the ends of a
// read -- now go through this loop, two tiles (4096 samples) per trip:
//   * the samples stay PACKED (two per register) from the load to the byte stores: delta and zig-zag as before (v_pk_*), then the eight
//     high bytes of a lane's values are gathered into two registers (v_perm_b32) and "is this byte non-zero" is byte-parallel
//     arithmetic on four values at once -- (((h & 0x7f7f7f7f) + 0x7f7f7f7f) | h) & 0x80808080 --; one multiplication turns the four
//     flags into the key byte (bits 0, 2, 4, 6), another into their prefix sums (a byte each: where each value's bytes go, and the
//     lane's byte count), instead of eight compares and a dozen selects;
//   * ONE workgroup scan for both tiles: their byte counts ride in the two halves of one register (a tile has at most 4096 bytes);
//   * byte emission as before (two ds_write_b8 per value, a one-byte value's zero high byte overwritten by the next value's low byte),
//     but straight from the packed registers (ds_write_b8_d16_hi for the odd values) at offsets cut out of the prefix-sum register;
//   * two stage buffers used in turn, so that the bytes that do not fill a 16-byte chunk move to the OTHER buffer while this one is
//     being flushed: two barriers per trip (eight before, for two tiles), and every thread of the workgroup has a chunk to flush.
// Same bytes, same probe and histogram hand-over (the flush is the old one); tests/test_gpu_parity.py holds both to the oracle.
template <bool PROBE, bool CNT>
struct I16Pairs
{
    // one stage buffer: the kept bytes + two tiles at 3/2 bytes per value (signal: 1.01; all of 2 would be 8 KB, and a workgroup's LDS
    // decides how many of them a CU holds).  A pair that does not fit ends the loop: the tile loop of svb_encode_range codes the rest.
#ifndef VBZ_ENC_BUF_BYTES
#define VBZ_ENC_BUF_BYTES (2 * WG * 8 * 3 / 2 + 32)
#endif
    static constexpr uint32_t TILE = WG * 8, BUF = VBZ_ENC_BUF_BYTES;
    const uint8_t* in;
    uint8_t* keys;
    uint8_t* gal;
    uint8_t* stage;      // two buffers of BUF bytes; [0] is the one the caller's own loop uses
    uint32_t* wsum;      // 8 words
    uint16_t* kb16;      // 2 x 512 halfwords: the control bytes of a pair on their way out (two sets used in turn)
    PeriodProbe* pp;
    CntLds* CL;
    uint32_t A, hmode;
    bool cnt_on;
    uint32_t probe_p0;

    // zig-zag of the wrap-around deltas of a lane's eight samples, packed (sse3.h:432-440); pw's top half = the sample in front
    static __device__ __forceinline__ void zz4(const uint4& q, uint32_t pw, uint32_t Z[4])
    {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t prevw = __builtin_amdgcn_alignbit(w[k], k ? w[k - 1] : pw, 16);
            const s16x2 d = __builtin_bit_cast(s16x2, w[k]) - __builtin_bit_cast(s16x2, prevw);
            Z[k] = __builtin_bit_cast(uint32_t, (s16x2)((d << (s16x2)1) ^ (d >> (s16x2)15)));
        }
    }
    // flags (bit 0 of every byte) "value j of four needs two bytes", from the packed values Za (values 0, 1) and Zb (values 2, 3)
    static __device__ __forceinline__ uint32_t flags4(uint32_t Za, uint32_t Zb)
    {
        const uint32_t h = __builtin_amdgcn_perm(Zb, Za, 0x07050301u);   // the four high bytes
        return ((((h & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | h) & 0x80808080u) >> 7;
    }
    // a lane's 8 .. 16 data bytes at s (LDS): both bytes of every value are written, a one-byte value's zero high byte is overwritten by
    // the next value's low byte (a later instruction of this lane); only the lane's LAST value must not touch the byte behind it
    static __device__ __forceinline__ void emit8(uint8_t* s, const uint32_t Z[4], uint32_t ps0, uint32_t ps1)
    {
        const uint32_t Y0 = Z[0] >> 8, Y1 = Z[1] >> 8, Y2 = Z[2] >> 8, Y3 = Z[3] >> 8;
        s[1] = (uint8_t)Y0;
        uint8_t* t = s + (ps0 & 0xFFu);
        t[1] = (uint8_t)(Z[0] >> 16);
        s[0] = (uint8_t)Z[0];   // (behind a store that may alias s[1]: the two are not to be fused into one unaligned 16-bit store, which is slow)
        t[2] = (uint8_t)(Y0 >> 16);
        t = s + ((ps0 >> 8) & 0xFFu);
        t[2] = (uint8_t)Z[1];
        t[3] = (uint8_t)Y1;
        t = s + ((ps0 >> 16) & 0xFFu);
        t[3] = (uint8_t)(Z[1] >> 16);
        t[4] = (uint8_t)(Y1 >> 16);
        uint8_t* const s4 = s + (ps0 >> 24);
        s4[4] = (uint8_t)Z[2];
        s4[5] = (uint8_t)Y2;
        t = s4 + (ps1 & 0xFFu);
        t[5] = (uint8_t)(Z[2] >> 16);
        t[6] = (uint8_t)(Y2 >> 16);
        t = s4 + ((ps1 >> 8) & 0xFFu);
        t[6] = (uint8_t)Z[3];
        t[7] = (uint8_t)Y3;
        t = s4 + ((ps1 >> 16) & 0xFFu);
        t[7] = (uint8_t)(Z[3] >> 16);
        if (Z[3] >> 24) t[8] = (uint8_t)(Y3 >> 16);
    }

    // one 16-byte chunk of the stage leaves for gal + F + 16 c (the flush of svb_encode_range, chunk by chunk)
    __device__ __forceinline__ void flush_chunk(const uint8_t* buf, uint64_t F, uint32_t c, int wv)
    {
        uint8_t* g = gal + F + 16ull * c;
        if (F == 0 && c == 0 && A != 0) {
            for (uint32_t j = A; j < 16; ++j) g[j] = buf[j];
            if (CNT && cnt_on)
                for (uint32_t j = A; j < 16; ++j) atomicAdd(&CL->hA[wv][buf[j]], 1u);
            return;
        }
        const uint4 v = *reinterpret_cast<const uint4*>(buf + 16u * c);
        *reinterpret_cast<uint4*>(g) = v;
        if (CNT && cnt_on) {
            const uint32_t cq = (uint32_t)(F >> 4) + c - (A ? 1u : 0u);
            if (((cq >> 6) & 3u) == 0u) cnt_count16(CL->hA[wv], v);
            else if (hmode == 2u) cnt_count16(CL->hB, v);
        }
        if (PROBE && probe_p0) {
            const uint32_t h = probe_hash(v.x);
            if (pp->val[h] == v.x) {
                const uint64_t rel = F + 16ull * c - A;
                const uint32_t j = pp->idx[h], k = j >> 2, sh = j & 3u;
                const uint32_t c0 = __builtin_amdgcn_alignbyte(pp->ctx[k + 1], pp->ctx[k], sh), c1 = __builtin_amdgcn_alignbyte(pp->ctx[k + 2], pp->ctx[k + 1], sh);
                const uint32_t c2 = __builtin_amdgcn_alignbyte(pp->ctx[k + 3], pp->ctx[k + 2], sh), c3 = __builtin_amdgcn_alignbyte(pp->ctx[k + 4], pp->ctx[k + 3], sh);
                if (rel >= (uint64_t)probe_p0 + j + PROBE_MIN_D && rel < 0xFFFFFFFFull && c0 == v.x && c1 == v.y && c2 == v.z && c3 == v.w)
                    atomicMin(&pp->best, (uint32_t)rel - (probe_p0 + j));
            }
        }
    }

    // tiles t0, t0 + TILE, ... while two whole ones are left; the bytes [F, P) the caller holds at the head of stage buffer 0 are
    // taken over and handed back the same way.  Returns the first value not coded.  All 256 threads; the caller's next barrier
    // orders the last LDS writes.
    static __device__ __forceinline__ uint64_t uniform64(uint64_t v)
    {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
    }
    __device__ __forceinline__ uint32_t run(uint32_t t0_in, uint32_t end_in, uint64_t& F_io, uint64_t& P_io)
    {
        const int tid = threadIdx.x, lane = tid & 63;
        // workgroup-uniform state, held in scalar registers
        uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0_in);
        const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)end_in);
        uint64_t F = uniform64(F_io), P = uniform64(P_io);
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t m0 = wv > 0 ? 0xFFFFFFFFu : 0u, m1 = wv > 1 ? 0xFFFFFFFFu : 0u, m2 = wv > 2 ? 0xFFFFFFFFu : 0u;
        uint32_t cur = 0;   // the stage buffer this trip writes
        const uint8_t* ip = in + ((size_t)t0 + (size_t)tid * 8) * 2;
        uint8_t* kp0 = keys + (t0 >> 2);   // the pair's control bytes: 1 KB from here
        // lane 0 of a wavefront: the aligned dword in front of its samples -- its top half is the sample in front (samples start 16-byte
        // aligned here, and the read's very first value has nothing in front of it)
        auto prev_of = [&](const uint8_t* p, bool first_value) -> uint32_t {
            return (lane == 0 && !first_value) ? *reinterpret_cast<const uint32_t*>(p - 4) : 0u;
        };
        uint4 na = *reinterpret_cast<const uint4*>(ip), nb = *reinterpret_cast<const uint4*>(ip + TILE * 2);
        uint32_t npa = prev_of(ip, t0 == 0 && tid == 0), npb = prev_of(ip + TILE * 2, false);
        uint32_t par = 0;
        while (end - t0 >= 2u * TILE) {
            const uint4 qa = na, qb = nb;
            const uint32_t pa = npa, pb = npb;
            if (end - t0 >= 4u * TILE) {   // the next pair's samples are requested a trip ahead
                na = *reinterpret_cast<const uint4*>(ip + TILE * 4);
                nb = *reinterpret_cast<const uint4*>(ip + TILE * 6);
                npa = prev_of(ip + TILE * 4, false);
                npb = prev_of(ip + TILE * 6, false);
            }
            uint32_t ZA[4], ZB[4];
            {
                uint32_t pw = wave_prev_lane_u32(qa.w);
                if (lane == 0) pw = pa;
                zz4(qa, pw, ZA);
                pw = wave_prev_lane_u32(qb.w);
                if (lane == 0) pw = pb;
                zz4(qb, pw, ZB);
            }
            const uint32_t fa0 = flags4(ZA[0], ZA[1]), fa1 = flags4(ZA[2], ZA[3]), fb0 = flags4(ZB[0], ZB[1]), fb1 = flags4(ZB[2], ZB[3]);
            // key bytes: flag j of four to bit 2 j (codes 0 / 1 only); prefix sums of the flags, a byte each
            const uint32_t ka = ((fa0 * 0x01041040u) >> 24) | (((fa1 * 0x01041040u) >> 24) << 8);
            const uint32_t kb = ((fb0 * 0x01041040u) >> 24) | (((fb1 * 0x01041040u) >> 24) << 8);
            const uint32_t pa0 = fa0 * 0x01010101u, pa1 = fa1 * 0x01010101u, pb0 = fb0 * 0x01010101u, pb1 = fb1 * 0x01010101u;
            // control bytes: a lane's two bytes per tile go to LDS, and behind the barrier ONE wavefront stores the pair's kilobyte of them
            // as 16 bytes per lane (two 2-byte stores per lane of every wavefront measured 0.36 ms of the kernel's 4.2: 1.6 GB in 128-byte
            // requests)
            kb16[par * 128 + (uint32_t)tid] = (uint16_t)ka;
            kb16[par * 128 + 256u + (uint32_t)tid] = (uint16_t)kb;
            // one scan for both tiles: tile A's byte count in the low half, tile B's in the high half
            const uint32_t L = (8u + (pa0 >> 24) + (pa1 >> 24)) | ((8u + (pb0 >> 24) + (pb1 >> 24)) << 16);
            const uint32_t inc = wave_incl_scan_u32(L);
            uint32_t* ws = wsum + par;
            if (lane == 63) ws[wv] = inc;
            wg_lds_barrier();   // (also: the previous trip's kept bytes stand in this trip's buffer, its flush has read the other one)
            if (tid < 64) {
                const uint4 kv = *reinterpret_cast<const uint4*>(kb16 + par * 128 + 8u * (uint32_t)tid);
                __builtin_memcpy(kp0 + 16u * (uint32_t)tid, &kv, 16);
            }
            const uint4 s4 = *reinterpret_cast<const uint4*>(ws);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s4.x + s4.y + s4.z + s4.w));
            const uint32_t ex = (s4.x & m0) + (s4.y & m1) + (s4.z & m2) + inc - L;
            const uint32_t R = (uint32_t)(P - F), totA = tot & 0xFFFFu, totB = tot >> 16;
            if (R + totA + totB + 1u > BUF) break;   // (workgroup-uniform; nothing of this pair is in the stage -- its control bytes are written again)
            uint8_t* const buf = stage + cur * BUF;
            emit8(buf + R + (ex & 0xFFFFu), ZA, pa0, pa1);
            emit8(buf + R + totA + (ex >> 16), ZB, pb0, pb1);
            wg_lds_barrier();
            const uint32_t endidx = R + totA + totB, nch = endidx >> 4, rem = endidx & 15u;
            if ((uint32_t)tid < nch) flush_chunk(buf, F, (uint32_t)tid, wv);
            if ((uint32_t)tid + WG < nch) flush_chunk(buf, F, (uint32_t)tid + WG, wv);
            if ((uint32_t)tid + 2 * WG < nch) flush_chunk(buf, F, (uint32_t)tid + 2 * WG, wv);   // (only when most values take two bytes)
            if ((uint32_t)tid < rem) stage[(cur ^ 1u) * BUF + tid] = buf[16u * nch + tid];   // the bytes short of a chunk: to the other buffer
            F += 16ull * nch;
            P += totA + totB;
            cur ^= 1u;
            par ^= 4u;
            t0 += 2u * TILE;
            ip += TILE * 4;
            kp0 += TILE / 2;
        }
        wg_lds_barrier();   // the last flush has read its buffer, the kept bytes stand in stage[cur]
        if (cur != 0) {
            if ((uint32_t)tid < (uint32_t)(P - F)) stage[tid] = stage[BUF + tid];
        }
        F_io = F;
        P_io = P;
        return t0;
    }
};

// Values [first, end) of a read of n values (first a multiple of the tile size): control bytes to keys[first/4 ...),
// data bytes to data[0 ...) -- `data` is where this range's data bytes start, any alignment; only bytes of the range
// are touched.  COUNT_ONLY: nothing is written, the data byte count is all that is wanted.  Returns the data bytes
// (workgroup-uniform).  All 256 threads.
// PROBE (one-workgroup-per-read kernel only, data starts at first == 0): see PeriodProbe; *hint_out gets the distance.
// CNT (one-workgroup-per-read kernel, first == 0): hmode != 0 = the data bytes' sample is counted on the way into *CL (see CntLds;
// hmode: EncPlan::hist_mode).
template <int ELEM, bool ZZ, bool I16ZZ, bool COUNT_ONLY, bool PROBE = false, bool CNT = false>
__device__ __forceinline__ uint64_t svb_encode_range(const uint8_t* in, uint32_t first, uint32_t end, uint8_t* keys, uint8_t* data,
                                                     uint8_t* stage, uint32_t* wsum, PeriodProbe* pp = nullptr, uint32_t* hint_out = nullptr,
                                                     CntLds* CL = nullptr, uint32_t hmode = 0)
{
    constexpr int VPL = Vpl<ELEM>::value;
    constexpr int TILE = WG * VPL;
    static_assert(!CNT || !COUNT_ONLY, "counting rides on the flush");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const bool cnt_on = CNT && hmode != 0;      // the data bytes' sample is counted on the way
    (void)wv;
    const uint32_t A = (uint32_t)((uintptr_t)data & 15u);
    uint8_t* gal = data - A;  // 16-byte aligned address space of the data section
    const bool in_aligned = (((uintptr_t)in) & 15u) == 0;

    uint64_t F = 0;  // bytes of the aligned space already flushed (multiple of 16)
    uint64_t P = A;  // next byte position in the aligned space; stage[] holds [F, P)
    uint32_t probe_p0 = 0;   // != 0: the probe table stands, with its probes at data bytes probe_p0 ...
    bool probe_runs = false; // the probes could not be placed: the distance 1 ("a run") is proposed (zstd_encode.hip: RUN_D)
    if (PROBE) {
        for (uint32_t i = tid; i < PROBE_TABLE; i += WG) pp->val[i] = i == 0 ? 1u : 0u;   // hash(0) = 0, hash(1) != 0
        if (tid == 0) {
            pp->best = 0xFFFFFFFFu;
            pp->lost = 0;
        }
        // (the scan of the first tile holds the barrier that orders these writes)
    }

    // the next tile's 16 bytes per lane are requested before this tile is processed (one load always in flight)
    uint4 qnext = make_uint4(0u, 0u, 0u, 0u);
    {
        const uint32_t i0 = first + (uint32_t)tid * VPL;
        if (in_aligned && i0 < end && end - i0 >= (uint32_t)VPL) qnext = *reinterpret_cast<const uint4*>(in + (size_t)i0 * ELEM);
    }
    constexpr bool FAST = I16ZZ && !COUNT_ONLY;   // pairs of whole int16 tiles take svb_i16zz_pairs (below)
    bool fast_done = !FAST || !in_aligned;
    for (uint32_t t0 = first; t0 < end; t0 += TILE) {
        if (FAST && !fast_done && (!PROBE || t0 != first)) {
            // (with PROBE the read's first tile goes through the loop below, which builds the probe table from it)
            fast_done = true;
            if (end - t0 >= 2u * (uint32_t)TILE) {
                I16Pairs<PROBE, CNT> fp = { in, keys, gal, stage, wsum, reinterpret_cast<uint16_t*>(stage + 2 * I16Pairs<PROBE, CNT>::BUF), pp, CL, A, hmode, cnt_on, probe_p0 };
                t0 = fp.run(t0, end, F, P);
                if (t0 >= end) break;
                const uint32_t i0 = t0 + (uint32_t)tid * VPL;
                qnext = make_uint4(0u, 0u, 0u, 0u);
                if (i0 < end && end - i0 >= (uint32_t)VPL) qnext = *reinterpret_cast<const uint4*>(in + (size_t)i0 * ELEM);
            }
        }
        const uint32_t i0 = t0 + (uint32_t)tid * VPL;
        const int valid = i0 >= end ? 0 : (end - i0 >= (uint32_t)VPL ? VPL : (int)(end - i0));
        const bool full = in_aligned && end - t0 >= (uint32_t)TILE;
        const uint4 q = qnext;
        {
            const uint32_t i1 = i0 + TILE;
            if (in_aligned && i1 < end && end - i1 >= (uint32_t)VPL) qnext = *reinterpret_cast<const uint4*>(in + (size_t)i1 * ELEM);
        }
        uint32_t u[VPL];
        if (I16ZZ && full) {
            // full int16 tile: wrap-around delta and 16-bit zig-zag (sse3.h:432-440) on two samples per instruction
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const uint32_t w[4] = { q.x, q.y, q.z, q.w };
            uint32_t pw = wave_prev_lane_u32(w[3]);  // its top half: the sample in front of this lane's first
            if (lane == 0) pw = i0 == 0 ? 0u : ((uint32_t)(uint16_t)load_elem(in + (size_t)(i0 - 1) * ELEM, ELEM) << 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t prevw = __builtin_amdgcn_alignbit(w[k], k ? w[k - 1] : pw, 16);
                const s16x2 d = __builtin_bit_cast(s16x2, w[k]) - __builtin_bit_cast(s16x2, prevw);
                const uint32_t zz = __builtin_bit_cast(uint32_t, (s16x2)((d << (s16x2)1) ^ (d >> (s16x2)15)));
                u[(2 * k) % VPL] = zz & 0xFFFFu;
                u[(2 * k + 1) % VPL] = zz >> 16;
            }
        } else {
        int32_t x[VPL];
        if (valid == VPL && in_aligned) {
            const uint32_t w[4] = { q.x, q.y, q.z, q.w };
            if (ELEM == 4) {
#pragma unroll
                for (int k = 0; k < VPL; ++k) x[k] = (int32_t)w[k & 3];
            } else if (ELEM == 2) {
#pragma unroll
                for (int k = 0; k < VPL; ++k) x[k] = (int16_t)(w[(k >> 1) & 3] >> (16 * (k & 1)));
            } else {
#pragma unroll
                for (int k = 0; k < VPL; ++k) x[k] = (int8_t)(w[(k >> 2) & 3] >> (8 * (k & 3)));
            }
        } else {
#pragma unroll
            for (int k = 0; k < VPL; ++k) x[k] = k < valid ? load_elem(in + (size_t)(i0 + k) * ELEM, ELEM) : 0;
        }
        if (ZZ) {
            // previous sample: neighbouring lane's last value; wave lane 0 re-reads it from memory
            int32_t prev = (int32_t)wave_prev_lane_u32((uint32_t)x[VPL - 1]);
            if (lane == 0) prev = (i0 == 0 || valid == 0) ? 0 : load_elem(in + (size_t)(i0 - 1) * ELEM, ELEM);
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                if (I16ZZ) {
                    // wrap-around int16 delta, 16-bit zig-zag (sse3.h:432-440)
                    uint32_t d = ((uint32_t)x[k] - (uint32_t)prev) & 0xFFFFu;
                    u[k] = ((d << 1) ^ (0u - (d >> 15))) & 0xFFFFu;
                } else {
                    uint32_t d = (uint32_t)x[k] - (uint32_t)prev;  // streamvbyte zigzag_delta_encode
                    u[k] = (d << 1) ^ (uint32_t)((int32_t)d >> 31);
                }
                prev = x[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < VPL; ++k) u[k] = (uint32_t)x[k];
        }
        }
        uint32_t keybits = 0, L = 0;
        if (I16ZZ && full) {  // full tile of one- or two-byte values
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const uint32_t code = u[k] >> 8 ? 1u : 0u;
                keybits |= code << (2 * k);
                L += code;
            }
            L += VPL;
            if (!COUNT_ONLY) {
                const uint16_t kk = (uint16_t)keybits;
                __builtin_memcpy(keys + (i0 >> 2), &kk, 2);
            }
        } else {
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            uint32_t code = (u[k] > 0xFFu) + (u[k] > 0xFFFFu) + (u[k] > 0xFFFFFFu);
            if (k < valid) {
                keybits |= code << (2 * k);
                L += code + 1;
            }
        }
        }
        // control bytes: VPL/4 per lane, contiguous across the wave
        if (!COUNT_ONLY && !(I16ZZ && full) && valid > 0) {
            uint8_t* kp = keys + (i0 >> 2);
            if (VPL == 8 && valid > 4) {
                uint16_t kk = (uint16_t)keybits;
                __builtin_memcpy(kp, &kk, 2);
            } else {
                kp[0] = (uint8_t)keybits;
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(L, wsum, tot);
        if (COUNT_ONLY) {
            P += tot;
            continue;
        }
        uint32_t o = (uint32_t)(P - F) + ex;
        if (I16ZZ && full) {
            // full tile, one or two bytes per value: both bytes are always written and a one-byte value's second
            // byte is overwritten by the lane's next value (a later instruction) -- no branches; only the lane's
            // last value must not touch the next lane's first byte
#pragma unroll
            for (int k = 0; k < VPL - 1; ++k) {
                stage[o] = (uint8_t)u[k];
                stage[o + 1] = (uint8_t)(u[k] >> 8);
                o += 1u + (u[k] > 0xFFu ? 1u : 0u);
            }
            stage[o] = (uint8_t)u[VPL - 1];
            if (u[VPL - 1] > 0xFFu) stage[o + 1] = (uint8_t)(u[VPL - 1] >> 8);
        } else {
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            if (k < valid) {
                uint32_t v = u[k];
                stage[o++] = (uint8_t)v;
                if (v > 0xFFu) stage[o++] = (uint8_t)(v >> 8);
                if (!I16ZZ) {
                    if (v > 0xFFFFu) stage[o++] = (uint8_t)(v >> 16);
                    if (v > 0xFFFFFFu) stage[o++] = (uint8_t)(v >> 24);
                }
            }
        }
        }
        wg_lds_barrier();
        const uint32_t endidx = (uint32_t)(P - F) + tot;
        const uint32_t nch = endidx >> 4;
        if (PROBE && t0 == 0 && endidx >= A + PROBE_P0 + PROBE_TRIES * PROBE_SHIFT + 48u) {
            // the first tile holds the probes: sixteen distinct dwords in sixteen different cells, else another place
            uint32_t p0 = PROBE_P0;
            for (uint32_t attempt = 0; attempt < PROBE_TRIES; ++attempt, p0 += PROBE_SHIFT) {
                uint32_t pw = 0, h = 0;
                if (tid < 16) {
                    const uint8_t* q = stage + A + p0 + tid;
                    pw = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                    h = probe_hash(pw);
                    pp->val[h] = pw;
                    pp->idx[h] = (uint8_t)tid;
                }
                wg_lds_barrier();
                if (tid < 16 && (pp->val[h] != pw || pp->idx[h] != (uint8_t)tid)) pp->lost = 1;   // two probes in one cell (or equal)
                wg_lds_barrier();
                const bool lost = pp->lost != 0;
                wg_lds_barrier();
                if (!lost) {
                    probe_p0 = p0;
                    break;
                }
                if (tid < 16) pp->val[h] = h == 0 ? 1u : 0u;
                if (tid == 0) pp->lost = 0;
                wg_lds_barrier();
            }
            if (!probe_p0) probe_runs = true;   // (no sixteen distinct dwords at any of the places tried: bytes that are all alike -- a run is proposed)
            if (probe_p0 && tid < 12) {
                const uint8_t* q = stage + A + probe_p0 + 4 * tid;
                pp->ctx[tid] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            }
            wg_lds_barrier();
        }
        for (uint32_t c = tid; c < nch; c += WG) {
            uint8_t* g = gal + F + 16ull * c;
            if (F == 0 && c == 0 && A != 0) {
                for (uint32_t j = A; j < 16; ++j) g[j] = stage[j];
                if (CNT && cnt_on)   // the unaligned head of the data bytes is part of the sample
                    for (uint32_t j = A; j < 16; ++j) atomicAdd(&CL->hA[wv][stage[j]], 1u);
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(stage + 16u * c);
                *reinterpret_cast<uint4*>(g) = v;
                if (CNT && cnt_on) {
                    // region_histogram's sample: 16-byte chunks are numbered from the first aligned one, 64 of them are a stripe, every
                    // fourth stripe is counted (hist_mode 2: the others too, apart)
                    const uint32_t cq = (uint32_t)(F >> 4) + c - (A ? 1u : 0u);
                    if (((cq >> 6) & 3u) == 0u) cnt_count16(CL->hA[wv], v);
                    else if (hmode == 2u) cnt_count16(CL->hB, v);
                }
                if (PROBE && probe_p0) {
                    const uint32_t h = probe_hash(v.x);
                    if (pp->val[h] == v.x) {   // rare: compare the chunk with the 16 bytes behind that probe
                        const uint64_t rel = F + 16ull * c - A;   // data byte of the chunk's first byte
                        const uint32_t j = pp->idx[h], k = j >> 2, sh = j & 3u;
                        const uint32_t c0 = __builtin_amdgcn_alignbyte(pp->ctx[k + 1], pp->ctx[k], sh), c1 = __builtin_amdgcn_alignbyte(pp->ctx[k + 2], pp->ctx[k + 1], sh);
                        const uint32_t c2 = __builtin_amdgcn_alignbyte(pp->ctx[k + 3], pp->ctx[k + 2], sh), c3 = __builtin_amdgcn_alignbyte(pp->ctx[k + 4], pp->ctx[k + 3], sh);
                        if (rel >= (uint64_t)probe_p0 + j + PROBE_MIN_D && rel < 0xFFFFFFFFull && c0 == v.x && c1 == v.y && c2 == v.z && c3 == v.w)
                            atomicMin(&pp->best, (uint32_t)rel - (probe_p0 + j));
                    }
                }
            }
        }
        const uint32_t rem = endidx & 15u;
        uint8_t keep = 0;
        if ((uint32_t)tid < rem) keep = stage[16u * nch + tid];
        wg_lds_barrier();
        if (nch > 0 && (uint32_t)tid < rem) stage[tid] = keep;
        F += 16ull * nch;
        P += tot;
        // the scan at the top of the next tile contains the barrier that orders these LDS writes
    }
    if (COUNT_ONLY) return P - A;
    wg_lds_barrier();
    if (PROBE && tid == 0) *hint_out = pp->best == 0xFFFFFFFFu ? (probe_runs ? 1u : 0u) : pp->best;
    {   // tail: bytes [F, P) still in LDS
        const uint32_t rem = (uint32_t)(P - F);
        const uint32_t lo = (F == 0) ? A : 0u;
        if ((uint32_t)tid >= lo && (uint32_t)tid < rem) {
            gal[F + tid] = stage[tid];
            if (CNT && cnt_on) atomicAdd(&CL->hA[0][stage[tid]], 1u);   // (the unaligned end: part of the sample too)
        }
    }
    return P - A;
}

template <int ELEM, bool I16ZZ>
struct EncStage
{
    // (int16 zig-zag: the two buffers of I16Pairs; the tile loop of svb_encode_range uses the head of the first, one tile's worst case)
    static constexpr int value = I16ZZ ? 2 * (int)I16Pairs<false, false>::BUF + 2048 : WG * Vpl<ELEM>::value * 4 + 32;   // (+ the control bytes' 2 x 1 KB)
};

template <bool CNT> struct CntLdsOf { typedef CntLds type; };
template <> struct CntLdsOf<false> { struct type { uint32_t none; }; };

#ifndef VBZ_SVBENC_WAVES
#define VBZ_SVBENC_WAVES 1
#endif
template <int ELEM, bool ZZ, bool I16ZZ, bool PROBE, bool CNT = false>
__global__ __launch_bounds__(WG, VBZ_SVBENC_WAVES) void svb_encode_kernel(ReadBatch b, uint32_t hdr, uint32_t strict_cap, uint32_t* period_hint, EncPlan* pre)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[EncStage<ELEM, I16ZZ>::value];
    __shared__ __attribute__((aligned(16))) uint32_t wsum[8];   // (I16Pairs uses two sets of four in turn)
    __shared__ PeriodProbe probe;   // (only the PROBE instantiations refer to it)
    __shared__ __attribute__((aligned(16))) typename CntLdsOf<CNT>::type cntl[1];   // (only the CNT instantiations have one)

    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    if (PROBE && tid == 0) period_hint[r] = 0;
    if (CNT && tid == 0) pre[r].hist_mode = 0;   // whatever becomes of the read, the entropy stage finds a verdict of this launch in its plan
    if (b.gate && b.gate[r] >= GATE_SKIP) {
        if (tid == 0 && b.gate[r] != GATE_SKIP) b.result[r] = b.gate[r];
        return;
    }
    const uint32_t size = b.src_size[r];
    const uint32_t err = svb_encode_check<ELEM, I16ZZ>(size, b.dst_cap[r], hdr, strict_cap);
    if (err) {
        if (tid == 0) b.result[r] = err;
        return;
    }
    const uint32_t n = size / ELEM;
    const uint32_t keyLen = (n + 3u) >> 2;
    const uint8_t* in = b.src + b.src_off[r];
    uint8_t* out = b.dst + b.dst_off[r];
    if (hdr) {
        if (tid < 4) out[tid] = (uint8_t)(size >> (8 * tid));
        out += 4;
    }
    CntLds* const CL = reinterpret_cast<CntLds*>(&cntl[0]);
    uint32_t hmode = 0;
    if (CNT && hdr == 0 && n >= CNT_MIN_VALUES) {
        // (a read of fewer values may give a stream the entropy stage does not cut into regions; a read of 32 K values and more has a
        // data-byte region that is always sampled first: the other bytes are then not counted here -- if the sample turns out not to
        // show enough byte values, the entropy stage counts the region itself)
        hmode = n >= CNT_HIST_SAMPLE_FROM ? 1u : 2u;
        uint32_t* z = &CL->hA[0][0];   // hA and hB are contiguous
        for (uint32_t i = tid; i < 5u * 256u; i += WG) z[i] = 0;
        // (the scan of the first tile holds the barrier that orders these writes)
    }
    const uint64_t bytes = svb_encode_range<ELEM, ZZ, I16ZZ, false, PROBE, CNT>(in, 0, n, out, out + keyLen, stage, wsum, PROBE ? &probe : nullptr,
                                                                                PROBE ? period_hint + r : nullptr, CL, hmode);
    if (tid == 0) b.result[r] = hdr + keyLen + (uint32_t)bytes;
    if (CNT && hmode) {
        wg_lds_barrier();   // the tail's atomics
        EncPlan* P = pre + r;
        P->reg[1].ctable[tid] = CL->hA[0][tid] + CL->hA[1][tid] + CL->hA[2][tid] + CL->hA[3][tid];
        if (hmode == 2u) P->histB[tid] = CL->hB[tid];
        if (tid == 0) P->hist_mode = hmode;
    }
}

// ---- one read on many workgroups ("segments"): for batches of few, large reads --------------------------------------
// A segment is SEG_TILES tiles of values.  seg_first[r] (exclusive scan of the reads' segment counts, seg_first[n_reads] =
// total) maps a workgroup to its (read, segment).  Pass 1 counts the data bytes of every segment, a scan turns the
// counts into data offsets (and gives the read's result), pass 2 encodes every segment at its offset.
constexpr int SEG_TILES = 8;

__device__ __forceinline__ bool seg_locate(const uint32_t* seg_first, uint32_t n_reads, uint32_t g, uint32_t& r, uint32_t& k)
{
    if (g >= seg_first[n_reads]) return false;
    uint32_t lo = 0, hi = n_reads;  // the last read with seg_first[r] <= g
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (seg_first[mid] <= g) lo = mid; else hi = mid;
    }
    r = lo;
    k = g - seg_first[lo];
    return true;
}

// SELF (calls whose segment tables are small: SEG_SELF_MAX): no scan launch between the passes -- a workgroup adds up what the segments
// of its read in front of it have announced by itself (a few loads per thread, the values sit in the L2), and a read's first
// workgroup does what the scan kernel's first thread does (the read's result).  A call of few reads is a chain of short dependent
// launches: each one saved is 4.5 us.  The loads grow with the square of the table (1024 segments: 4 MB out of the L2 per pass).
constexpr uint32_t SEG_SELF_MAX = 1024;
static uint32_t seg_self_max()
{
    static const uint32_t v = [] {
        const char* e = getenv("VBZ_HIP_SEG_SELF_MAX");   // 0: always the scan launches
        return e ? (uint32_t)strtoul(e, nullptr, 10) : SEG_SELF_MAX;
    }();
    return v;
}

// all 256 threads: { sum of arr[lo .. mid), sum of arr[lo .. hi) }, lo <= mid <= hi
__device__ __forceinline__ void seg_sums(const uint32_t* arr, uint32_t lo, uint32_t mid, uint32_t hi, uint64_t* sums_s, uint64_t& before, uint64_t& total)
{
    uint64_t a = 0, t = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += WG) {
        const uint32_t v = arr[i];
        t += v;
        a += i < mid ? v : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        t += __shfl_xor(t, d, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sums_s[2 * (threadIdx.x >> 6)] = a;
        sums_s[2 * (threadIdx.x >> 6) + 1] = t;
    }
    __syncthreads();
    before = sums_s[0] + sums_s[2] + sums_s[4] + sums_s[6];
    total = sums_s[1] + sums_s[3] + sums_s[5] + sums_s[7];
}

template <int ELEM, bool ZZ, bool I16ZZ, bool COUNT_ONLY, bool SELF = false>
__global__ __launch_bounds__(WG) void svb_seg_encode_kernel(ReadBatch b, uint32_t hdr, uint32_t strict_cap, const uint32_t* seg_first,
                                                            uint32_t* seg_bytes, const uint64_t* seg_off)
{
    constexpr int SEG = WG * Vpl<ELEM>::value * SEG_TILES;
    __shared__ __attribute__((aligned(16))) uint8_t stage[COUNT_ONLY ? 16 : EncStage<ELEM, I16ZZ>::value];
    __shared__ __attribute__((aligned(16))) uint32_t wsum[8];   // (I16Pairs uses two sets of four in turn)
    __shared__ uint64_t sums_s[8];
    uint32_t r, k;
    if (!seg_locate(seg_first, b.n_reads, blockIdx.x, r, k)) return;
    const int tid = threadIdx.x;
    if (COUNT_ONLY && tid == 0) seg_bytes[blockIdx.x] = 0;
    uint64_t self_off = 0;
    if (SELF && !COUNT_ONLY) {
        // the data offset of this segment, and (first workgroup of the read) the read's result: svb_seg_encode_scan_kernel's work
        uint64_t total;
        seg_sums(seg_bytes, seg_first[r], blockIdx.x, seg_first[r + 1], sums_s, self_off, total);
        if (k == 0 && tid == 0 && !(b.gate && b.gate[r] == GATE_SKIP)) {
            uint32_t res;
            if (b.gate && b.gate[r] >= E_FIRST) res = b.gate[r];
            else {
                const uint32_t size0 = b.src_size[r];
                res = svb_encode_check<ELEM, I16ZZ>(size0, b.dst_cap[r], hdr, strict_cap);
                if (!res) {
                    res = hdr + ((size0 / ELEM + 3u) >> 2) + (uint32_t)total;
                    if (hdr) {
                        uint8_t* o4 = b.dst + b.dst_off[r];
                        for (int j = 0; j < 4; ++j) o4[j] = (uint8_t)(size0 >> (8 * j));
                    }
                }
            }
            b.result[r] = res;
        }
    }
    if (b.gate && b.gate[r] >= GATE_SKIP) return;
    const uint32_t size = b.src_size[r];
    if (svb_encode_check<ELEM, I16ZZ>(size, b.dst_cap[r], hdr, strict_cap)) return;
    const uint32_t n = size / ELEM;
    const uint32_t keyLen = (n + 3u) >> 2;
    const uint32_t first = k * (uint32_t)SEG;
    const uint32_t end = n - first > (uint32_t)SEG ? first + SEG : n;
    if (first >= n) return;  // the single (empty) segment of an empty read
    const uint8_t* in = b.src + b.src_off[r];
    uint8_t* out = b.dst + b.dst_off[r] + hdr;
    if (COUNT_ONLY) {
        const uint64_t bytes = svb_encode_range<ELEM, ZZ, I16ZZ, true>(in, first, end, nullptr, nullptr, stage, wsum);
        if (tid == 0) seg_bytes[blockIdx.x] = (uint32_t)bytes;
    } else {
        (void)svb_encode_range<ELEM, ZZ, I16ZZ, false>(in, first, end, out, out + keyLen + (SELF ? self_off : seg_off[blockIdx.x]), stage, wsum);
    }
}

// one workgroup per read: data offsets of its segments (exclusive scan of their byte counts) and the read's result
template <int ELEM, bool I16ZZ>
__global__ __launch_bounds__(WG) void svb_seg_encode_scan_kernel(ReadBatch b, uint32_t hdr, uint32_t strict_cap, const uint32_t* seg_first,
                                                                 const uint32_t* seg_bytes, uint64_t* seg_off)
{
    __shared__ uint32_t wsum[4];
    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t s0 = seg_first[r], s1 = seg_first[r + 1];
    uint64_t carry = 0;
    for (uint32_t base = s0; base < s1; base += WG) {
        const uint32_t i = base + (uint32_t)tid;
        const uint32_t v = i < s1 ? seg_bytes[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(v, wsum, tot);
        if (i < s1) seg_off[i] = carry + ex;
        carry += tot;
    }
    if (tid == 0 && !(b.gate && b.gate[r] == GATE_SKIP)) {
        uint32_t res;
        if (b.gate && b.gate[r] >= E_FIRST) res = b.gate[r];
        else {
            const uint32_t size = b.src_size[r];
            res = svb_encode_check<ELEM, I16ZZ>(size, b.dst_cap[r], hdr, strict_cap);
            if (!res) {
                const uint32_t n = size / ELEM;
                res = hdr + ((n + 3u) >> 2) + (uint32_t)carry;
                if (hdr) {
                    uint8_t* out = b.dst + b.dst_off[r];
                    for (int j = 0; j < 4; ++j) out[j] = (uint8_t)(size >> (8 * j));
                }
            }
        }
        b.result[r] = res;
    }
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
// ---- pairs of whole int16 zig-zag tiles, decoded two loads ahead (round 6) --------------------------------------------------------------
// The tile loop of svb_decode_range asks memory twice per tile, one request behind the other: the control bytes, then -- where the
// scan of their lengths says the tile's data bytes start -- the data bytes; with eight workgroups per CU that chain of round trips, not
// the 21.4 GB the launch moves, is what its 4.7 ms per 65 536 reads were (the fewer instructions and barriers of rounds 2 - 5 all
// measured +- 0).  Whole tiles of int16 values with one- and two-byte codes -- everything but the ends of a read this library or the
// reference wrote -- go through this loop, two tiles per trip, as a pipeline: a trip PLANS the pair after the one it decodes (scan of
// the lengths the control bytes announce: where its data bytes stand), REQUESTS that pair's data bytes into registers and the control
// bytes of the pair after it, then decodes the pair whose bytes the trip before left in LDS, and last moves the requested bytes into
// the other stage buffer.  Every request has a whole trip to arrive, and a trip has two barriers (ten before, for two tiles).  The
// lengths of both tiles ride in the halves of one register through one scan; the delta sums need a scan each (16-bit wrap-around).
// A pair that is not of that kind (3- or 4-byte codes, a stream shorter than announced, more bytes than a stage buffer holds) ends
// the loop before anything of it is touched, and the tile loop -- which decides every verdict -- goes on from there.
struct I16DecPairs
{
    static constexpr uint32_t TILE = WG * 8, BUF = 2 * WG * 8 * 3 / 2 + 48;   // a stage buffer: two tiles at 3/2 bytes per value + alignment slack
    static constexpr uint32_t WS_LEN = 0, WS_DELTA = 8, WS_FLAG = 24, WS_WORDS = 28;   // LDS words: 2 x 4 length sums, 2 x 8 delta sums, the flag
    const uint8_t* in;        // control bytes
    const uint8_t* data;
    uint32_t dataBytes;
    uint8_t* out;             // 16-byte aligned
    uint8_t* stage;           // two buffers of BUF bytes
    uint32_t* ws;             // WS_WORDS words, 16-byte aligned

    static __device__ __forceinline__ uint32_t announced(uint32_t k) { return 8u + (uint32_t)__popc(k & 0x5555u) + 2u * (uint32_t)__popc(k & 0xAAAAu); }

    // a lane's eight values (codes 0 / 1) from the bytes at buf[o ...]: inclusive sums of the un-zig-zagged deltas (mod 2^32; the caller keeps 16 bits)
    static __device__ __forceinline__ uint32_t decode8(const uint8_t* buf, uint32_t o, uint32_t keybits, uint32_t s[8])
    {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t two = (keybits >> (2 * k)) & 1u;
            const uint32_t lo = buf[o], hi = buf[o + 1];
            uint32_t v = lo | (two ? hi << 8 : 0u);
            o += 1u + two;
            v = (v >> 1) ^ (0u - (v & 1u));   // sse3.h:516-523
            acc += v;
            s[k] = acc;
        }
        return acc;
    }

    // pairs from t0 on while two whole tiles are left; pos / run as in svb_decode_range.  Returns the first value not decoded (a multiple of
    // the tile size from t0).  All 256 threads; ends with a barrier.
    __device__ __forceinline__ uint32_t run(uint32_t t0_in, uint32_t end_in, uint64_t& pos_io, uint32_t& run_io)
    {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t m0 = wv > 0 ? 0xFFFFFFFFu : 0u, m1 = wv > 1 ? 0xFFFFFFFFu : 0u, m2 = wv > 2 ? 0xFFFFFFFFu : 0u;
        const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)end_in);
        uint32_t tn = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0_in);   // the pair the next trip plans
        uint64_t posn = I16Pairs<false, false>::uniform64(pos_io);              // ... and where its data bytes start
        uint32_t rn = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_io);
        auto load_keys = [&](uint32_t t, uint32_t& ka, uint32_t& kb) {
            const uint8_t* kp = in + ((t + (uint32_t)tid * 8u) >> 2);
            uint16_t a, b;
            __builtin_memcpy(&a, kp, 2);
            __builtin_memcpy(&b, kp + TILE / 4, 2);
            ka = a;
            kb = b;
        };
        bool more = end - tn >= 2u * TILE, have = false;
        uint32_t kna = 0, knb = 0;
        if (more) load_keys(tn, kna, knb);
        if (tid == 0) ws[WS_FLAG] = 0;
        uint32_t cur = 0, par = 0;
        uint32_t kca = 0, kcb = 0, exC = 0, totC = 0, misC = 0, tc = 0;   // the pair in the stage
        while (more || have) {
            uint32_t Ln = 0, incn = 0;
            if (more) {
                Ln = announced(kna) | (announced(knb) << 16);
                incn = wave_incl_scan_u32(Ln);
                if (lane == 63) ws[WS_LEN + par * 4 + wv] = incn;
                if (__any(((kna | knb) & 0xAAAAu) != 0) && lane == 0) ws[WS_FLAG] = 1;   // codes of 3 or 4 bytes: not for this loop
            }
            wg_lds_barrier();
            uint32_t exN = 0, totN = 0, misN = 0, kn2a = 0, kn2b = 0;
            uint4 d0 = make_uint4(0u, 0u, 0u, 0u), d1 = d0;
            bool more2 = false;
            if (more) {
                const uint4 s4 = *reinterpret_cast<const uint4*>(ws + WS_LEN + par * 4);
                totN = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s4.x + s4.y + s4.z + s4.w));
                exN = (s4.x & m0) + (s4.y & m1) + (s4.z & m2) + incn - Ln;
                const uint32_t bad = (uint32_t)__builtin_amdgcn_readfirstlane((int)ws[WS_FLAG]);
                const uint32_t bytes = (totN & 0xFFFFu) + (totN >> 16);
                misN = (uint32_t)((uintptr_t)(data + posn) & 15u);
                if (bad || posn + bytes > dataBytes || misN + bytes + 16u > BUF) {
                    more = false;   // (workgroup-uniform) the tile loop takes it from here
                } else {
                    const uint8_t* ga = data + posn - misN;
                    const uint32_t nch = (misN + bytes + 15u) >> 4;
                    if ((uint32_t)tid < nch) d0 = *reinterpret_cast<const uint4*>(ga + 16u * (uint32_t)tid);
                    if ((uint32_t)tid + WG < nch) d1 = *reinterpret_cast<const uint4*>(ga + 16u * ((uint32_t)tid + WG));
                    more2 = end - tn >= 4u * TILE;
                    if (more2) load_keys(tn + 2u * TILE, kn2a, kn2b);
                }
            }
            uint32_t sA[8], sB[8], accA = 0, accB = 0, incA = 0, incB = 0;
            if (have) {
                const uint8_t* buf = stage + cur * BUF;
                accA = decode8(buf, misC + (exC & 0xFFFFu), kca, sA);
                accB = decode8(buf, misC + (totC & 0xFFFFu) + (exC >> 16), kcb, sB);
                incA = wave_incl_scan_u32(accA);
                incB = wave_incl_scan_u32(accB);
                if (lane == 63) {
                    ws[WS_DELTA + par * 8 + wv] = incA;
                    ws[WS_DELTA + par * 8 + 4 + wv] = incB;
                }
            }
            wg_lds_barrier();
            if (have) {
                const uint4 a4 = *reinterpret_cast<const uint4*>(ws + WS_DELTA + par * 8), b4 = *reinterpret_cast<const uint4*>(ws + WS_DELTA + par * 8 + 4);
                const uint32_t totA = a4.x + a4.y + a4.z + a4.w, totB = b4.x + b4.y + b4.z + b4.w;
                const uint32_t baseA = rn + (a4.x & m0) + (a4.y & m1) + (a4.z & m2) + incA - accA;
                const uint32_t baseB = rn + totA + (b4.x & m0) + (b4.y & m1) + (b4.z & m2) + incB - accB;
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = ((baseA + sA[2 * k]) & 0xFFFFu) | ((baseA + sA[2 * k + 1]) << 16);
                uint8_t* op = out + ((size_t)tc + (size_t)tid * 8) * 2;
                *reinterpret_cast<uint4*>(op) = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = ((baseB + sB[2 * k]) & 0xFFFFu) | ((baseB + sB[2 * k + 1]) << 16);
                *reinterpret_cast<uint4*>(op + TILE * 2) = make_uint4(w[0], w[1], w[2], w[3]);
                rn = (uint32_t)__builtin_amdgcn_readfirstlane((int)(rn + totA + totB));
            }
            have = more;
            if (more) {   // the requested bytes into the other buffer: the next trip decodes them
                cur ^= 1u;
                uint8_t* nb = stage + cur * BUF;
                const uint32_t bytes = (totN & 0xFFFFu) + (totN >> 16), nch = (misN + bytes + 15u) >> 4;
                if ((uint32_t)tid < nch) *reinterpret_cast<uint4*>(nb + 16u * (uint32_t)tid) = d0;
                if ((uint32_t)tid + WG < nch) *reinterpret_cast<uint4*>(nb + 16u * ((uint32_t)tid + WG)) = d1;
                kca = kna;
                kcb = knb;
                exC = exN;
                totC = totN;
                misC = misN;
                tc = tn;
                posn += bytes;
                tn += 2u * TILE;
                kna = kn2a;
                knb = kn2b;
                more = more2;
            }
            par ^= 1u;
        }
        wg_lds_barrier();
        pos_io = posn;
        run_io = rn;
        return tn;
    }
};


// (per-read validation of the decoders: svb_decode_check in svb_wave.h, shared with the wave decoder)
// MODE 0: decode values [first, end) of a stream of `count` values whose data bytes start at data[pos] and whose delta
//         chain stands at `run`; store them.  MODE 1: only add up the data bytes the control bytes announce.
//         MODE 2: decode without storing (the total of the deltas is wanted).  pos / run are updated; returns false when
//         the stream is shorter than its control bytes claim.  All 256 threads.
template <int ELEM, bool ZZ, bool I16ZZ, int MODE>
__device__ __forceinline__ bool svb_decode_range(const uint8_t* in, const uint8_t* data, uint32_t dataBytes, uint32_t count, uint32_t first,
                                                 uint32_t end, uint64_t& pos_io, uint32_t& run_io, uint8_t* out, uint8_t* stage, uint32_t* wsum)
{
    constexpr int VPL = Vpl<ELEM>::value;
    constexpr int TILE = WG * VPL;
    const int tid = threadIdx.x;
    const bool out_aligned = (((uintptr_t)out) & 15u) == 0;
    const uint32_t* stage32 = reinterpret_cast<const uint32_t*>(stage);
    uint64_t pos = pos_io;   // data bytes consumed so far
    uint32_t run = run_io;   // running value of the delta chain
    bool good = true;
    uint32_t t_start = first;
    if (I16ZZ && MODE == 0 && out_aligned && end - first >= 2u * (uint32_t)TILE) {   // pairs of whole tiles: I16DecPairs
        I16DecPairs dp = { in, data, dataBytes, out, stage, wsum };
        t_start = dp.run(first, end, pos, run);
    }
    for (uint32_t t0 = t_start; t0 < end; t0 += TILE) {
        const uint32_t i0 = t0 + (uint32_t)tid * VPL;
        const int valid = i0 >= end ? 0 : (end - i0 >= (uint32_t)VPL ? VPL : (int)(end - i0));
        uint32_t keybits = 0;
        if (valid > 0) {
            const uint8_t* kp = in + (i0 >> 2);
            keybits = kp[0];
            if (VPL == 8 && valid > 4) keybits |= (uint32_t)kp[1] << 8;
        }
        // bytes of the lane's values: each 2-bit code is length - 1; the sum of the 2-bit fields of a word is
        // (word & 0x5555) + ((word >> 1) & 0x5555) folded -- two population counts
        const uint32_t kb = valid == VPL ? keybits : (keybits & ((1u << (2 * valid)) - 1u));
        const uint32_t L = (uint32_t)valid + (uint32_t)__popc(kb & 0x5555u) + 2u * (uint32_t)__popc(kb & 0xAAAAu);
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(L, wsum, tot);
        if (MODE == 1) {
            pos += tot;
            continue;
        }
        if (pos + tot > dataBytes) {  // stream shorter than its control bytes claim
            good = false;
            break;
        }
        // stage the tile's data bytes: aligned 16-byte chunks, coalesced
        const uint8_t* g0 = data + pos;
        const uint32_t mis = (uint32_t)((uintptr_t)g0 & 15u);
        const uint8_t* ga = g0 - mis;
        const uint32_t nch = (mis + tot + 15u) >> 4;
        for (uint32_t c = tid; c < nch; c += WG)
            *reinterpret_cast<uint4*>(stage + 16u * c) = *reinterpret_cast<const uint4*>(ga + 16ull * c);
        wg_lds_barrier();
        uint32_t o = mis + ex;
        // int16 zig-zag path: SIMD body vs scalar tail of the reference (sse3.h:494-540 vs 542-572)
        const bool body = I16ZZ && ((i0 >> 3) < (count >> 3)) && (dataBytes - (pos + ex) >= 32u);
        uint32_t s[VPL];
        uint32_t acc = 0;
        // full tile of one- and two-byte codes (what the encoder writes for int16): values stay below 2^16, where the
        // body and the tail of the reference agree, so the lane just picks up its bytes
        if (I16ZZ && end - t0 >= (uint32_t)TILE && !__any((keybits & 0xAAAAu) != 0)) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const uint32_t two = (keybits >> (2 * k)) & 1u;
                const uint32_t lo = stage[o], hi = stage[o + 1];
                uint32_t v = lo | (two ? hi << 8 : 0u);
                o += 1u + two;
                v = (v >> 1) ^ (0u - (v & 1u));
                acc += v;
                s[k] = acc;
            }
        } else {
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
            if (ZZ) {
                if (I16ZZ && body) v &= 0xFFFFu;  // keep the low 16 bits (sse3.h:510-514)
                v = (v >> 1) ^ (0u - (v & 1u));
                acc += v;
                s[k] = acc;
            } else {
                s[k] = v;
            }
        }
        }
        uint32_t base = 0;
        if (ZZ) {
            uint32_t ttot;
            base = run + block_excl_scan_u32(acc, wsum, ttot);
            run += ttot;
        } else {
            wg_lds_barrier();  // stage is overwritten by the next tile
        }
        if (MODE == 2) {
            pos += tot;
            continue;
        }
        if (valid == VPL && out_aligned) {
            uint32_t w[4];
            if (ELEM == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = base + s[k % VPL];
            } else if (ELEM == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    w[k] = ((base + s[(2 * k) % VPL]) & 0xFFFFu) | ((base + s[(2 * k + 1) % VPL]) << 16);
            } else {
                w[2] = w[3] = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    w[k] = ((base + s[(4 * k) % VPL]) & 0xFFu) | (((base + s[(4 * k + 1) % VPL]) & 0xFFu) << 8) |
                           (((base + s[(4 * k + 2) % VPL]) & 0xFFu) << 16) | ((base + s[(4 * k + 3) % VPL]) << 24);
            }
            if (ELEM == 1) {
                *reinterpret_cast<uint2*>(out + (size_t)i0) = make_uint2(w[0], w[1]);
            } else {
                *reinterpret_cast<uint4*>(out + (size_t)i0 * ELEM) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < VPL; ++k)
                if (k < valid) store_elem(out + (size_t)(i0 + k) * ELEM, ELEM, base + s[k]);
        }
        pos += tot;
    }
    pos_io = pos;
    run_io = run;
    return good;
}

#ifndef VBZ_SVBDEC_WAVES
#define VBZ_SVBDEC_WAVES 1
#endif
template <int ELEM, bool ZZ, bool I16ZZ>
__global__ __launch_bounds__(WG, VBZ_SVBDEC_WAVES) void svb_decode_kernel(ReadBatch b)
{
    constexpr int STAGE = I16ZZ ? 2 * (int)I16DecPairs::BUF : WG * Vpl<ELEM>::value * 4 + 48;   // (int16 zig-zag: the two buffers of I16DecPairs; the tile loop's 8240 bytes fit)
    static_assert(STAGE >= WG * Vpl<ELEM>::value * 4 + 48, "the tile loop's stage");
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t wsum[I16ZZ ? (int)I16DecPairs::WS_WORDS : 4];

    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    if (b.gate && b.gate[r] >= GATE_SKIP) {
        if (tid == 0 && b.gate[r] != GATE_SKIP) b.result[r] = b.gate[r];
        return;
    }
    const uint32_t in_size = b.src_size[r];
    if (in_size >= E_FIRST) {  // the previous stage failed for this read
        if (tid == 0) b.result[r] = in_size;
        return;
    }
    const uint32_t out_size = b.dst_cap[r];  // exact decoded byte count
    uint32_t res;
    if (svb_decode_check<ELEM, I16ZZ>(in_size, out_size, res)) {
        if (tid == 0) b.result[r] = res;
        return;
    }
    const uint32_t count = out_size / ELEM;
    const uint32_t keyLen = (count + 3u) >> 2;
    const uint8_t* in = b.src + b.src_off[r];
    const uint32_t dataBytes = in_size - keyLen;
    uint64_t pos = 0;
    uint32_t run = 0;
    const bool good = svb_decode_range<ELEM, ZZ, I16ZZ, 0>(in, in + keyLen, dataBytes, count, 0, count, pos, run, b.dst + b.dst_off[r], stage, wsum);
    if (tid == 0) b.result[r] = (!good || pos != dataBytes) ? E_STREAM : count * ELEM;
}

// ---- segmented decode: pass 1 adds up the data bytes each segment's control bytes announce; a scan gives every
// segment its data offset and decides the read's verdict (the stream is good iff the announced total equals the data
// bytes present -- the same predicate streamvbyte_validate_stream computes); zig-zag streams then need the delta total
// of every segment (pass 2, decode without storing) and a second scan before the storing pass.
// SELF (see svb_seg_encode_kernel): no scan launches -- MODE 2 / 0 workgroups add up the announced lengths (seg_val, left by MODE 1)
// in front of them and over the whole read (the verdict: good iff the total equals the bytes present), MODE 2 leaves its delta
// totals in seg_run_io and MODE 0 adds those up too; a read's first MODE 0 workgroup writes the read's result.
template <int ELEM, bool ZZ, bool I16ZZ, int MODE, bool SELF = false>
__global__ __launch_bounds__(WG) void svb_seg_decode_kernel(ReadBatch b, const uint32_t* seg_first, uint32_t* seg_val, const uint64_t* seg_pos,
                                                            uint32_t* seg_run_io)
{
    const uint32_t* seg_run = seg_run_io;
    constexpr int SEG = WG * Vpl<ELEM>::value * SEG_TILES;
    constexpr int STAGE = MODE == 1 ? 16 : ((I16ZZ && MODE == 0) ? 2 * (int)I16DecPairs::BUF : WG * Vpl<ELEM>::value * 4 + 48);
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t wsum[(I16ZZ && MODE == 0) ? (int)I16DecPairs::WS_WORDS : 4];
    __shared__ uint64_t sums_s[8];
    uint32_t r, k;
    if (!seg_locate(seg_first, b.n_reads, blockIdx.x, r, k)) return;
    const int tid = threadIdx.x;
    if (MODE != 0 && tid == 0) (SELF && MODE == 2 ? seg_run_io : seg_val)[blockIdx.x] = 0;
    uint64_t self_pos = 0, self_total = 0, self_run = 0;
    if (SELF && MODE != 1) {
        seg_sums(seg_val, seg_first[r], blockIdx.x, seg_first[r + 1], sums_s, self_pos, self_total);
        if (MODE == 0 && ZZ) {
            uint64_t dummy;
            seg_sums(seg_run, seg_first[r], blockIdx.x, seg_first[r + 1], sums_s, self_run, dummy);
        }
        if (MODE == 0 && k == 0 && tid == 0 && !(b.gate && b.gate[r] == GATE_SKIP)) {   // svb_seg_decode_scan_kernel<.., VERDICT>'s first thread
            uint32_t res0;
            if (b.gate && b.gate[r] >= E_FIRST) res0 = b.gate[r];
            else if (b.src_size[r] >= E_FIRST) res0 = b.src_size[r];
            else if (!svb_decode_check<ELEM, I16ZZ>(b.src_size[r], b.dst_cap[r], res0)) {
                const uint32_t count0 = b.dst_cap[r] / ELEM;
                const uint32_t dataBytes0 = b.src_size[r] - ((count0 + 3u) >> 2);
                res0 = self_total == dataBytes0 ? count0 * ELEM : E_STREAM;
            }
            b.result[r] = res0;
        }
    }
    if (b.gate && b.gate[r] >= GATE_SKIP) return;
    const uint32_t in_size = b.src_size[r];
    if (in_size >= E_FIRST) return;
    const uint32_t out_size = b.dst_cap[r];
    uint32_t res;
    if (svb_decode_check<ELEM, I16ZZ>(in_size, out_size, res)) return;
    if (SELF) {
        if (MODE != 1 && self_total != (uint64_t)(in_size - ((out_size / ELEM + 3u) >> 2))) return;   // the stream is malformed
    } else if (MODE != 1 && b.result[r] >= E_FIRST) return;  // the scan found the stream malformed
    const uint32_t count = out_size / ELEM;
    const uint32_t keyLen = (count + 3u) >> 2;
    const uint32_t first = k * (uint32_t)SEG;
    if (first >= count) return;
    const uint32_t end = count - first > (uint32_t)SEG ? first + SEG : count;
    const uint8_t* in = b.src + b.src_off[r];
    const uint32_t dataBytes = in_size - keyLen;
    uint64_t pos = MODE == 1 ? 0 : (SELF ? self_pos : seg_pos[blockIdx.x]);
    uint32_t run = (MODE == 0 && ZZ) ? (SELF ? (uint32_t)self_run : seg_run[blockIdx.x]) : 0u;
    const uint64_t pos0 = pos;
    (void)svb_decode_range<ELEM, ZZ, I16ZZ, MODE>(in, in + keyLen, dataBytes, count, first, end, pos, run, b.dst + b.dst_off[r], stage, wsum);
    if (MODE == 1 && tid == 0) seg_val[blockIdx.x] = (uint32_t)(pos - pos0);
    if (MODE == 2 && tid == 0) (SELF ? seg_run_io : seg_val)[blockIdx.x] = run;
}

// one workgroup per read.  VERDICT: exclusive scan of the announced data bytes -> seg_pos, and the read's result;
// otherwise: exclusive scan (mod 2^32) of the delta totals -> seg_run.
template <int ELEM, bool I16ZZ, bool VERDICT>
__global__ __launch_bounds__(WG) void svb_seg_decode_scan_kernel(ReadBatch b, const uint32_t* seg_first, const uint32_t* seg_val, uint64_t* seg_pos,
                                                                 uint32_t* seg_run)
{
    __shared__ uint32_t wsum[4];
    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t s0 = seg_first[r], s1 = seg_first[r + 1];
    uint64_t carry = 0;
    for (uint32_t base = s0; base < s1; base += WG) {
        const uint32_t i = base + (uint32_t)tid;
        const uint32_t v = i < s1 ? seg_val[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(v, wsum, tot);
        if (i < s1) {
            if (VERDICT) seg_pos[i] = carry + ex;
            else seg_run[i] = (uint32_t)carry + ex;
        }
        carry += tot;
    }
    if (VERDICT && tid == 0 && !(b.gate && b.gate[r] == GATE_SKIP)) {
        uint32_t res;
        if (b.gate && b.gate[r] >= E_FIRST) res = b.gate[r];
        else if (b.src_size[r] >= E_FIRST) res = b.src_size[r];
        else if (!svb_decode_check<ELEM, I16ZZ>(b.src_size[r], b.dst_cap[r], res)) {
            const uint32_t count = b.dst_cap[r] / ELEM;
            const uint32_t dataBytes = b.src_size[r] - ((count + 3u) >> 2);
            res = carry == dataBytes ? count * ELEM : E_STREAM;
        }
        b.result[r] = res;
    }
}

// ------------------------------------------------------------------------------------------------
// v1 nibble ("half") codec for 1-byte integers: reference vbz/v1/vbz_streamvbyte_impl.h:20-216
//   code 0: value 0, no data; 1: one nibble; 2: two nibbles; 3: four nibbles (low 16 bits of the value);
//   nibbles are appended low nibble first.  Same tiling as the byte codec, but offsets are in nibbles: a lane
//   assembles its <= 32 nibbles in registers and ORs them into a zeroed LDS stage (two lanes can share a byte).
// ------------------------------------------------------------------------------------------------
template <bool ZZ>
__global__ __launch_bounds__(WG) void svb_half_encode_kernel(ReadBatch b, uint32_t hdr)
{
    constexpr int VPL = 8;
    constexpr int TILE = WG * VPL;
    constexpr int STAGE_W = (TILE * 2 + 64) / 4;  // a tile emits at most 4 nibbles per value
    __shared__ __attribute__((aligned(16))) uint32_t stage[STAGE_W];
    __shared__ uint32_t wsum[4];

    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    if (b.gate && b.gate[r] >= GATE_SKIP) {
        if (tid == 0 && b.gate[r] != GATE_SKIP) b.result[r] = b.gate[r];
        return;
    }
    const uint32_t n = b.src_size[r];  // one byte per value
    const uint32_t cap = b.dst_cap[r];
    const uint32_t keyLen = (n + 3u) >> 2;
    {   // the reference requires the worst case of the byte codec (vbz/vbz.cpp:171-174)
        const uint64_t worst = (uint64_t)keyLen + 4ull * n + hdr;
        if (worst > cap) {
            if (tid == 0) b.result[r] = worst > 0xFFFFFFF0ull ? E_INPUT_SIZE : E_DESTINATION_SIZE;
            return;
        }
    }
    const uint8_t* in = b.src + b.src_off[r];
    uint8_t* out = b.dst + b.dst_off[r];
    if (hdr) {
        if (tid < 4) out[tid] = (uint8_t)(n >> (8 * tid));
        out += 4;
    }
    uint8_t* keys = out;
    uint8_t* data = out + keyLen;
    for (int i = tid; i < STAGE_W; i += WG) stage[i] = 0;
    wg_lds_barrier();

    uint64_t flushed = 0;  // data bytes already written
    uint32_t held = 0;     // nibbles held in stage[] (they start at nibble 0 of the stage)
    for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
        const uint32_t i0 = t0 + (uint32_t)tid * VPL;
        const int valid = i0 >= n ? 0 : (n - i0 >= (uint32_t)VPL ? VPL : (int)(n - i0));
        int32_t prev = 0;
        if (ZZ && valid > 0 && i0 > 0) prev = load_elem(in + i0 - 1, 1);
        uint32_t keybits = 0, nibs = 0;
        unsigned __int128 acc = 0;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            if (k < valid) {
                const int32_t x = load_elem(in + i0 + k, 1);
                uint32_t u = (uint32_t)x;
                if (ZZ) {
                    const uint32_t d = (uint32_t)x - (uint32_t)prev;  // streamvbyte zigzag_delta_encode
                    u = (d << 1) ^ (uint32_t)((int32_t)d >> 31);
                    prev = x;
                }
                const uint32_t code = u == 0 ? 0u : (u < 16u ? 1u : (u < 256u ? 2u : 3u));
                const uint32_t nn = (1u << code) >> 1;
                const uint32_t bits = nn == 4 ? (u & 0xFFFFu) : (u & ((1u << (4 * nn)) - 1u));
                acc |= (unsigned __int128)bits << (4 * nibs);
                nibs += nn;
                keybits |= code << (2 * k);
            }
        }
        if (valid > 0) {
            uint8_t* kp = keys + (i0 >> 2);
            kp[0] = (uint8_t)keybits;
            if (valid > 4) kp[1] = (uint8_t)(keybits >> 8);
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(nibs, wsum, tot);
        if (nibs) {
            const uint32_t o = held + ex;  // nibble offset in the stage
            const uint32_t sh = 4 * (o & 7);
            const unsigned __int128 lo = acc << sh;
            const uint32_t top = sh ? (uint32_t)(acc >> (128 - sh)) : 0u;
            const uint32_t w[5] = { (uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)(lo >> 64), (uint32_t)(lo >> 96), top };
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (w[k]) atomicOr(&stage[(o >> 3) + k], w[k]);
        }
        wg_lds_barrier();
        // write the complete 16-byte chunks, keep the rest (including a half-filled byte) at the front
        const uint32_t endn = held + tot;
        const uint32_t nch = endn >> 5;  // 32 nibbles per chunk
        for (uint32_t c = tid; c < nch; c += WG) {
            const uint4 v = *reinterpret_cast<const uint4*>(stage + 4 * c);
            __builtin_memcpy(data + flushed + 16ull * c, &v, 16);
        }
        const uint32_t remw = ((endn & 31u) + 7u) >> 3;  // words still in use behind the chunks
        uint32_t keep = 0;
        if ((uint32_t)tid < remw) keep = stage[4 * nch + tid];
        wg_lds_barrier();
        for (uint32_t i = tid; i < ((endn + 7u) >> 3) + 1u && i < (uint32_t)STAGE_W; i += WG) stage[i] = 0;
        wg_lds_barrier();
        if ((uint32_t)tid < remw) stage[tid] = keep;
        flushed += 16ull * nch;
        held = endn & 31u;
        wg_lds_barrier();
    }
    const uint32_t tailBytes = (held + 1u) >> 1;
    if ((uint32_t)tid < tailBytes) data[flushed + tid] = (uint8_t)(stage[tid >> 2] >> (8 * (tid & 3)));
    if (tid == 0) b.result[r] = hdr + keyLen + (uint32_t)flushed + tailBytes;
}

template <bool ZZ>
__global__ __launch_bounds__(WG) void svb_half_decode_kernel(ReadBatch b)
{
    constexpr int VPL = 8;
    constexpr int TILE = WG * VPL;
    constexpr int STAGE = TILE * 2 + 64;
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGE];
    __shared__ uint32_t wsum[4];

    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    if (b.gate && b.gate[r] >= GATE_SKIP) {
        if (tid == 0 && b.gate[r] != GATE_SKIP) b.result[r] = b.gate[r];
        return;
    }
    const uint32_t in_size = b.src_size[r];
    if (in_size >= E_FIRST) {
        if (tid == 0) b.result[r] = in_size;
        return;
    }
    const uint32_t count = b.dst_cap[r];  // exact decoded byte count = value count
    // half validate_stream (vbz/v1/vbz_streamvbyte_impl.h)
    if (in_size == 0 || count == 0) {
        if (tid == 0) b.result[r] = (in_size == count) ? 0u : E_STREAM;
        return;
    }
    const uint32_t keyLen = (count + 3u) >> 2;
    if (keyLen > in_size) {
        if (tid == 0) b.result[r] = E_STREAM;
        return;
    }
    const uint8_t* in = b.src + b.src_off[r];
    const uint8_t* data = in + keyLen;
    uint8_t* out = b.dst + b.dst_off[r];
    const uint32_t dataBytes = in_size - keyLen;
    const uint32_t* stage32 = reinterpret_cast<const uint32_t*>(stage);

    uint64_t posn = 0;  // nibbles consumed so far
    uint32_t run = 0;   // running value of the delta chain
    bool bad = false;
    for (uint32_t t0 = 0; t0 < count; t0 += TILE) {
        const uint32_t i0 = t0 + (uint32_t)tid * VPL;
        const int valid = i0 >= count ? 0 : (count - i0 >= (uint32_t)VPL ? VPL : (int)(count - i0));
        uint32_t keybits = 0;
        if (valid > 0) {
            const uint8_t* kp = in + (i0 >> 2);
            keybits = kp[0];
            if (valid > 4) keybits |= (uint32_t)kp[1] << 8;
        }
        uint32_t nibs = 0;
#pragma unroll
        for (int k = 0; k < VPL; ++k)
            if (k < valid) nibs += (1u << ((keybits >> (2 * k)) & 3u)) >> 1;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(nibs, wsum, tot);
        if (((posn + tot + 1u) >> 1) > dataBytes) {  // stream shorter than its control bytes claim
            bad = true;
            break;
        }
        // stage the bytes that hold nibbles [posn, posn + tot): aligned 16-byte chunks, coalesced
        const uint8_t* g0 = data + (posn >> 1);
        const uint32_t mis = (uint32_t)((uintptr_t)g0 & 15u);
        const uint8_t* ga = g0 - mis;
        const uint32_t nbytes = (uint32_t)(((posn + tot + 1u) >> 1) - (posn >> 1));
        const uint32_t nch = (mis + nbytes + 15u) >> 4;
        for (uint32_t c = tid; c < nch; c += WG)
            *reinterpret_cast<uint4*>(stage + 16u * c) = *reinterpret_cast<const uint4*>(ga + 16ull * c);
        wg_lds_barrier();
        uint32_t o = 2u * mis + (uint32_t)(posn & 1u) + ex;  // nibble offset in the stage
        uint32_t s[VPL];
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            uint32_t v = 0;
            if (k < valid) {
                const uint32_t nn = (1u << ((keybits >> (2 * k)) & 3u)) >> 1;
                if (nn) {
                    const uint32_t w0 = stage32[o >> 3];
                    const uint32_t w1 = stage32[(o >> 3) + 1];
                    v = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (4u * (o & 7u)));
                    v &= 0xFFFFu >> (16u - 4u * nn);
                    o += nn;
                }
            }
            if (ZZ) {
                v = (v >> 1) ^ (0u - (v & 1u));
                acc += v;
                s[k] = acc;
            } else {
                s[k] = v;
            }
        }
        uint32_t base = 0;
        if (ZZ) {
            uint32_t ttot;
            base = run + block_excl_scan_u32(acc, wsum, ttot);
            run += ttot;
        } else {
            wg_lds_barrier();  // stage is overwritten by the next tile
        }
#pragma unroll
        for (int k = 0; k < VPL; ++k)
            if (k < valid) out[i0 + k] = (uint8_t)(base + s[k]);
        posn += tot;
    }
    if (tid == 0) b.result[r] = (bad || ((posn + 1u) >> 1) != dataBytes) ? E_STREAM : count;
}

template <typename K>
hipError_t launch1(K kernel, const ReadBatch& b, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(kernel, dim3(b.n_reads), dim3(WG), 0, s, b);
    return hipGetLastError();
}

}  // namespace

bool svb_encode_fills_plans(int integer_size, bool zigzag, bool half) { return integer_size == 2 && zigzag && !half; }

hipError_t launch_svb_encode(const ReadBatch& b, int integer_size, bool zigzag, uint32_t hdr, bool strict_cap, bool half, uint32_t* period_hint,
                             void* plans, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    dim3 g(b.n_reads), t(WG);
    if (half) {  // v1, 1-byte integers: the nibble codec
        if (integer_size != 1) return hipErrorInvalidValue;
        if (period_hint) (void)hipMemsetAsync(period_hint, 0, 4ull * b.n_reads, s);   // (the nibble stream is not probed)
        if (zigzag) hipLaunchKernelGGL((svb_half_encode_kernel<true>), g, t, 0, s, b, hdr);
        else hipLaunchKernelGGL((svb_half_encode_kernel<false>), g, t, 0, s, b, hdr);
        return hipGetLastError();
    }
    const uint32_t sc = strict_cap ? 1u : 0u;
    EncPlan* pre = zstd_encode_plans(plans);
    if (pre && svb_encode_fills_plans(integer_size, zigzag, half) && hdr == 0 && !strict_cap) {
        // the int16 zig-zag stream of a read on its way to the entropy stage: its data bytes' sample counted on the way
        if (period_hint) hipLaunchKernelGGL((svb_encode_kernel<2, true, true, true, true>), g, t, 0, s, b, hdr, sc, period_hint, pre);
        else hipLaunchKernelGGL((svb_encode_kernel<2, true, true, false, true>), g, t, 0, s, b, hdr, sc, period_hint, pre);
        return hipGetLastError();
    }
    pre = nullptr;
#define X(E, Z, I)                                                                                                         \
    if (period_hint) hipLaunchKernelGGL((svb_encode_kernel<E, Z, I, true>), g, t, 0, s, b, hdr, sc, period_hint, pre);     \
    else hipLaunchKernelGGL((svb_encode_kernel<E, Z, I, false>), g, t, 0, s, b, hdr, sc, period_hint, pre)
    if (integer_size == 2 && zigzag) { X(2, true, true); }
    else if (integer_size == 2) { X(2, false, false); }
    else if (integer_size == 4 && zigzag) { X(4, true, false); }
    else if (integer_size == 4) { X(4, false, false); }
    else if (integer_size == 1 && zigzag) { X(1, true, false); }
    else if (integer_size == 1) { X(1, false, false); }
    else return hipErrorInvalidValue;
#undef X
    return hipGetLastError();
}

hipError_t launch_svb_decode(const ReadBatch& b, int integer_size, bool zigzag, bool half, hipStream_t s)
{
    if (half) {
        if (integer_size != 1) return hipErrorInvalidValue;
        return zigzag ? launch1(svb_half_decode_kernel<true>, b, s) : launch1(svb_half_decode_kernel<false>, b, s);
    }
    if (integer_size == 2 && zigzag) return launch1(svb_decode_kernel<2, true, true>, b, s);
    if (integer_size == 2) return launch1(svb_decode_kernel<2, false, false>, b, s);
    if (integer_size == 4 && zigzag) return launch1(svb_decode_kernel<4, true, false>, b, s);
    if (integer_size == 4) return launch1(svb_decode_kernel<4, false, false>, b, s);
    if (integer_size == 1 && zigzag) return launch1(svb_decode_kernel<1, true, false>, b, s);
    if (integer_size == 1) return launch1(svb_decode_kernel<1, false, false>, b, s);
    return hipErrorInvalidValue;
}

// ---- segmented launches (few, large reads) --------------------------------------------------------------------------
uint32_t svb_seg_unit_bytes(int integer_size)
{
    return (uint32_t)(WG * (integer_size == 4 ? 4 : 8) * SEG_TILES * integer_size);  // raw bytes of one segment
}

#define VBZ_SVB_DISPATCH(X)                                      \
    do {                                                         \
        if (integer_size == 2 && zigzag) { X(2, true, true); }   \
        else if (integer_size == 2) { X(2, false, false); }      \
        else if (integer_size == 4 && zigzag) { X(4, true, false); } \
        else if (integer_size == 4) { X(4, false, false); }      \
        else if (integer_size == 1 && zigzag) { X(1, true, false); } \
        else if (integer_size == 1) { X(1, false, false); }      \
        else return hipErrorInvalidValue;                        \
    } while (0)

hipError_t launch_svb_encode_seg(const ReadBatch& b, int integer_size, bool zigzag, uint32_t hdr, bool strict_cap, const uint32_t* seg_first,
                                 uint32_t max_segs, uint32_t* seg_bytes, uint64_t* seg_off, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    const uint32_t sc = strict_cap ? 1u : 0u;
    const bool self = max_segs <= seg_self_max();
#define X(E, Z, I)                                                                                                                   \
    hipLaunchKernelGGL((svb_seg_encode_kernel<E, Z, I, true>), dim3(max_segs), dim3(WG), 0, s, b, hdr, sc, seg_first, seg_bytes, seg_off);  \
    if (self) {                                                                                                                      \
        hipLaunchKernelGGL((svb_seg_encode_kernel<E, Z, I, false, true>), dim3(max_segs), dim3(WG), 0, s, b, hdr, sc, seg_first, seg_bytes, seg_off); \
    } else {                                                                                                                         \
        hipLaunchKernelGGL((svb_seg_encode_scan_kernel<E, I>), dim3(b.n_reads), dim3(WG), 0, s, b, hdr, sc, seg_first, seg_bytes, seg_off);    \
        hipLaunchKernelGGL((svb_seg_encode_kernel<E, Z, I, false>), dim3(max_segs), dim3(WG), 0, s, b, hdr, sc, seg_first, seg_bytes, seg_off); \
    }
    VBZ_SVB_DISPATCH(X);
#undef X
    return hipGetLastError();
}

hipError_t launch_svb_decode_seg(const ReadBatch& b, int integer_size, bool zigzag, const uint32_t* seg_first, uint32_t max_segs,
                                 uint32_t* seg_val, uint64_t* seg_pos, uint32_t* seg_run, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    const bool self = max_segs <= seg_self_max();
#define X(E, Z, I)                                                                                                                          \
    hipLaunchKernelGGL((svb_seg_decode_kernel<E, Z, I, 1>), dim3(max_segs), dim3(WG), 0, s, b, seg_first, seg_val, seg_pos, seg_run);             \
    if (self) {                                                                                                                             \
        if (Z) hipLaunchKernelGGL((svb_seg_decode_kernel<E, Z, I, 2, true>), dim3(max_segs), dim3(WG), 0, s, b, seg_first, seg_val, seg_pos, seg_run); \
        hipLaunchKernelGGL((svb_seg_decode_kernel<E, Z, I, 0, true>), dim3(max_segs), dim3(WG), 0, s, b, seg_first, seg_val, seg_pos, seg_run);   \
    } else {                                                                                                                                \
        hipLaunchKernelGGL((svb_seg_decode_scan_kernel<E, I, true>), dim3(b.n_reads), dim3(WG), 0, s, b, seg_first, seg_val, seg_pos, seg_run);   \
        if (Z) {                                                                                                                            \
            hipLaunchKernelGGL((svb_seg_decode_kernel<E, Z, I, 2>), dim3(max_segs), dim3(WG), 0, s, b, seg_first, seg_val, seg_pos, seg_run);     \
            hipLaunchKernelGGL((svb_seg_decode_scan_kernel<E, I, false>), dim3(b.n_reads), dim3(WG), 0, s, b, seg_first, seg_val, seg_pos, seg_run); \
        }                                                                                                                                   \
        hipLaunchKernelGGL((svb_seg_decode_kernel<E, Z, I, 0>), dim3(max_segs), dim3(WG), 0, s, b, seg_first, seg_val, seg_pos, seg_run);         \
    }
    VBZ_SVB_DISPATCH(X);
#undef X
    return hipGetLastError();
}

}  // namespace vbzhip
