// tools/stream_phase_bench.hip -- scaling microbenchmark of the entropy decoder's STREAM PHASE.
//
// The stream phase of zstd_decode.hip (flush_tasks_ring: 64 lanes, one Huffman bit stream each, table look-ups in LDS, compressed
// bytes through a per-lane LDS ring, 32/64-byte store bursts) is 63 % of that kernel and runs there at 2 waves per SIMD because the
// rest of the 3000-line kernel holds 227 registers and 19 KB of LDS.  This tool runs THAT LOOP ALONE, on resident synthetic
// frames of the benchmark's shape (64 streams per frame, the byte statistics of the SURVEY 8d signal's svb data bytes), and sweeps
// what the production kernel cannot: waves per SIMD (LDS padding), ring size, the memory side switched off (no loads / no stores /
// neither), and the table-entry layout.  It answers: what are 3, 4, 5, 6 waves per SIMD worth, and is the phase bound by the
// dependent chain, the LDS or the memory side?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/stream_phase_bench tools/stream_phase_bench.hip
//   tools/stream_phase_bench [--frames 65536] [--distinct 1024] [--reps 5] [--csv out.csv]       the sweep
//   tools/stream_phase_bench --calib                                                              LDS counter calibration kernels
//       (run under `rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES`: every kernel makes a known
//        number of ds_read_u16 gathers with a known conflict degree)
//
// Streams follow RFC 8878 4.2.2 (backward bit streams with an end mark, codes of at most 11 bits), so the loop is the production
// loop, not a model of it.  Not part of the product: nothing here is linked into libvbz_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e__ = (x);                                                              \
        if (e__ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// data bytes of the svb streams of reads 0..7 of the SURVEY 8d generator (seed 5), scaled to 60000 for the commonest byte
static const uint16_t HIST[256] = {
    40900, 60000, 43442, 40396, 40515, 39203, 40196, 39951, 39663, 39150, 39288, 39179, 37746, 37739, 37695, 37003, 37240, 35996, 36429, 35456,
    35219, 34434, 34393, 32800, 33018, 31876, 31604, 30658, 31036, 29532, 29525, 28109, 28496, 26770, 27007, 25511, 25502, 24091, 24616, 23086,
    23129, 21493, 21856, 20200, 20331, 19535, 19346, 17753, 17603, 16424, 16821, 15357, 15628, 14471, 14592, 13628, 13558, 12493, 12239, 11626,
    11280, 10309, 10351, 9707,  9518,  8666,  8375,  7668,  7923,  7090,  6978,  6388,  6305,  5741,  5790,  5114,  5001,  4582,  4575,  4081,
    4153,  3735,  3732,  3238,  3219,  2827,  2878,  2524,  2539,  2333,  2246,  2110,  2127,  1929,  1863,  1604,  1670,  1428,  1345,  1287,
    1278,  1062,  1089,  961,   1026,  871,   801,   694,   726,   651,   622,   534,   530,   530,   496,   467,   438,   442,   389,   404,
    387,   406,   438,   360,   329,   317,   346,   329,   302,   283,   259,   283,   261,   288,   285,   249,   234,   263,   251,   271,
    237,   256,   261,   203,   227,   183,   227,   222,   249,   229,   234,   256,   232,   227,   191,   222,   210,   196,   259,   242,
    220,   208,   220,   198,   193,   217,   244,   213,   205,   181,   225,   181,   183,   176,   210,   191,   198,   145,   174,   196,
    205,   162,   179,   196,   157,   157,   188,   196,   210,   198,   169,   186,   164,   174,   171,   193,   196,   176,   183,   191,
    205,   225,   176,   150,   169,   210,   157,   157,   196,   208,   205,   191,   200,   181,   164,   137,   210,   162,   196,   191,
    193,   179,   208,   147,   176,   154,   147,   135,   196,   142,   164,   188,   210,   188,   150,   150,   147,   162,   147,   162,
    174,   130,   167,   167,   213,   147,   164,   169,   133,   193,   171,   167,   152,   186,   174,   152 };

constexpr int WAVE = 64;
constexpr int TLOG = 11;
constexpr int STREAMS = 64;

// ---------------------------------------------------------------------------------------------------------------- host side
struct Code
{
    uint8_t len[256];
    uint16_t code[256];
    uint16_t table[1 << TLOG];  // symbol | nbBits << 8 (production layout)
};

static void build_code(Code& c)
{
    // Huffman lengths by the two-queue merge, then limited to TLOG bits by the usual repair (lengthen the cheapest).
    struct Node { uint64_t w; int l, r; };
    std::vector<Node> nodes;
    std::vector<int> order(256);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [](int a, int b) { return HIST[a] != HIST[b] ? HIST[a] < HIST[b] : a < b; });
    for (int s : order) nodes.push_back({ HIST[s], -1, s });
    size_t qa = 0, qb = 256;
    auto pop = [&]() -> int {
        if (qb >= nodes.size() || (qa < 256 && nodes[qa].w <= nodes[qb].w)) return (int)qa++;
        return (int)qb++;
    };
    for (int k = 0; k < 255; ++k) {
        const int a = pop(), b = pop();
        nodes.push_back({ nodes[a].w + nodes[b].w, a, b });
    }
    std::vector<int> depth(nodes.size(), 0);
    for (int i = (int)nodes.size() - 1; i >= 256; --i) {
        depth[nodes[i].l] = depth[i] + 1;
        depth[nodes[i].r] = depth[i] + 1;
    }
    int len[256];
    for (int i = 0; i < 256; ++i) len[nodes[i].r] = depth[i];
    // limit: clamp, then pay back the Kraft debt
    int64_t kraft = 0;
    for (int s = 0; s < 256; ++s) {
        if (len[s] > TLOG) len[s] = TLOG;
        kraft += 1ll << (TLOG - len[s]);
    }
    while (kraft > (1ll << TLOG)) {  // lengthen the least frequent symbol that is not yet at the limit
        int best = -1;
        for (int s : order)
            if (len[s] < TLOG) { best = s; break; }
        kraft -= 1ll << (TLOG - len[best] - 1);
        ++len[best];
    }
    while (kraft < (1ll << TLOG)) {  // give spare cells to the most frequent symbol they fit
        for (int i = 255; i >= 0; --i) {
            const int s = order[i];
            if (len[s] > 1 && kraft + (1ll << (TLOG - len[s])) <= (1ll << TLOG)) {
                kraft += 1ll << (TLOG - len[s]);
                --len[s];
                break;
            }
        }
    }
    // RFC 8878 4.2.1: cells by increasing weight, then symbol value
    uint32_t start = 0;
    for (int w = 1; w <= TLOG; ++w)
        for (int s = 0; s < 256; ++s)
            if (TLOG + 1 - len[s] == w) {
                const uint32_t cells = 1u << (w - 1);
                c.len[s] = (uint8_t)len[s];
                c.code[s] = (uint16_t)(start >> (w - 1));
                for (uint32_t i = 0; i < cells; ++i) c.table[start + i] = (uint16_t)(s | (len[s] << 8));
                start += cells;
            }
    if (start != (1u << TLOG)) {
        fprintf(stderr, "code construction failed\n");
        exit(1);
    }
}

struct Task { uint32_t src, size, out, cnt; };

static uint64_t rng_next(uint64_t& s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one frame: STREAMS streams of `cnt` symbols each; returns compressed size; syms gets the content
static uint32_t make_frame(const Code& c, const std::vector<uint32_t>& cdf, uint32_t cnt, uint64_t seed, uint8_t* comp, uint32_t comp_cap,
                           uint8_t* syms, Task* tasks)
{
    uint32_t pos = 0;
    uint64_t rs = seed;
    for (int st = 0; st < STREAMS; ++st) {
        uint8_t* y = syms + (size_t)st * cnt;
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t u = (uint32_t)(rng_next(rs) % cdf.back());
            y[i] = (uint8_t)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
        }
        // backward stream: the last symbol is written first
        uint64_t acc = 0;
        int nb = 0;
        const uint32_t begin = pos;
        for (int64_t i = (int64_t)cnt - 1; i >= 0; --i) {
            acc |= (uint64_t)c.code[y[i]] << nb;
            nb += c.len[y[i]];
            while (nb >= 8) {
                if (pos >= comp_cap) { fprintf(stderr, "frame slot too small\n"); exit(1); }
                comp[pos++] = (uint8_t)acc;
                acc >>= 8;
                nb -= 8;
            }
        }
        acc |= 1ull << nb;  // end mark
        comp[pos++] = (uint8_t)acc;
        tasks[st] = { begin, pos - begin, (uint32_t)st * cnt, cnt };
    }
    return pos;
}

// ---------------------------------------------------------------------------------------------------------------- device side
typedef __attribute__((address_space(1))) const uint8_t gcu8;
typedef __attribute__((address_space(1))) uint8_t gu8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];  // padding that sets the occupancy

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// RING dwords per lane in the LDS ring (a top-up point every RING symbols), BURST bytes stored together by a lane
// LOADS: the input side on or off.  STORES: 0 nothing is stored; 1 every lane stores its own BURST bytes (BURST / 16 adjacent 16-byte stores: a
// wave-level store touches 64 different lines); 2 the lanes' bytes go through LDS and leave as whole BURST-byte segments, BURST / 16
// adjacent lanes writing one stream's segment (a wave-level store touches 1024 / BURST segments).  FMT 0: table entry symbol | nb << 8 (production); 1: nb | symbol << 8 (the shift
// amount is the entry itself, one operation less on the dependent chain)
template <int RING, int BURST, bool LOADS, int STORES, int FMT>
__global__ __launch_bounds__(WAVE) void stream_kernel(const uint8_t* __restrict__ src, uint64_t src_stride, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                                      const Task* __restrict__ tasks, const uint16_t* __restrict__ table, uint32_t* __restrict__ verdict,
                                                      uint32_t pad_words)
{
    constexpr int BATCH = RING / 2;
    constexpr int PERIOD = RING;  // symbols between two top-up points: a period eats at most 11 * PERIOD / 32 <= BATCH dwords
    __shared__ __attribute__((aligned(16))) uint16_t T[1 << TLOG];
    __shared__ uint32_t ringbuf[RING + 1][WAVE];
    constexpr int G = BURST / 16;  // 16-byte granules per segment
    __shared__ u32x4 stage[STORES == 2 ? G : 1][WAVE];
    __shared__ uint32_t outbase[WAVE];
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (pad_words && lane == 0) reinterpret_cast<volatile uint32_t*>(dyn_lds)[pad_words - 1] = 0;  // keeps the padding allocated
    // the frame's decoding table: 4 KB, coalesced
    {
        const u32x4* g = reinterpret_cast<const u32x4*>(table);
        u32x4* l = reinterpret_cast<u32x4*>(T);
#pragma unroll
        for (int k = 0; k < (1 << TLOG) * 2 / 16 / WAVE; ++k) {
            u32x4 v = g[k * WAVE + lane];
            if (FMT == 1) {  // nb | symbol << 8
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t x = v[j];
                    v[j] = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
                }
            }
            l[k * WAVE + lane] = v;
        }
    }
    const Task t = tasks[(size_t)f * STREAMS + lane];
    gcu8* p = (gcu8*)(src + (size_t)f * src_stride) + t.src;
    gu8* o = (gu8*)(dst + (size_t)f * dst_stride) + t.out;
    const uint32_t nbytes = t.size;
    uint32_t cnt = t.cnt;
    uint32_t* ring = &ringbuf[0][0] + lane;
    outbase[lane] = t.out;
    gu8* const dstf = (gu8*)(dst + (size_t)f * dst_stride);
    uint32_t burst_iter = 0;
    wave_lds_sync();

    bool bad = false;
    int32_t n = -1;
    uint32_t nextbyte = 0, widx = 0;
    uint32_t acc = 0;  // STORES == false: what would have been stored
    {
        const uint32_t last = LOADS ? (nbytes ? p[nbytes - 1] : 0u) : 0x80u;
        if (last == 0) {
            bad = true;
            cnt = 0;
        } else {
            n = -(int32_t)(8 - (31 - __clz((int)last)));
            nextbyte = nbytes;
        }
    }
    uint32_t pend[BATCH];
    auto fetch_batch = [&]() {
        if (!LOADS) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) pend[k] = (widx + (uint32_t)k + 1u) * 0x9E3779B1u ^ ((uint32_t)lane * 0x85EBCA6Bu);
            nextbyte = nextbyte >= 4u * BATCH ? nextbyte - 4u * BATCH : 0u;
            return;
        }
        typedef __attribute__((address_space(1), aligned(1))) const u32x4 gq4;
        typedef __attribute__((address_space(1), aligned(1))) const uint32_t gq1;
#ifdef VBZ_SIMPLE_FETCH
        // No slow path: the batch below `nextbyte` is read whole even where it reaches below the start of the stream.  Those bytes
        // are never consumed by a stream that ends where it must (the final check on n), so their value does not matter; they are
        // readable because a stream never starts a frame (headers, tree, jump table precede it) and the arena has slack.
        {
            gcu8* q = p + (int32_t)nextbyte - 4 * BATCH;
#pragma unroll
            for (int v = 0; v < BATCH / 4; ++v) {
                const u32x4 x = *(gq4*)(q + 16 * (BATCH / 4 - 1 - v));
                pend[4 * v + 0] = x.w;
                pend[4 * v + 1] = x.z;
                pend[4 * v + 2] = x.y;
                pend[4 * v + 3] = x.x;
            }
            nextbyte = nextbyte >= 4u * BATCH ? nextbyte - 4u * BATCH : 0u;
        }
#else
        if (nextbyte >= 4u * BATCH) {
            gcu8* q = p + nextbyte - 4 * BATCH;
#pragma unroll
            for (int v = 0; v < BATCH / 4; ++v) {
                const u32x4 x = *(gq4*)(q + 16 * (BATCH / 4 - 1 - v));
                pend[4 * v + 0] = x.w;
                pend[4 * v + 1] = x.z;
                pend[4 * v + 2] = x.y;
                pend[4 * v + 3] = x.x;
            }
            nextbyte -= 4 * BATCH;
        } else {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                uint32_t v = 0;
                if (nextbyte >= 4) {
                    v = *(gq1*)(p + nextbyte - 4);
                    nextbyte -= 4;
                } else if (nextbyte > 0) {
                    for (uint32_t j = 0; j < nextbyte; ++j) v |= (uint32_t)p[j] << (8 * (j + 4 - nextbyte));
                    nextbyte = 0;
                }
                pend[k] = v;
            }
        }
#endif
    };
#define RING_PUT()                                                                                     \
    do {                                                                                               \
        const uint32_t wb__ = (uint32_t)RING - (widx & (uint32_t)(RING - 1));                          \
        _Pragma("unroll") for (int k = 0; k < BATCH; ++k) ring[(wb__ - (uint32_t)k) * WAVE] = pend[k]; \
        if (wb__ == (uint32_t)RING) ring[0] = pend[0];                                                 \
        widx += BATCH;                                                                                 \
    } while (0)
    for (int q = 0; q < 2; ++q) {
        fetch_batch();
        RING_PUT();
    }
    fetch_batch();

    int32_t tprev = n >> 5;
    uint32_t w0 = ring[((((uint32_t)tprev) & (uint32_t)(RING - 1)) + 1u) * WAVE];
    uint32_t w1 = ring[(((uint32_t)tprev) & (uint32_t)(RING - 1)) * WAVE];
    uint32_t w2 = ring[(((uint32_t)tprev - 1u) & (uint32_t)(RING - 1)) * WAVE];
    constexpr uint32_t sL = 32 - TLOG;
#define NB(e) (FMT == 0 ? ((e) >> 8) : ((e) & 0xFFu))
#define SYM(e) (FMT == 0 ? ((e) & 0xFFu) : ((e) >> 8))
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
        e1 = T[x__ >> sL];                                                   \
        x__ <<= (FMT == 0 ? (e1 >> 8) : (e1 & 31u));                         \
        e2 = T[x__ >> sL];                                                   \
    } while (0)
#define QUAD(dstword)                                                                             \
    do {                                                                                          \
        uint32_t e1, e2, e3, e4;                                                                  \
        HUF_PAIR(e1, e2);                                                                         \
        n -= (int32_t)(NB(e1) + NB(e2));                                                          \
        HUF_PAIR(e3, e4);                                                                         \
        n -= (int32_t)(NB(e3) + NB(e4));                                                          \
        dstword = SYM(e1) | (SYM(e2) << 8) | (SYM(e3) << 16) | (SYM(e4) << 24);                   \
    } while (0)
#define TOP_UP()                                                             \
    do {                                                                     \
        if (widx - (((uint32_t)~n) >> 5) <= (uint32_t)(RING - BATCH)) {      \
            RING_PUT();                                                      \
            fetch_batch();                                                   \
        }                                                                    \
    } while (0)
    typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
    static_assert(BURST % 16 == 0 && BURST >= 16 && (BURST % PERIOD == 0 || PERIOD % BURST == 0), "burst bytes");
    while (__any(cnt > 0)) {
        if constexpr (STORES == 2) {
            const uint64_t act = __ballot(cnt >= (uint32_t)BURST);
            if (act) {
                if (cnt >= (uint32_t)BURST) {
                    uint32_t ow[4];
#pragma unroll
                    for (int q = 0; q < BURST / 4; ++q) {
                        if ((4 * q) % PERIOD == 0) TOP_UP();
                        QUAD(ow[q & 3]);
                        if ((q & 3) == 3) {  // granule q / 4 of this lane's segment; rotated so that reads and writes are conflict-free
                            const u32x4 ov = { ow[0], ow[1], ow[2], ow[3] };
                            stage[q >> 2][(lane + (16 / G) * (q >> 2)) & 63] = ov;
                        }
                    }
                    o += BURST;
                    cnt -= BURST;
                }
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const uint32_t st = (uint32_t)i * (WAVE / G) + (uint32_t)lane / G, pp = (uint32_t)lane % G;
                    const u32x4 v = stage[pp][(st + (16 / G) * pp) & 63];
                    const uint32_t ob = outbase[st];
                    if ((act >> st) & 1ull) *(gs4*)(dstf + ob + burst_iter * BURST + 16u * pp) = v;
                }
                ++burst_iter;
                wave_lds_sync();
                continue;
            }
        } else if (cnt >= (uint32_t)BURST) {
            // BURST symbols leave with BURST / 16 adjacent 16-byte stores; a top-up point every PERIOD symbols
            uint32_t ow[BURST / 4];
#pragma unroll
            for (int q = 0; q < BURST / 4; ++q) {
                if ((4 * q) % PERIOD == 0) TOP_UP();
                QUAD(ow[q]);
            }
#pragma unroll
            for (int q = 0; q < BURST / 16; ++q) {
                const u32x4 ov = { ow[4 * q], ow[4 * q + 1], ow[4 * q + 2], ow[4 * q + 3] };
                if (STORES) *(gs4*)(o + 16 * q) = ov;
                else acc ^= ov.x ^ ov.y ^ ov.z ^ ov.w;
            }
            o += BURST;
            cnt -= BURST;
            continue;
        }
        TOP_UP();
#pragma unroll
        for (int i = 0; i < (PERIOD < 16 ? PERIOD : 16); ++i) {
            if (cnt > 0) {
                uint32_t e1, e2;
                HUF_PAIR(e1, e2);
                (void)e2;
                n -= (int32_t)NB(e1);
                if (STORES) *o = (uint8_t)SYM(e1);
                else acc ^= SYM(e1);
                ++o;
                --cnt;
            }
        }
    }
    if (LOADS && !bad && n != -(int32_t)(8u * nbytes)) bad = true;
    if (STORES == 0 && acc == 0x7E57AB1Eu) *o = 1;  // never true in practice: keeps the decode alive
    if (__any(bad) && lane == 0) atomicAdd(verdict, 1u);
#undef HUF_PAIR
#undef QUAD
#undef TOP_UP
#undef RING_PUT
#undef NB
#undef SYM
}

// ---- the same phase with the top-up restructured (what the production kernel gets in round 4) -----------------------------------
// stream_kernel waits `vmcnt(0)` in front of every ring write: the batch it commits was requested an iteration earlier (a
// loop-carried register set), the compiler must pick a count that is right on every path into the loop, and on gfx9 loads and
// stores retire through ONE in-order counter -- so each top-up also waits for the stores just issued, a full store latency per
// 32 symbols.  Here a batch is requested and committed inside the same stretch of straight-line code (request, decode a period,
// store, commit), so the wait in front of the commit is a counted vmcnt(stores issued since); the request needs no slow path (the
// batch is read whole even where it reaches below the start of the stream -- bytes no conforming stream consumes); a period is
// RING / 2 symbols so that the ring can never run dry between a request and its commit.
template <int RING, int BURST, bool LOADS, int STORES, int FMT, int PF = 0>
__global__ __launch_bounds__(WAVE) void stream_kernel2(const uint8_t* __restrict__ src, uint64_t src_stride, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                                       const Task* __restrict__ tasks, const uint16_t* __restrict__ table, uint32_t* __restrict__ verdict,
                                                       uint32_t pad_words)
{
    constexpr int BATCH = RING / 2;
    constexpr int PERIOD = RING / 2;  // symbols between two top-up points: at most 11 * PERIOD / 32 dwords, see the invariants above
    static_assert(11 * PERIOD <= 16 * (BATCH + 1), "ring could run dry");
    __shared__ __attribute__((aligned(16))) uint16_t T[1 << TLOG];
    __shared__ uint32_t ringbuf[RING + 1][WAVE];
    constexpr int G = BURST / 16;
    __shared__ u32x4 stage[(STORES == 2 || STORES == 4) ? G : 1][WAVE];
    __shared__ uint32_t outbase[WAVE];
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (pad_words && lane == 0) reinterpret_cast<volatile uint32_t*>(dyn_lds)[pad_words - 1] = 0;
    {
        const u32x4* g = reinterpret_cast<const u32x4*>(table);
        u32x4* l = reinterpret_cast<u32x4*>(T);
#pragma unroll
        for (int k = 0; k < (1 << TLOG) * 2 / 16 / WAVE; ++k) {
            u32x4 v = g[k * WAVE + lane];
            if (FMT == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t x = v[j];
                    v[j] = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
                }
            }
            l[k * WAVE + lane] = v;
        }
    }
    if (PF) {
        // touch every 128-byte line of the frame's compressed bytes once, in address order: one sequential read per frame brings them
        // into the memory-side cache; the lanes' scattered ring requests find them there
        const uint32_t last_task_end = __shfl((int)(tasks[(size_t)f * STREAMS + 63].src + tasks[(size_t)f * STREAMS + 63].size), 0, 64);
        const uint8_t* fb = src + (size_t)f * src_stride;
        uint32_t sink = 0;
        for (uint32_t off = (uint32_t)lane * 128u; off < last_task_end; off += 64u * 128u) {
            uint32_t v;
            asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(fb + off) : "memory");
            sink ^= v;  // (never waited for on purpose: the value is garbage, only the request matters)
        }
        if (PF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sink == 0x12345u && pad_words == 0xFFFFFFFFu) verdict[1] = sink;
    }
    const Task t = tasks[(size_t)f * STREAMS + lane];
    gcu8* p = (gcu8*)(src + (size_t)f * src_stride) + t.src;
    gu8* o = (gu8*)(dst + (size_t)f * dst_stride) + t.out;
    gu8* const dstf = (gu8*)(dst + (size_t)f * dst_stride);
    const uint32_t nbytes = t.size;
    uint32_t cnt = t.cnt;
    uint32_t* ring = &ringbuf[0][0] + lane;
    outbase[lane] = t.out;
    uint32_t burst_iter = 0;
    wave_lds_sync();

    bool bad = false;
    int32_t n = -1;
    uint32_t nextbyte = 0, widx = 0;
    uint32_t acc = 0;
    {
        const uint32_t last = LOADS ? (nbytes ? p[nbytes - 1] : 0u) : 0x80u;
        if (last == 0) {
            bad = true;
            cnt = 0;
        } else {
            n = -(int32_t)(8 - (31 - __clz((int)last)));
            nextbyte = nbytes;
        }
    }
    typedef __attribute__((address_space(1), aligned(1))) const u32x4 gq4;
#define FETCH(pend)                                                                                                     \
    do {                                                                                                                \
        if (LOADS) {                                                                                                    \
            gcu8* q__ = p + (int32_t)nextbyte - 4 * BATCH;                                                              \
            _Pragma("unroll") for (int v = 0; v < BATCH / 4; ++v) {                                                     \
                const u32x4 x__ = *(gq4*)(q__ + 16 * (BATCH / 4 - 1 - v));                                              \
                pend[4 * v + 0] = x__.w;                                                                                \
                pend[4 * v + 1] = x__.z;                                                                                \
                pend[4 * v + 2] = x__.y;                                                                                \
                pend[4 * v + 3] = x__.x;                                                                                \
            }                                                                                                           \
        } else {                                                                                                        \
            _Pragma("unroll") for (int k = 0; k < BATCH; ++k) pend[k] = (widx + (uint32_t)k + 1u) * 0x9E3779B1u ^ ((uint32_t)lane * 0x85EBCA6Bu); \
        }                                                                                                               \
        nextbyte = nextbyte >= 4u * BATCH ? nextbyte - 4u * BATCH : 0u;                                                 \
    } while (0)
#define RING_PUT(pend)                                                                                 \
    do {                                                                                               \
        const uint32_t wb__ = (uint32_t)RING - (widx & (uint32_t)(RING - 1));                          \
        _Pragma("unroll") for (int k = 0; k < BATCH; ++k) ring[(wb__ - (uint32_t)k) * WAVE] = pend[k]; \
        if (wb__ == (uint32_t)RING) ring[0] = pend[0];                                                 \
        widx += BATCH;                                                                                 \
    } while (0)
    for (int q = 0; q < 2; ++q) {
        uint32_t pend0[BATCH];
        FETCH(pend0);
        RING_PUT(pend0);
    }
    int32_t tprev = n >> 5;
    uint32_t w0 = ring[((((uint32_t)tprev) & (uint32_t)(RING - 1)) + 1u) * WAVE];
    uint32_t w1 = ring[(((uint32_t)tprev) & (uint32_t)(RING - 1)) * WAVE];
    uint32_t w2 = ring[(((uint32_t)tprev - 1u) & (uint32_t)(RING - 1)) * WAVE];
    constexpr uint32_t sL = 32 - TLOG;
#define NB(e) (FMT == 0 ? ((e) >> 8) : ((e) & 0xFFu))
#define SYM(e) (FMT == 0 ? ((e) & 0xFFu) : ((e) >> 8))
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
        e1 = T[x__ >> sL];                                                   \
        x__ <<= (FMT == 0 ? (e1 >> 8) : (e1 & 31u));                         \
        e2 = T[x__ >> sL];                                                   \
    } while (0)
#define QUAD(dstword)                                                                             \
    do {                                                                                          \
        uint32_t e1, e2, e3, e4;                                                                  \
        HUF_PAIR(e1, e2);                                                                         \
        n -= (int32_t)(NB(e1) + NB(e2));                                                          \
        HUF_PAIR(e3, e4);                                                                         \
        n -= (int32_t)(NB(e3) + NB(e4));                                                          \
        dstword = SYM(e1) | (SYM(e2) << 8) | (SYM(e3) << 16) | (SYM(e4) << 24);                   \
    } while (0)
#define ROOM() (widx - (((uint32_t)~n) >> 5) <= (uint32_t)(RING - BATCH))
    typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
    static_assert(BURST % PERIOD == 0 && BURST % 16 == 0 && PERIOD % 4 == 0, "burst / period");
    constexpr int NPER = BURST / PERIOD;
    while (__any(cnt > 0)) {
        const bool burst = cnt >= (uint32_t)BURST;
        const uint64_t act = __ballot(burst);
        if (burst) {
            constexpr bool STAGED = STORES == 2 || STORES == 4;
            uint32_t ow[STAGED ? 4 : BURST / 4];
#pragma unroll
            for (int h = 0; h < NPER; ++h) {
                uint32_t pend[BATCH];
                const bool issue = ROOM();
                if (issue) FETCH(pend);
#pragma unroll
                for (int q = h * PERIOD / 4; q < (h + 1) * PERIOD / 4; ++q) {
                    if (STAGED) {
                        QUAD(ow[q & 3]);
                        if ((q & 3) == 3) {
                            const u32x4 ov = { ow[0], ow[1], ow[2], ow[3] };
                            stage[q >> 2][(lane + (16 / G) * (q >> 2)) & 63] = ov;
                        }
                    } else {
                        QUAD(ow[q]);
                    }
                }
                if (h == NPER - 1 && !STAGED) {  // the burst's stores go out in front of the last commit: its wait is vmcnt(stores)
#pragma unroll
                    for (int q = 0; q < BURST / 16; ++q) {
                        const u32x4 ov = { ow[4 * q], ow[4 * q + 1], ow[4 * q + 2], ow[4 * q + 3] };
                        if (STORES == 1) *(gs4*)(o + 16 * q) = ov;
                        else if (STORES == 3) __builtin_nontemporal_store(ov, (gs4*)(o + 16 * q));
                        else acc ^= ov.x ^ ov.y ^ ov.z ^ ov.w;
                    }
                }
                if (issue) RING_PUT(pend);
            }
            o += BURST;
            cnt -= BURST;
        }
        if ((STORES == 2 || STORES == 4) && act) {
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t st = (uint32_t)i * (WAVE / G) + (uint32_t)lane / G, pp = (uint32_t)lane % G;
                const u32x4 v = stage[pp][(st + (16 / G) * pp) & 63];
                const uint32_t ob = outbase[st];
                if ((act >> st) & 1ull) {
                    if (STORES == 4) __builtin_nontemporal_store(v, (gs4*)(dstf + ob + burst_iter * BURST + 16u * pp));
                    else *(gs4*)(dstf + ob + burst_iter * BURST + 16u * pp) = v;
                }
            }
            ++burst_iter;
            wave_lds_sync();
        }
        if (act) continue;
        // tails: fewer than BURST symbols left in every lane
        {
            uint32_t pend[BATCH];
            const bool issue = ROOM() && cnt > 0;
            if (issue) FETCH(pend);
#pragma unroll
            for (int i = 0; i < PERIOD; ++i) {
                if (cnt > 0) {
                    uint32_t e1, e2;
                    HUF_PAIR(e1, e2);
                    (void)e2;
                    n -= (int32_t)NB(e1);
                    if (STORES) *o = (uint8_t)SYM(e1);
                    else acc ^= SYM(e1);
                    ++o;
                    --cnt;
                }
            }
            if (issue) RING_PUT(pend);
        }
    }
    if (LOADS && !bad && n != -(int32_t)(8u * nbytes)) bad = true;
    if (STORES == 0 && acc == 0x7E57AB1Eu) *o = 1;
    if (__any(bad) && lane == 0) atomicAdd(verdict, 1u);
#undef HUF_PAIR
#undef QUAD
#undef ROOM
#undef RING_PUT
#undef FETCH
#undef NB
#undef SYM
}

struct Bench
{
    uint8_t* d_src;
    uint8_t* d_dst;
    Task* d_tasks;
    uint16_t* d_table;
    uint32_t* d_verdict;
    uint64_t src_stride, dst_stride;
    uint32_t frames, cnt, reps;
    double clock_ghz;
    FILE* csv;
};

// ---- decode waves + one WRITER wave per workgroup -----------------------------------------------------------------------------------
// On gfx9 a wave's loads and stores retire through one in-order counter: a ring request issued behind a burst of stores comes back
// only when those stores have been acknowledged.  Here the decode waves never store (tails apart): a lane's bytes go to an LDS stage,
// SEG bytes per lane, and the workgroup's last wave -- which never loads -- writes them out as whole SEG-byte segments (SEG / 16
// adjacent lanes per segment).  Hand-over through an LDS flag per decode wave (LDS operations of a wave execute in order, so the flag
// write follows the stage writes and the writer's stage reads follow its flag read).
template <int RING, int SEG, int NDEC, bool NT>
__global__ __launch_bounds__(WAVE * (NDEC + 1)) void stream_kernel3(const uint8_t* __restrict__ src, uint64_t src_stride, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                                                    const Task* __restrict__ tasks, const uint16_t* __restrict__ table, uint32_t* __restrict__ verdict,
                                                                    uint32_t pad_words, uint32_t frames)
{
    constexpr int BATCH = RING / 2, PERIOD = RING / 2, G = SEG / 16, NPER = SEG / PERIOD;
    static_assert(SEG % PERIOD == 0 && PERIOD % 4 == 0, "segment / period");
    __shared__ __attribute__((aligned(16))) uint16_t Tall[NDEC][1 << TLOG];
    __shared__ uint32_t ringall[NDEC][RING + 1][WAVE];
    __shared__ u32x4 stageall[NDEC][G][WAVE];
    __shared__ uint32_t outbaseall[NDEC][WAVE];
    __shared__ uint32_t ctl[NDEC][8];  // 0 flag (1: stage full), 1 segment index, 2/3 lanes that take part, 4 done
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (pad_words && threadIdx.x == 0) reinterpret_cast<volatile uint32_t*>(dyn_lds)[pad_words - 1] = 0;
    if (threadIdx.x < NDEC * 8) (&ctl[0][0])[threadIdx.x] = 0;
    __syncthreads();
    typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
    if (wv == NDEC) {
        // ---- the writer
        uint32_t done_mask = 0;
        while (done_mask != (1u << NDEC) - 1u) {
            bool idle = true;
#pragma unroll 1
            for (int w = 0; w < NDEC; ++w) {
                if (done_mask & (1u << w)) continue;
                volatile uint32_t* c = ctl[w];
                const uint32_t fin = c[4];   // read BEFORE the flag: a wave that is done hands nothing over any more
                const uint32_t flag = c[0];
                if (flag == 1) {
                    idle = false;
                    const uint32_t iter = c[1];
                    const uint64_t act = (uint64_t)c[2] | ((uint64_t)c[3] << 32);
                    const uint32_t f = blockIdx.x * NDEC + w;
                    gu8* const dstf = (gu8*)(dst + (size_t)f * dst_stride);
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const uint32_t st = (uint32_t)i * (WAVE / G) + (uint32_t)lane / G, pp = (uint32_t)lane % G;
                        const u32x4 v = stageall[w][pp][(st + (16 / G) * pp) & 63];
                        const uint32_t ob = outbaseall[w][st];
                        if ((act >> st) & 1ull) {
                            if (NT) __builtin_nontemporal_store(v, (gs4*)(dstf + ob + iter * SEG + 16u * pp));
                            else *(gs4*)(dstf + ob + iter * SEG + 16u * pp) = v;
                        }
                    }
                    wave_lds_sync();
                    c[0] = 0;
                } else if (flag == 0 && fin == 1) {
                    done_mask |= 1u << w;
                }
            }
            if (idle) __builtin_amdgcn_s_sleep(8);
        }
        return;
    }
    const uint32_t f = blockIdx.x * NDEC + wv;
    volatile uint32_t* myctl = ctl[wv];
    if (f >= frames) {
        if (lane == 0) myctl[4] = 1;
        return;
    }
    uint16_t* T = Tall[wv];
    {
        const u32x4* g = reinterpret_cast<const u32x4*>(table);
        u32x4* l = reinterpret_cast<u32x4*>(T);
#pragma unroll
        for (int k = 0; k < (1 << TLOG) * 2 / 16 / WAVE; ++k) l[k * WAVE + lane] = g[k * WAVE + lane];
    }
    const Task t = tasks[(size_t)f * STREAMS + lane];
    gcu8* p = (gcu8*)(src + (size_t)f * src_stride) + t.src;
    gu8* o = (gu8*)(dst + (size_t)f * dst_stride) + t.out;
    const uint32_t nbytes = t.size;
    uint32_t cnt = t.cnt;
    uint32_t* ring = &ringall[wv][0][0] + lane;
    outbaseall[wv][lane] = t.out;
    uint32_t seg_iter = 0;
    wave_lds_sync();
    bool bad = false;
    int32_t n = -1;
    uint32_t nextbyte = 0, widx = 0;
    {
        const uint32_t last = nbytes ? p[nbytes - 1] : 0u;
        if (last == 0) {
            bad = true;
            cnt = 0;
        } else {
            n = -(int32_t)(8 - (31 - __clz((int)last)));
            nextbyte = nbytes;
        }
    }
    typedef __attribute__((address_space(1), aligned(1))) const u32x4 gq4;
#define FETCH(pend)                                                                                                     \
    do {                                                                                                                \
        gcu8* q__ = p + (int32_t)nextbyte - 4 * BATCH;                                                                  \
        _Pragma("unroll") for (int v = 0; v < BATCH / 4; ++v) {                                                         \
            const u32x4 x__ = *(gq4*)(q__ + 16 * (BATCH / 4 - 1 - v));                                                  \
            pend[4 * v + 0] = x__.w;                                                                                    \
            pend[4 * v + 1] = x__.z;                                                                                    \
            pend[4 * v + 2] = x__.y;                                                                                    \
            pend[4 * v + 3] = x__.x;                                                                                    \
        }                                                                                                               \
        nextbyte = nextbyte >= 4u * BATCH ? nextbyte - 4u * BATCH : 0u;                                                 \
    } while (0)
#define RING_PUT(pend)                                                                                 \
    do {                                                                                               \
        const uint32_t wb__ = (uint32_t)RING - (widx & (uint32_t)(RING - 1));                          \
        _Pragma("unroll") for (int k = 0; k < BATCH; ++k) ring[(wb__ - (uint32_t)k) * WAVE] = pend[k]; \
        if (wb__ == (uint32_t)RING) ring[0] = pend[0];                                                 \
        widx += BATCH;                                                                                 \
    } while (0)
    for (int q = 0; q < 2; ++q) {
        uint32_t pend0[BATCH];
        FETCH(pend0);
        RING_PUT(pend0);
    }
    int32_t tprev = n >> 5;
    uint32_t w0 = ring[((((uint32_t)tprev) & (uint32_t)(RING - 1)) + 1u) * WAVE];
    uint32_t w1 = ring[(((uint32_t)tprev) & (uint32_t)(RING - 1)) * WAVE];
    uint32_t w2 = ring[(((uint32_t)tprev - 1u) & (uint32_t)(RING - 1)) * WAVE];
    constexpr uint32_t sL = 32 - TLOG;
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
        e1 = T[x__ >> sL];                                                   \
        x__ <<= (e1 >> 8);                                                   \
        e2 = T[x__ >> sL];                                                   \
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
    while (__any(cnt > 0)) {
        const bool burst = cnt >= (uint32_t)SEG;
        const uint64_t act = __ballot(burst);
        if (act) {
            while (myctl[0] != 0) __builtin_amdgcn_s_sleep(1);  // the writer still reads the stage
            wave_lds_sync();
            if (burst) {
                uint32_t ow[4];
#pragma unroll
                for (int h = 0; h < NPER; ++h) {
                    uint32_t pend[BATCH];
                    const bool issue = ROOM();
                    if (issue) FETCH(pend);
#pragma unroll
                    for (int q = h * PERIOD / 4; q < (h + 1) * PERIOD / 4; ++q) {
                        QUAD(ow[q & 3]);
                        if ((q & 3) == 3) {
                            const u32x4 ov = { ow[0], ow[1], ow[2], ow[3] };
                            stageall[wv][q >> 2][(lane + (16 / G) * (q >> 2)) & 63] = ov;
                        }
                    }
                    if (issue) RING_PUT(pend);
                }
                o += SEG;
                cnt -= SEG;
            }
            wave_lds_sync();
            if (lane == 0) {
                myctl[1] = seg_iter;
                myctl[2] = (uint32_t)act;
                myctl[3] = (uint32_t)(act >> 32);
                myctl[0] = 1;
            }
            ++seg_iter;
            continue;
        }
        {
            uint32_t pend[BATCH];
            const bool issue = ROOM() && cnt > 0;
            if (issue) FETCH(pend);
#pragma unroll
            for (int i = 0; i < PERIOD; ++i) {
                if (cnt > 0) {
                    uint32_t e1, e2;
                    HUF_PAIR(e1, e2);
                    (void)e2;
                    n -= (int32_t)(e1 >> 8);
                    *o = (uint8_t)e1;
                    ++o;
                    --cnt;
                }
            }
            if (issue) RING_PUT(pend);
        }
    }
    if (!bad && n != -(int32_t)(8u * nbytes)) bad = true;
    if (__any(bad) && lane == 0) atomicAdd(verdict, 1u);
    if (lane == 0) myctl[4] = 1;
#undef HUF_PAIR
#undef QUAD
#undef ROOM
#undef RING_PUT
#undef FETCH
}

template <int RING, int SEG, int NDEC, bool NT>
static double run_k3(const Bench& B, int wg_per_cu, const char* label)
{
    auto k = stream_kernel3<RING, SEG, NDEC, NT>;
    hipFuncAttributes at;
    CK(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(k)));
    const uint32_t static_lds = (uint32_t)at.sharedSizeBytes;
    const uint32_t share = (163840u / (uint32_t)wg_per_cu) / 512u * 512u;
    if (static_lds > share) return -1.0;
    uint32_t pad = (share - static_lds) & ~3u;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, WAVE * (NDEC + 1), pad));
    const uint32_t blocks = (B.frames + NDEC - 1) / NDEC;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipMemset(B.d_verdict, 0, 4));
    k<<<blocks, WAVE * (NDEC + 1), pad>>>(B.d_src, B.src_stride, B.d_dst, B.dst_stride, B.d_tasks, B.d_table, B.d_verdict, pad / 4, B.frames);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (uint32_t r = 0; r < B.reps; ++r) {
        CK(hipEventRecord(a));
        k<<<blocks, WAVE * (NDEC + 1), pad>>>(B.d_src, B.src_stride, B.d_dst, B.dst_stride, B.d_tasks, B.d_table, B.d_verdict, pad / 4, B.frames);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
        sum += ms;
    }
    uint32_t verdict = 0;
    CK(hipMemcpy(&verdict, B.d_verdict, 4, hipMemcpyDeviceToHost));
    const double ms = sum / B.reps;
    printf("k3 %-24s ring %2d seg %3d B  %d decode waves + writer per workgroup, vgpr %3d lds %6u+%6u  %d wg/CU (API %d) = %d decode waves/CU  %7.3f ms (best %7.3f) %s\n",
           label, RING, SEG, NDEC, at.numRegs, static_lds, pad, wg_per_cu, occ, wg_per_cu * NDEC, ms, best, verdict ? "STREAM ERRORS" : "");
    if (B.csv)
        fprintf(B.csv, "k3 %s nt%d,%d,%d,0,%d,%u,%d,%d,%.4f,%.4f,0\n", label, (int)NT, RING, SEG, at.numRegs, static_lds + pad, wg_per_cu * NDEC, occ, ms, best);
    fflush(stdout);
    return ms;
}

// ---- the memory side alone: the lanes of a wave walk their streams like the decoder does (64-byte pieces of 64 different streams per
// instruction group, downwards through the input, upwards through the output) and do nothing else.  What this pattern gets from the
// memory system is the ceiling of the stream phase whatever the decoder's arithmetic costs.
template <int MODE, bool NT, int LP, int SP>
__global__ __launch_bounds__(WAVE) void pattern_kernel(const uint8_t* __restrict__ src, uint64_t src_stride, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                                       const Task* __restrict__ tasks, uint32_t pad_words, uint32_t delay)
{
    // LP / SP: bytes a lane reads / writes in one go (adjacent 16-byte accesses)
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (pad_words && lane == 0) reinterpret_cast<volatile uint32_t*>(dyn_lds)[pad_words - 1] = 0;
    const Task t = tasks[(size_t)f * STREAMS + lane];
    typedef __attribute__((address_space(1), aligned(1))) const u32x4 gq4;
    typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
    gcu8* p = (gcu8*)(src + (size_t)f * src_stride) + t.src;
    gu8* o = (gu8*)(dst + (size_t)f * dst_stride) + t.out;
    const uint32_t in_total = t.size / LP * LP, out_total = t.cnt / SP * SP;
    uint32_t in_done = 0, out_done = 0;
    u32x4 acc = { 1u, 2u, 3u, (uint32_t)lane };
    const uint32_t iters = (t.cnt + 63u) / 64u;  // one iteration per 64 output bytes, like the decoder's bursts
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t want_in = (uint32_t)(((uint64_t)(it + 1) * in_total) / iters), want_out = (uint32_t)(((uint64_t)(it + 1) * out_total) / iters);
        if ((MODE & 1) && in_done + LP <= want_in) {
#pragma unroll
            for (int k = 0; k < LP / 16; ++k) {
                const u32x4 v = *(gq4*)(p + (in_total - LP - in_done) + 16 * k);
                acc ^= v;
            }
            in_done += LP;
        }
        if ((MODE & 2) && out_done + SP <= want_out) {
#pragma unroll
            for (int k = 0; k < SP / 16; ++k) {
                if (NT) __builtin_nontemporal_store(acc, (gs4*)(o + out_done + 16 * k));
                else *(gs4*)(o + out_done + 16 * k) = acc;
            }
            out_done += SP;
        }
        for (uint32_t d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16);
    }
    if (!(MODE & 2) && acc.x == 0x7E57AB1Eu) *o = 1;
}

template <int MODE, bool NT, int LP = 64, int SP = 64>
static void run_pattern(const Bench& B, int waves_per_simd, uint32_t delay, const char* label)
{
    auto k = pattern_kernel<MODE, NT, LP, SP>;
    const uint32_t share = (163840u / (4u * (uint32_t)waves_per_simd)) / 512u * 512u;
    uint32_t pad = share & ~3u;
    if (waves_per_simd >= 8) pad = 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    k<<<B.frames, WAVE, pad>>>(B.d_src, B.src_stride, B.d_dst, B.dst_stride, B.d_tasks, pad / 4, delay);
    CK(hipDeviceSynchronize());
    float sum = 0;
    for (uint32_t r = 0; r < B.reps; ++r) {
        CK(hipEventRecord(a));
        k<<<B.frames, WAVE, pad>>>(B.d_src, B.src_stride, B.d_dst, B.dst_stride, B.d_tasks, pad / 4, delay);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        sum += ms;
    }
    const double ms = sum / B.reps;
    printf("pattern %-22s read %3d B write %3d B per lane and access  waves/SIMD %d delay %2u  %7.3f ms\n", label, LP, SP, waves_per_simd, delay, ms);
    fflush(stdout);
}

// ---- LDS counter calibration: ITER ds_read_u16 gathers per lane, lanes STRIDE u16 entries apart (mod 2048 entries).
// Known conflict degree: entries 2 apart = every lane its own dword of consecutive banks (conflict-free);
// 64 apart = dwords 32 apart = the 32 lanes of a group on ONE bank (32-way); 0 = one address (broadcast).
template <int STRIDE, bool DEP>
__global__ __launch_bounds__(WAVE) void lds_gather_kernel(uint32_t iters, uint32_t zero, uint32_t* out)
{
    __shared__ uint16_t T[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += WAVE) T[i] = (uint16_t)zero;  // zeros the compiler does not know to be zeros
    wave_lds_sync();
    const volatile uint16_t* V = T;  // every read is issued
    uint32_t idx = ((uint32_t)lane * STRIDE) & 2047u, acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t v = V[idx];
            acc += v;
            // the next address keeps the lanes' bank pattern (a multiple of 64 entries = 32 dwords further on); DEP: it waits for the value read
            idx = (idx + (DEP ? v : 0u) + 64u * (uint32_t)(k + 1)) & 2047u;
        }
    }
    if (acc == 0xFFFFFFFFu) out[lane] = acc;
}

template <int STRIDE, bool DEP>
static void run_calib(uint32_t blocks, uint32_t iters, uint32_t* d_out)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    lds_gather_kernel<STRIDE, DEP><<<blocks, WAVE>>>(iters, 0u, d_out);
    CK(hipEventRecord(a));
    lds_gather_kernel<STRIDE, DEP><<<blocks, WAVE>>>(iters, 0u, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double reads = (double)blocks * iters * 16.0;  // wave-level ds_read_u16 instructions
    printf("calib stride %3d %s: %8.3f ms, %u blocks x %u x 16 wave-level ds_read_u16 = %.3e, %.2f ns per CU per read (256 CUs)\n", STRIDE,
           DEP ? "dependent  " : "independent", ms, blocks, iters, reads, ms * 1e6 / (reads / 256.0));
}

// ---------------------------------------------------------------------------------------------------------------- driver

static int g_wg_per_cu = 0;  // != 0: single-wave workgroups per CU (for occupancies that are no multiple of four)
template <int RING, int BURST, bool LOADS, int STORES, int FMT, int KERNEL = 1>
static double run_one(const Bench& B, int waves_per_simd, const char* label)
{
    auto k = KERNEL == 1 ? stream_kernel<RING, BURST, LOADS, STORES, FMT> : (KERNEL == 2 ? stream_kernel2<RING, BURST, LOADS, STORES, FMT, 0> : (KERNEL == 3 ? stream_kernel2<RING, BURST, LOADS, STORES, FMT, 1> : stream_kernel2<RING, BURST, LOADS, STORES, FMT, 2>));
    hipFuncAttributes at;
    CK(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(k)));
    const uint32_t static_lds = (uint32_t)at.sharedSizeBytes;
    // LDS per workgroup such that exactly 4 * waves_per_simd single-wave workgroups fit a CU (160 KB, 512-byte granules);
    // a kernel that needs more than that share cannot run at this occupancy
    const uint32_t share = (163840u / (g_wg_per_cu ? (uint32_t)g_wg_per_cu : 4u * (uint32_t)waves_per_simd)) / 512u * 512u;
    if (static_lds > share) return -1.0;
    // not so much that one more fits
    uint32_t pad = share - static_lds;
    pad &= ~3u;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, WAVE, pad));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipMemset(B.d_verdict, 0, 4));
    k<<<B.frames, WAVE, pad>>>(B.d_src, B.src_stride, B.d_dst, B.dst_stride, B.d_tasks, B.d_table, B.d_verdict, pad / 4);  // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (uint32_t r = 0; r < B.reps; ++r) {
        CK(hipEventRecord(a));
        k<<<B.frames, WAVE, pad>>>(B.d_src, B.src_stride, B.d_dst, B.dst_stride, B.d_tasks, B.d_table, B.d_verdict, pad / 4);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
        sum += ms;
    }
    uint32_t verdict = 0;
    CK(hipMemcpy(&verdict, B.d_verdict, 4, hipMemcpyDeviceToHost));
    const double ms = sum / B.reps;
    // cycles a resident wave spends per symbol of its lanes: (resident waves) x time / (frames x symbols per lane)
    const double resident = std::min<double>((double)occ * 256.0, B.frames);
    const double cyc_per_sym = ms * 1e-3 * B.clock_ghz * 1e9 * resident / ((double)B.frames * B.cnt);
    printf("k%d %-34s ring %2d burst %2d B fmt %d  vgpr %3d lds %5u+%5u  waves/SIMD %d (API: %2d wg/CU)  %7.3f ms (best %7.3f)  %6.1f cycles/symbol/wave  %s\n",
           KERNEL, label, RING, BURST, FMT, at.numRegs, static_lds, pad, waves_per_simd, occ, ms, best, cyc_per_sym,
           (LOADS && verdict) ? "STREAM ERRORS" : "");
    (void)STORES;
    if (B.csv)
        fprintf(B.csv, "k%d %s,%d,%d,%d,%d,%u,%d,%d,%.4f,%.4f,%.2f\n", KERNEL, label, RING, BURST, FMT, at.numRegs, static_lds + pad, waves_per_simd, occ, ms, best,
                cyc_per_sym);
    fflush(stdout);
    return ms;
}

template <int RING, int BURST, int FMT>
static void sweep_modes(const Bench& B, int w)
{
    run_one<RING, BURST, true, 1, FMT>(B, w, "loads+stores");
    run_one<RING, BURST, true, 0, FMT>(B, w, "loads, no stores");
    run_one<RING, BURST, false, 1, FMT>(B, w, "no loads, stores");
    run_one<RING, BURST, false, 0, FMT>(B, w, "no loads, no stores");
}

int main(int argc, char** argv)
{
    uint32_t frames = 65536, distinct = 1024, reps = 5, cnt = 1720;
    bool calib = false, quick = false, mini = false;
    const char* csv_path = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--distinct") && i + 1 < argc) distinct = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--symbols") && i + 1 < argc) cnt = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--csv") && i + 1 < argc) csv_path = argv[++i];
        else if (!strcmp(argv[i], "--calib")) calib = true;
        else if (!strcmp(argv[i], "--quick")) quick = true;
        else if (!strcmp(argv[i], "--mini")) mini = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    if (calib) {
        uint32_t* d_out;
        CK(hipMalloc(&d_out, 4096));
        const uint32_t blocks = 256 * 16, iters = 4096;  // 16 waves per CU, 65536 gathers per wave
        run_calib<0, false>(blocks, iters, d_out);
        run_calib<2, false>(blocks, iters, d_out);
        run_calib<4, false>(blocks, iters, d_out);
        run_calib<8, false>(blocks, iters, d_out);
        run_calib<16, false>(blocks, iters, d_out);
        run_calib<64, false>(blocks, iters, d_out);
        run_calib<2, true>(blocks, iters, d_out);
        run_calib<8, true>(blocks, iters, d_out);
        return 0;
    }
    Code code;
    build_code(code);
    {
        double bits = 0, tot = 0;
        int hist[12] = {};
        for (int s = 0; s < 256; ++s) {
            bits += (double)HIST[s] * code.len[s];
            tot += HIST[s];
            hist[code.len[s]]++;
        }
        printf("code: %.3f bits per symbol; symbols per length:", bits / tot);
        for (int l = 1; l <= 11; ++l) printf(" %d:%d", l, hist[l]);
        printf("\n");
    }
    std::vector<uint32_t> cdf(256);
    {
        uint32_t a = 0;
        for (int s = 0; s < 256; ++s) cdf[s] = (a += HIST[s]);
    }
    const uint64_t src_stride = ((uint64_t)STREAMS * (cnt + 8) + 255) / 256 * 256 + 256;  // up to 8 bits per symbol on average (the code spends 6.4)
    const uint64_t dst_stride = ((uint64_t)STREAMS * cnt + 255) / 256 * 256;
    std::vector<uint8_t> h_src((size_t)distinct * src_stride), h_sym((size_t)distinct * dst_stride);
    std::vector<Task> h_tasks((size_t)distinct * STREAMS);
    {
        unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        std::vector<uint64_t> bytes(nt, 0);
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                for (uint32_t f = t; f < distinct; f += nt)
                    bytes[t] += make_frame(code, cdf, cnt, 0x1234 + f, h_src.data() + (size_t)f * src_stride, (uint32_t)src_stride - 64, h_sym.data() + (size_t)f * dst_stride,
                                           h_tasks.data() + (size_t)f * STREAMS);
            });
        for (auto& x : th) x.join();
        uint64_t all = std::accumulate(bytes.begin(), bytes.end(), 0ull);
        printf("%u distinct frames x %d streams x %u symbols; %.1f compressed bytes per frame (%.3f per symbol); replicated to %u frames (%.2f GB in, %.2f GB out)\n",
               distinct, STREAMS, cnt, (double)all / distinct, (double)all / distinct / (STREAMS * cnt), frames, frames * (double)src_stride / 1e9,
               frames * (double)dst_stride / 1e9);
    }
    Bench B;
    B.frames = frames;
    B.cnt = cnt;
    B.reps = reps;
    B.src_stride = src_stride;
    B.dst_stride = dst_stride;
    B.clock_ghz = prop.clockRate / 1e6;
    B.csv = csv_path ? fopen(csv_path, "w") : nullptr;
    if (B.csv) fprintf(B.csv, "mode,ring,burst,fmt,vgpr,lds_bytes,waves_per_simd,wg_per_cu_api,ms_avg,ms_best,cycles_per_symbol_per_wave\n");
    CK(hipMalloc(&B.d_src, (size_t)frames * src_stride + 512));
    B.d_src += 256;  // stream_kernel2 reads up to a batch below the start of a stream (production: a stream never starts an arena)
    CK(hipMalloc(&B.d_dst, (size_t)frames * dst_stride + 256));
    CK(hipMalloc(&B.d_tasks, (size_t)frames * STREAMS * sizeof(Task)));
    CK(hipMalloc(&B.d_table, sizeof(code.table)));
    CK(hipMalloc(&B.d_verdict, 64));
    CK(hipMemcpy(B.d_table, code.table, sizeof(code.table), hipMemcpyHostToDevice));
    for (uint32_t f0 = 0; f0 < frames; f0 += distinct) {
        const uint32_t m = std::min(distinct, frames - f0);
        if (f0 == 0) {
            CK(hipMemcpy(B.d_src, h_src.data(), (size_t)m * src_stride, hipMemcpyHostToDevice));
            CK(hipMemcpy(B.d_tasks, h_tasks.data(), (size_t)m * STREAMS * sizeof(Task), hipMemcpyHostToDevice));
        } else {
            CK(hipMemcpy(B.d_src + (size_t)f0 * src_stride, B.d_src, (size_t)m * src_stride, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(B.d_tasks + (size_t)f0 * STREAMS, B.d_tasks, (size_t)m * STREAMS * sizeof(Task), hipMemcpyDeviceToDevice));
        }
    }
    // correctness of the loop as compiled here: the first `distinct` frames against the host's symbols
    {
        CK(hipMemset(B.d_dst, 0xEE, (size_t)frames * dst_stride));
        CK(hipMemset(B.d_verdict, 0, 4));
        stream_kernel<32, 64, true, 1, 0><<<frames, WAVE>>>(B.d_src, src_stride, B.d_dst, dst_stride, B.d_tasks, B.d_table, B.d_verdict, 0);
        CK(hipDeviceSynchronize());
        std::vector<uint8_t> back((size_t)distinct * dst_stride);
        CK(hipMemcpy(back.data(), B.d_dst + (size_t)(frames - distinct) * dst_stride, back.size(), hipMemcpyDeviceToHost));
        size_t wrong = 0;
        for (uint32_t f = 0; f < distinct; ++f)
            wrong += memcmp(back.data() + (size_t)f * dst_stride, h_sym.data() + (size_t)f * dst_stride, (size_t)STREAMS * cnt) != 0;
        CK(hipMemset(B.d_dst, 0xEE, (size_t)frames * dst_stride));
        stream_kernel<16, 64, true, 2, 1><<<frames, WAVE>>>(B.d_src, src_stride, B.d_dst, dst_stride, B.d_tasks, B.d_table, B.d_verdict, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(back.data(), B.d_dst, back.size(), hipMemcpyDeviceToHost));
        for (uint32_t f = 0; f < distinct; ++f)
            wrong += memcmp(back.data() + (size_t)f * dst_stride, h_sym.data() + (size_t)f * dst_stride, (size_t)STREAMS * cnt) != 0;
        CK(hipMemset(B.d_dst, 0xEE, (size_t)frames * dst_stride));
        stream_kernel2<32, 64, true, 1, 0><<<frames, WAVE>>>(B.d_src, src_stride, B.d_dst, dst_stride, B.d_tasks, B.d_table, B.d_verdict, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(back.data(), B.d_dst, back.size(), hipMemcpyDeviceToHost));
        for (uint32_t f = 0; f < distinct; ++f)
            wrong += memcmp(back.data() + (size_t)f * dst_stride, h_sym.data() + (size_t)f * dst_stride, (size_t)STREAMS * cnt) != 0;
        CK(hipMemset(B.d_dst, 0xEE, (size_t)frames * dst_stride));
        stream_kernel2<16, 128, true, 2, 1><<<frames, WAVE>>>(B.d_src, src_stride, B.d_dst, dst_stride, B.d_tasks, B.d_table, B.d_verdict, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(back.data(), B.d_dst, back.size(), hipMemcpyDeviceToHost));
        for (uint32_t f = 0; f < distinct; ++f)
            wrong += memcmp(back.data() + (size_t)f * dst_stride, h_sym.data() + (size_t)f * dst_stride, (size_t)STREAMS * cnt) != 0;
        CK(hipMemset(B.d_dst, 0xEE, (size_t)frames * dst_stride));
        stream_kernel3<32, 64, 3, true><<<(frames + 2) / 3, WAVE * 4>>>(B.d_src, src_stride, B.d_dst, dst_stride, B.d_tasks, B.d_table, B.d_verdict, 0, frames);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(back.data(), B.d_dst, back.size(), hipMemcpyDeviceToHost));
        for (uint32_t f = 0; f < distinct; ++f)
            wrong += memcmp(back.data() + (size_t)f * dst_stride, h_sym.data() + (size_t)f * dst_stride, (size_t)STREAMS * cnt) != 0;
        uint32_t verdict = 0;
        CK(hipMemcpy(&verdict, B.d_verdict, 4, hipMemcpyDeviceToHost));
        printf("check: %zu frames differ from the host's symbols, %u streams reported corrupt\n", wrong, verdict);
        if (wrong || verdict) return 1;
    }
    const int sweep_w[] = { 1, 2, 3, 4, 5, 6 };
    if (mini) {  // a handful of launches for a PMC pass (FETCH_SIZE / WRITE_SIZE per kernel)
        B.reps = 2;
        run_one<32, 64, true, 1, 0, 1>(B, 2, "loads+stores");
        run_one<32, 64, true, 1, 0, 2>(B, 2, "loads+stores");
        run_one<32, 64, true, 0, 0, 2>(B, 2, "loads, no stores");
        run_one<32, 64, false, 1, 0, 2>(B, 2, "no loads, stores");
        run_one<16, 128, true, 2, 0, 2>(B, 2, "loads + staged stores");
        run_one<16, 128, false, 2, 0, 1>(B, 2, "no loads, staged stores");
        run_one<32, 64, true, 1, 0, 2>(B, 1, "loads+stores");
        for (int cu : { 5, 6, 7 }) {
            g_wg_per_cu = cu;
            run_one<64, 128, true, 1, 0, 2>(B, 0, "128 B fetch + 128 B bursts");
            run_one<64, 128, true, 0, 0, 2>(B, 0, "128 B fetch, no stores");
            run_one<64, 128, false, 1, 0, 2>(B, 0, "no loads, 128 B bursts");
            run_one<64, 128, false, 0, 0, 2>(B, 0, "no loads, no stores");
            run_one<64, 64, true, 1, 0, 2>(B, 0, "128 B fetch + 64 B bursts");
            run_one<32, 128, true, 1, 0, 2>(B, 0, "64 B fetch + 128 B bursts");
            run_one<32, 64, true, 1, 0, 2>(B, 0, "64 B fetch + 64 B bursts");
        }
        g_wg_per_cu = 0;
        for (int w = 1; w <= 3; ++w) {
            run_one<64, 128, true, 1, 0, 2>(B, w, "128 B fetch + 128 B bursts");
            run_one<64, 128, true, 0, 0, 2>(B, w, "128 B fetch, no stores");
            run_one<64, 128, false, 1, 0, 2>(B, w, "no loads, 128 B bursts");
            run_one<64, 64, true, 1, 0, 2>(B, w, "128 B fetch + 64 B bursts");
            run_one<32, 128, true, 1, 0, 2>(B, w, "64 B fetch + 128 B bursts");
            run_one<32, 64, true, 1, 0, 2>(B, w, "64 B fetch + 64 B bursts");
        }
        for (int w : { 2, 8 }) {
            run_pattern<1, false, 64, 64>(B, w, 0, "loads");
            run_pattern<2, false, 64, 64>(B, w, 0, "stores");
            run_pattern<3, false, 64, 64>(B, w, 0, "loads+stores");
            run_pattern<1, false, 128, 128>(B, w, 0, "loads");
            run_pattern<2, false, 128, 128>(B, w, 0, "stores");
            run_pattern<3, false, 128, 128>(B, w, 0, "loads+stores");
            run_pattern<3, false, 64, 128>(B, w, 0, "loads+stores");
            run_pattern<3, false, 128, 64>(B, w, 0, "loads+stores");
            run_pattern<3, false, 256, 256>(B, w, 0, "loads+stores");
            run_pattern<3, false, 64, 256>(B, w, 0, "loads+stores");
            run_pattern<3, false, 256, 64>(B, w, 0, "loads+stores");
            run_pattern<3, false, 512, 512>(B, w, 0, "loads+stores");
            run_pattern<3, false, 32, 32>(B, w, 0, "loads+stores");
            run_pattern<3, false, 16, 16>(B, w, 0, "loads+stores");
        }
        if (mini) return 0;
        for (int w = 2; w <= 3; ++w) {
            run_one<32, 64, true, 1, 0, 2>(B, w, "loads+stores");
            run_one<32, 64, true, 1, 0, 3>(B, w, "prefetch, loads+stores");
            run_one<32, 64, true, 1, 0, 4>(B, w, "prefetch+wait, loads+stores");
            run_one<32, 64, true, 0, 0, 3>(B, w, "prefetch, loads");
            run_one<32, 128, true, 1, 0, 3>(B, w, "prefetch, loads+128 B bursts");
            run_one<16, 128, true, 4, 0, 3>(B, w, "prefetch, staged 128 nt");
            run_one<16, 64, true, 1, 0, 3>(B, w, "prefetch, loads+stores");
        }
        run_k3<32, 64, 3, false>(B, 3, "writer wave");
        run_k3<32, 64, 3, true>(B, 3, "writer wave nt");
        run_k3<32, 64, 4, true>(B, 2, "writer wave nt");
        run_k3<32, 128, 3, true>(B, 2, "writer wave nt");
        run_k3<16, 64, 3, true>(B, 3, "writer wave nt");
        run_k3<16, 64, 3, true>(B, 4, "writer wave nt");
        run_k3<16, 64, 4, true>(B, 3, "writer wave nt");
        for (int w = 2; w <= 3; ++w) {
            run_one<32, 128, true, 1, 0, 2>(B, w, "loads + 128 B bursts");
            run_one<32, 128, true, 3, 0, 2>(B, w, "loads + 128 B bursts nt");
            run_one<32, 64, true, 3, 0, 2>(B, w, "loads + 64 B bursts nt");
            run_one<32, 128, true, 2, 0, 2>(B, w, "loads + staged 128");
            run_one<32, 128, true, 4, 0, 2>(B, w, "loads + staged 128 nt");
            run_one<16, 128, true, 4, 0, 2>(B, w, "loads + staged 128 nt");
            run_one<32, 64, true, 4, 0, 2>(B, w, "loads + staged 64 nt");
            run_one<32, 128, false, 4, 0, 2>(B, w, "no loads, staged 128 nt");
        }
        return 0;
    }
    printf("\n== production loop (ring 32, 64-byte bursts, entries symbol | nb << 8): the kernel's stream phase at 1..3 waves per SIMD ==\n");
    for (int w : sweep_w) {
        if (quick && w != 2 && w != 4) continue;
        sweep_modes<32, 64, 0>(B, w);
    }
    printf("\n== ring 16, 32-byte bursts ==\n");
    for (int w : sweep_w) {
        if (quick && w != 2 && w != 4) continue;
        sweep_modes<16, 32, 0>(B, w);
    }
    printf("\n== ring 8, 32-byte bursts ==\n");
    for (int w : sweep_w) {
        if (quick && w != 2 && w != 4 && w != 6) continue;
        sweep_modes<8, 32, 0>(B, w);
    }
    printf("\n== entries nb | symbol << 8 (one shift less on the chain) ==\n");
    for (int w : sweep_w) {
        if (quick && w != 4) continue;
        run_one<32, 64, true, 1, 1>(B, w, "loads+stores");
        run_one<16, 32, true, 1, 1>(B, w, "loads+stores");
        run_one<16, 64, true, 1, 1>(B, w, "loads+stores");
        run_one<8, 16, true, 1, 1>(B, w, "loads+stores");
        run_one<8, 32, true, 1, 1>(B, w, "loads+stores");
        run_one<8, 64, true, 1, 1>(B, w, "loads+stores");
        run_one<8, 32, false, 0, 1>(B, w, "no loads, no stores");
    }
    printf("\n== stores staged through LDS: whole 64- / 128-byte segments written by 4 / 8 adjacent lanes ==\n");
    for (int w : sweep_w) {
        run_one<32, 64, true, 2, 0>(B, w, "loads + staged stores");
        run_one<32, 128, true, 2, 0>(B, w, "loads + staged stores");
        run_one<16, 64, true, 2, 0>(B, w, "loads + staged stores");
        run_one<16, 128, true, 2, 0>(B, w, "loads + staged stores");
        run_one<16, 64, false, 2, 0>(B, w, "no loads, staged stores");
        run_one<16, 128, false, 2, 0>(B, w, "no loads, staged stores");
        run_one<8, 64, true, 2, 0>(B, w, "loads + staged stores");
    }
    printf("\n== restructured top-up (request, decode, store, commit in one straight line: counted waits) ==\n");
    for (int w : sweep_w) {
        run_one<32, 64, true, 1, 0, 2>(B, w, "loads+stores");
        run_one<32, 64, true, 0, 0, 2>(B, w, "loads, no stores");
        run_one<32, 64, false, 1, 0, 2>(B, w, "no loads, stores");
        run_one<32, 64, false, 0, 0, 2>(B, w, "no loads, no stores");
        run_one<16, 64, true, 1, 0, 2>(B, w, "loads+stores");
        run_one<16, 32, true, 1, 0, 2>(B, w, "loads+stores");
        run_one<16, 64, false, 0, 0, 2>(B, w, "no loads, no stores");
        run_one<32, 128, true, 2, 0, 2>(B, w, "loads + staged stores");
        run_one<16, 128, true, 2, 0, 2>(B, w, "loads + staged stores");
        run_one<16, 64, true, 2, 0, 2>(B, w, "loads + staged stores");
        run_one<32, 64, true, 1, 1, 2>(B, w, "loads+stores");
    }
    if (B.csv) fclose(B.csv);
    return 0;
}
