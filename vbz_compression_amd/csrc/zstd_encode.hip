// zstd_encode.hip -- the entropy stage of VBZ on gfx950: a zstd-FORMAT (RFC 8878) frame encoder.
//
// Replaces the reference's ZSTD_compress call (vbz/vbz.cpp:194-207; external libzstd 1.4.8).
// The output is a standard single-segment zstd frame that libzstd -- and therefore the reference's
// vbz_decompress -- decodes to exactly the svb stream it was given.  It is NOT byte-identical to
// libzstd's output: libzstd's level-1 match finder is a serial hash-chain walk with no parallel
// form, and on nanopore signal >= 98 % of its output bytes are Huffman-coded literals anyway
// (SURVEY.md section 0.4).  This encoder therefore spends its effort where the bytes are:
//
//   * the svb stream is cut into two REGIONS, control bytes and data bytes, because their byte
//     statistics differ completely (control bytes are mostly 0x00); each region gets its own
//     Huffman table: code lengths by package-merge (optimal under the 11-bit limit; libzstd's own
//     construction, a serial merge chain, is never shorter), tree description written the way
//     libzstd writes it (zstd_entropy.h holds the serial statements of both);
//   * the data-byte region is cut into near-equal BLOCKS of 4 Huffman streams each; the first block
//     carries the tree description, the others are "treeless" (reuse the table), which the format
//     allows.  A frame thus exposes up to 64 independent bit streams: one per lane of the wavefront
//     that decodes the frame (zstd_decode.hip);
//   * the control-byte region becomes one block whose long zero runs are zstd SEQUENCES (copy from
//     offset 1 = repeat offset 1, which costs no bits) -- what libzstd's match finder gets out of
//     that region, found here with a bit-parallel tokeniser instead of a hash chain -- followed by
//     a skippable frame with decoder checkpoints (CP_MAGIC below);
//   * raw and RLE blocks are used where Huffman coding does not pay (tiny or constant regions), which
//     also reproduces the reference's known answers for tiny inputs (vbz/test/vbz_test.cpp:238).
//
// One wavefront (64 lanes) per frame, 16 waves per CU.  Per region: histogram (LDS atomics into four
// private copies; a quarter of the bytes where that is enough) -> table construction (all lanes: bitonic
// sort, package-merge, two-lane FSE chains for the tree description) -> packing: the wave packs one
// stream at a time, 1024 symbols per step, with a prefix sum of bit counts; streams are written in
// frame order and block headers are filled in last.  The kernel's time follows the number of
// instructions a wavefront executes, so nothing in it is left to one lane that all lanes can do.
// Algorithmic HBM bytes per svb byte: 1 read + ~0.67 written.
#include "vbz_kernels.h"
#include "zstd_entropy.h"

namespace vbzhip {

namespace {

constexpr int WAVE = 64;
constexpr uint32_t BLOCK_MAX = 128u << 10;
constexpr uint32_t MIN_BLOCK = 4096;      // target block size is at least this
constexpr uint32_t HUF_BLOCK_MAX = 92u << 10;   // content of a Huffman-coded block: 11/8 of it + tree + headers < BLOCK_MAX
constexpr uint32_t SPLIT_MIN = 2048;      // frames smaller than this are one region
constexpr int MAXBLK = 16;                // blocks handled per pass (x4 streams = 64 lanes)
// Large reads (span mode, see EncSpan): blocks of at most SPAN_BLOCK bytes, SPAN_BLOCKS of them per span.  A span is what ONE
// wavefront encodes -- its own histogram, its own Huffman table (first block with the tree, the others treeless) -- into a
// temporary slot; a compaction pass then strings the spans of a frame together.  16 KB blocks keep the decoder's streams
// short (4 KB of content per lane) at 0.1 % of header overhead.
#ifndef VBZ_SPAN_BLOCK_KB
#define VBZ_SPAN_BLOCK_KB 16
#endif
constexpr uint32_t SPAN_BLOCK = VBZ_SPAN_BLOCK_KB << 10;
// blocks per span: 4 (64 KB of stream per wavefront: a 40 MB buffer is 300 spans, one 400 k-sample read 8) up to 256 MB of
// stream, 16 beyond (the decoder then has all 64 lanes of a wavefront busy).  A span costs its tree description, ~0.15 %.
#ifndef VBZ_SPAN_SMALL_BLOCKS
#define VBZ_SPAN_SMALL_BLOCKS 2
#endif
constexpr uint32_t SPAN_BYTES_SMALL = SPAN_BLOCK * VBZ_SPAN_SMALL_BLOCKS, SPAN_BYTES_LARGE = SPAN_BLOCK * MAXBLK;
constexpr uint32_t SPAN_LARGE_FROM = 256u << 20;
// The control-byte region is cut into spans of at most KEYSPAN_BYTES (twice that from KEYSPAN_LARGE_FROM control bytes on),
// each ONE block whose zero runs become sequences (what the one-wavefront path does with the whole region): tokenising and
// the serial sequence chain of the decoder are the long pole of a lone wavefront, so these spans are short.  Measured: one
// 400 k-sample read decodes in 0.23 ms with 16 KB spans, 0.17 ms with 8 KB (ratio 2.3923 -> 2.3912); eight 40 MB buffers
// per call, which have spans enough either way, lose 4 % of their decode rate to the extra spans -- hence the two sizes.
#ifndef VBZ_KEYSPAN_KB
#define VBZ_KEYSPAN_KB 8
#endif
constexpr uint32_t KEYSPAN_BYTES = VBZ_KEYSPAN_KB << 10, KEYSPAN_LARGE_FROM = 1u << 20;
// ... and half of it in calls with shared tables, which are a matter of latency: a control-byte span behind the first has no
// checkpoints for its sequence chain (the trailer describes the frame's first sequences section), and the decoder's wavefront walks
// the ~55 sequences per kilobyte one after the other -- 110 k cycles for a span of 6 KB, the longest wavefront of a decompress call
// (profiles/r05_experiments.md)
#ifndef VBZ_KEYSPAN_SHARED_KB
#define VBZ_KEYSPAN_SHARED_KB 4
#endif
constexpr uint32_t KEYSPAN_BYTES_SHARED = VBZ_KEYSPAN_SHARED_KB << 10;
__host__ __device__ constexpr uint32_t keyspan_bytes_for(uint32_t K) { return K >= KEYSPAN_LARGE_FROM ? 2u * KEYSPAN_BYTES : KEYSPAN_BYTES; }
__host__ __device__ constexpr uint32_t span_bytes_for(uint32_t N) { return N >= SPAN_LARGE_FROM ? SPAN_BYTES_LARGE : SPAN_BYTES_SMALL; }
#ifndef VBZ_STEP_LANE
#define VBZ_STEP_LANE 16
#endif
constexpr int STEP_LANE = VBZ_STEP_LANE;    // symbols packed per lane per step
static_assert(STEP_LANE == 16, "the packing step is written for 16 symbols per lane");
constexpr int STEP_DW = STEP_LANE / 4;      // dwords per lane chunk
constexpr int STEP_SYMS = WAVE * STEP_LANE; // symbols packed per wave step
constexpr int OBUF_WORDS = (STEP_SYMS * 11) / 32 + 8;  // one step packs at most 2048 symbols of 11 bits

// One span of a frame in span mode (built by zstd_span_plan_kernel, one per workgroup of the encode kernel).
struct EncSpan
{
    uint32_t read;       // index of the read in the batch
    uint32_t r0, r1;     // byte range of the svb stream this span codes
    uint32_t flags;      // SPAN_*
    uint64_t tmp_off;    // where its output goes in the temporary arena
    uint32_t tmp_cap;
    uint32_t ord;        // ordinal of the span in its frame
};
constexpr uint32_t SPAN_FIRST = 1, SPAN_LAST = 2, SPAN_KEYSEQ = 4, SPAN_SKIP = 8;
constexpr uint32_t SPAN_SHARED = 16;   // a span of the data bytes that is packed with the table of the read's whole data-byte region
constexpr uint32_t SPAN_TREE = 32;     // ... and carries that table's description (the first of them)

// Shared tables (round 5).  A lone wavefront needs ~25 us to build a Huffman table and ~15 us to pack 8 KB with it; with a table per
// 32 KB span the table construction was the larger part of the 0.12 ms the entropy stage took for ONE large read.  The data bytes of a
// read with a control-byte region now get one table for the whole region.  The launch that codes the control-byte spans (which keep
// a table each: their literals are known only after the tokeniser) has one more wavefront per read in the TABLE ROLE: it builds the
// table of the read's data bytes with region_plan, as for any region, from the histogram zstd_span_count_kernel has left -- so the
// table construction (60 us for a lone wavefront and 256 symbols) runs beside the control-byte spans, not in front of them.
// zstd_span_pack_kernel then packs the data bytes in spans of SHSPAN_BYTES,
// one block each, the first with the tree description, the others treeless.  The decoder gives every treeless span the block with
// the tree (zstd_decode.hip: DecSpan::tree_pos) and verifies that it was the one in force.
// Whether a call uses them is the host's choice (zstd_span_shared_bytes): yes, with spans of 8 KB, one block, while the call's spans fit
// the device at once -- there the longest wavefront is the call's latency --; no beyond: a batch of large buffers is a matter of
// throughput, where three launches one after the other lose against one launch in which every span builds its own table (measured:
// eight 40 MB buffers per call, zstd_encode 0.38 -> 0.49 ms with shared tables and spans of 32 KB; profiles/r05_experiments.md).
// (tools/time_small_batch.py, reads of 100 k samples, compress / decompress ms per call with and without: 8 reads 0.125 / 0.142 against
// 0.160 / 0.174, 32 reads 0.156 / 0.180 against 0.185 / 0.199, 100 reads 0.247 / 0.324 against 0.276 / 0.291, 200 reads 0.372 / 0.585 against
// 0.388 / 0.417: from ~2 000 spans on every treeless span's reading of the tree is throughput lost, not latency hidden.)
constexpr uint32_t SHSPAN_BYTES = 8u << 10, SHSPAN_BATCH_FROM = 20u << 20;   // (of the bound on the call's stream bytes)
constexpr uint32_t SHSPAN_BYTES_LARGE = 4u * SPAN_BLOCK;  // reads of SPAN_LARGE_FROM bytes of stream and more
constexpr uint32_t SHSPAN_MIN_REGION = 2u * SHSPAN_BYTES;

// per read: the data-byte region's histogram, the ticket of the workgroups that count it, its table
struct SpanRegion
{
    uint32_t hist[256];     // (the first SPANREGION_ZEROED words are zeroed by zstd_span_plan_kernel)
    uint32_t done;          // (unused)
    uint32_t nshared;       // shared spans of the read (0: none -- the read's spans bring their own tables)
    uint32_t mode;          // 0 raw, 1 rle, 2 huffman (region_plan's verdict on the whole region)
    uint32_t treeSize, huffLog;
    uint32_t pad0[3];
    uint32_t ctable[256];   // code | length << 16
    uint32_t tree[34];
    uint32_t pad1[2];
};
constexpr uint32_t SPANREGION_ZEROED = 264;
static_assert(sizeof(SpanRegion) % 16 == 0 && offsetof(SpanRegion, ctable) == 4 * SPANREGION_ZEROED, "SpanRegion");

struct EncLds
{
    uint32_t hist[256];
    uint2 ctable[256];     // { code, length }: one 8-byte read per symbol, nothing to take apart in the packing loop
    uint8_t nbBits[256];
    uint8_t weights[260];
    uint8_t tree[136];
    int32_t treeSize;
    uint32_t mode;      // 0 raw, 1 rle, 2 huffman
    uint32_t huffLog;
    uint32_t rankcnt[16];   // symbols per code length
    // A region is planned (table construction) before anything of it is sized or packed, and the next region is
    // planned only after this one is written: the two workspaces share their LDS (16 waves per CU instead of 13).
    union
    {
        HufPmWksp pm;       // table construction: sorted leaves, packages, per-level package masks
        FseWeightWksp fw;   // FSE coding of the weights (after the construction)
        struct
        {
            uint32_t ssize[WAVE];   // compressed bytes of each stream of the current pass
            uint32_t sbeg[WAVE];    // first byte (region offset) of each stream of the current pass
            uint32_t scnt[WAVE];    // symbols in each stream
            uint32_t obuf[OBUF_WORDS]; // bit buffer of the stream being packed
            uint16_t seqcode[8 + 256];  // sequences section: codes of a round (+ the tail of the round before)
            uint32_t seqpiece[256];     //                    what the state chains put out
        };
    };
    SeqCTables seq;             // encoding tables of the predefined LL / ML distributions
    uint32_t cp[64];            // decoder checkpoints of the sequences section (see CP_MAGIC)
    uint32_t cpCount, cpSpacing;
    __device__ __forceinline__ void set_code(uint32_t s, uint32_t code, uint32_t nb) { ctable[s] = make_uint2(code, nb); }
};

// The staged encoder's planning launch (zstd_plan_kernel) works on the same functions with LDS of its own, two wavefronts per read:
// a region's table -- histogram, code lengths, codes, tree description
struct TableLds
{
    uint32_t hist[256];
    uint32_t ctable[256];      // code | length << 16 (what the plan stores)
    uint8_t nbBits[256];
    uint8_t weights[260];
    uint8_t tree[136];
    int32_t treeSize;
    uint32_t mode;      // 0 raw, 1 rle, 2 huffman
    uint32_t huffLog;
    uint32_t rankcnt[16];
    union
    {
        HufPmWksp pm;
        FseWeightWksp fw;
    };
    __device__ __forceinline__ void set_code(uint32_t s, uint32_t code, uint32_t nb) { ctable[s] = code | (nb << 16); }
};
// the control bytes' tokeniser (no LDS at all) and their sequences section
struct TokSeqLds
{
    uint32_t obuf[128];         // bit buffer of the section
    uint16_t seqcode[8 + 256];
    uint32_t seqpiece[256];
    SeqCTables seq;
    uint32_t cp[64];
    uint32_t cpCount, cpSpacing;
};

__device__ __forceinline__ void put_le(uint8_t* p, uint64_t v, int n)
{
    for (int i = 0; i < n; ++i) {
        uint32_t byte = (uint32_t)(v >> (8 * i)) & 0xFFu;
        // One plain byte store per byte.  Without this barrier hipcc (ROCm 7.2, gfx950) fuses the stores of a 3-byte
        // block header into a 16-bit store built with v_lshrrev_b32_sdwa + v_bitop3_b16, and the second byte came out
        // wrong on hardware for some builds (caught by the libzstd cross-decode test).
        asm volatile("" : "+v"(byte));
        p[i] = (uint8_t)byte;
    }
}

// Reductions over the wavefront (all 64 lanes active): DPP row shifts and broadcasts bring the result to lane 63, v_readlane hands it to
// everybody in a scalar register -- six butterfly shuffles through the LDS crossbar were six round trips on a dependent chain.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#define VBZ_DPP_MAX(ctrl, rowmask) do { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xF, false); v = o_ > v ? o_ : v; } while (0)
    VBZ_DPP_MAX(0x111, 0xF);  // row_shr:1 (a lane without a source reads 0: the identity of an unsigned maximum)
    VBZ_DPP_MAX(0x112, 0xF);
    VBZ_DPP_MAX(0x114, 0xF);
    VBZ_DPP_MAX(0x118, 0xF);
    VBZ_DPP_MAX(0x142, 0xA);  // row_bcast:15
    VBZ_DPP_MAX(0x143, 0xC);  // row_bcast:31
#undef VBZ_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// histogram of in[0..n) into L.hist using all 64 lanes.  LDS atomics on a shared bin serialise, so four lane
// groups count into four private copies (in the table-construction workspace, idle at this point) that are summed at
// the end; every byte is one branch-free ds_add.
//
// Regions of HIST_SAMPLE_FROM bytes or more are first counted on a quarter of their bytes (one kilobyte out of every four,
// plus the unaligned ends).  If that sample shows HIST_SAMPLE_SEEN or more of the 256 byte values, the table is built from
// it, every value it missed counted once: every byte of the region has a word, the code is within 0.03 % of the one the
// full count would give on signal data, and at worst -- sixteen values that really never occur -- 0.2 % longer.  Otherwise
// (narrow distributions, where words for absent values would cost more) the other three quarters are added and the
// histogram is exact, as for the smaller regions.  Returns the number of bytes counted into L.hist.
constexpr uint32_t HIST_SAMPLE_FROM = 32u << 10, HIST_SAMPLE_SEEN = 240;

// the bytes of in[0..n) the sample counts: the unaligned ends and one stripe of 64 aligned 16-byte chunks in four
__device__ __forceinline__ uint32_t hist_sample_bytes(const uint8_t* in, uint32_t n)
{
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)in & 15u)) & 15u);
    const uint32_t h = head < n ? head : n;
    const uint32_t nvec = (n - h) >> 4, tail0 = h + (nvec << 4);
    const uint32_t stripes = (nvec + WAVE - 1) / WAVE, groups = stripes >> 2, last = stripes & 3u;
    uint32_t chunks = groups * WAVE;     // sampled chunks: the first stripe of every group of four ...
    if (last) chunks += (nvec - groups * 4u * WAVE) < (uint32_t)WAVE ? (nvec - groups * 4u * WAVE) : (uint32_t)WAVE;
    return h + (n - tail0) + 16u * chunks;
}

template <class LDS>
__device__ __forceinline__ uint32_t region_histogram(LDS& L, const uint8_t* in, uint32_t n, int lane, bool allow_sample);

// The same histogram from what the svb encoder left in the read's plan (EncPlan::hist_mode; svb_kernels.hip TOK counts exactly the
// bytes region_histogram would: `sample` = the sampled data bytes, `rest` = the others or nullptr): L.hist and the return value are
// region_histogram's.  What the hand-over cannot answer (an exact count without `rest`) is counted from memory as before.
template <class LDS>
__device__ __forceinline__ uint32_t region_histogram_from_plan(LDS& L, const uint8_t* in, uint32_t n, int lane, bool allow_sample, const uint32_t* sample, const uint32_t* rest)
{
    uint32_t seen = 0;
    uint32_t c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = sample[lane + 64 * j];
        seen += c[j] ? 1u : 0u;
    }
    seen = wave_sum_u32(seen);
    if (allow_sample && n >= HIST_SAMPLE_FROM && seen >= HIST_SAMPLE_SEEN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) L.hist[lane + 64 * j] = c[j] ? c[j] : 1u;
        wave_lds_sync();
        return hist_sample_bytes(in, n) + (256u - seen);
    }
    if (!rest) return region_histogram(L, in, n, lane, allow_sample);
#pragma unroll
    for (int j = 0; j < 4; ++j) L.hist[lane + 64 * j] = c[j] + rest[lane + 64 * j];
    wave_lds_sync();
    return n;
}

template <class LDS>
__device__ __forceinline__ uint32_t region_histogram(LDS& L, const uint8_t* in, uint32_t n, int lane, bool allow_sample)
{
    static_assert(sizeof(HufPmWksp) >= 4 * 256 * sizeof(uint32_t), "the sub-histograms live in the table-construction workspace");
    uint32_t* sub = reinterpret_cast<uint32_t*>(&L.pm);
    for (int i = lane; i < 4 * 256; i += WAVE) sub[i] = 0;
    wave_lds_sync();
    uint32_t* mine = sub + 256 * (lane & 3);
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)in & 15u)) & 15u);
    const uint32_t h = head < n ? head : n;
    if ((uint32_t)lane < h) atomicAdd(&mine[in[lane]], 1u);
    const uint32_t nvec = (n - h) >> 4;
    const uint4* vp = reinterpret_cast<const uint4*>(in + h);
    const uint32_t tail0 = h + (nvec << 4);
    if (tail0 + (uint32_t)lane < n) atomicAdd(&mine[in[tail0 + lane]], 1u);
    // chunk c (16 bytes) belongs to stripe c / WAVE; LANES stripes, STEP stripes apart, are in flight per trip
    auto count = [&](uint32_t first_stripe, uint32_t stripe_step, uint32_t trip_stripes, int loads) {
        for (uint32_t s0 = first_stripe; s0 * WAVE < nvec; s0 += trip_stripes) {
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // independent 16-byte loads in flight per lane
                const uint32_t c = (s0 + (uint32_t)u * stripe_step) * WAVE + (uint32_t)lane;
                q[u] = (u < loads && c < nvec) ? vp[c] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t c = (s0 + (uint32_t)u * stripe_step) * WAVE + (uint32_t)lane;
                if (u < loads && c < nvec) {
                    const uint32_t w[4] = { q[u].x, q[u].y, q[u].z, q[u].w };
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(&mine[(w[k] >> (8 * j)) & 0xFFu], 1u);
                    }
                }
            }
        }
    };
    auto total = [&]() -> uint32_t {   // sub-histograms -> L.hist; returns the number of byte values seen
        wave_lds_sync();
        uint32_t seen = 0;
        for (int i = lane; i < 256; i += WAVE) {
            const uint32_t c = sub[i] + sub[256 + i] + sub[512 + i] + sub[768 + i];
            L.hist[i] = c;
            seen += c ? 1u : 0u;
        }
        seen = wave_sum_u32(seen);
        wave_lds_sync();
        return seen;
    };
    if (allow_sample && n >= HIST_SAMPLE_FROM) {
        count(0, 4, 16, 4);                      // stripes 0, 4, 8, 12 of every sixteen
        const uint32_t seen = total();
        if (seen >= HIST_SAMPLE_SEEN) {
            for (int i = lane; i < 256; i += WAVE)
                if (L.hist[i] == 0) L.hist[i] = 1;   // a value the sample missed may still occur: it gets a (long) word
            wave_lds_sync();
            return hist_sample_bytes(in, n) + (256u - seen);
        }
        count(1, 1, 4, 3);                       // the rest: stripes 1, 2, 3 of every four
    } else {
        count(0, 1, 4, 4);
    }
    (void)total();
    return n;
}

// ---- table construction, wave-cooperative ---------------------------------------------------------------
// Code lengths: the present symbols are sorted by a bitonic network over the 64 lanes (four keys per lane), the optimal
// lengths under the limit come from package-merge with every list of a level merged by the whole wavefront (binary
// searches of the leaves among the packages and of the packages among the leaves), canonical codes from ballots in symbol
// order -- the same lengths and codes as huf_build_pm() of zstd_entropy.h, its serial statement, which the CPU suite checks
// for optimality and the GPU suite against the tree descriptions in the device's frames.  libzstd builds the unlimited
// Huffman tree and repairs it (HUF_setMaxHeight); package-merge is never longer (0.02 % shorter on signal data) and has no
// serial chain of 255 merges.  The tree description (huf_write_tree_wave below) is libzstd's for these lengths.
// all lanes.  L.hist[0..maxSym] -> L.nbBits / L.ctable; returns the table log (uniform).
// sub-phase timers of the timed kernel build (tools/phase_timing.py): slots 6.. of the phase counters
#define SUB(k) do { if (tsub) { unsigned long long tn = __builtin_readcyclecounter(); tsub[k] += tn - *tl; *tl = tn; } } while (0)
// #{i : arr[i] < v} (STRICT) or #{i : arr[i] <= v} over a sorted array padded with 0xFFFFFFFF; at most 255 (top = the
// largest power of two <= the number of real entries, wave-uniform).  One LDS read and three VALU operations per step.
template <bool STRICT>
__device__ __forceinline__ uint32_t sorted_count(const uint32_t* arr, uint32_t v, uint32_t top)
{
    uint32_t lo = 0;
    for (uint32_t step = top; step; step >>= 1) {
        const uint32_t x = arr[lo + step - 1];
        lo += (STRICT ? x < v : x <= v) ? step : 0u;
    }
    return lo;
}

template <class LDS>
__device__ __forceinline__ uint32_t huf_build_wave(LDS& L, uint32_t maxSym, uint32_t maxNbBits, int lane, unsigned long long* tsub = nullptr, unsigned long long* tl = nullptr)
{
    HufPmWksp& K = L.pm;
    // --- sort the present symbols by (count, 255 - symbol), ascending: bitonic network over 256 keys, element e = 4*lane + j
    uint32_t key[4];
    {
        const uint4 h = *reinterpret_cast<const uint4*>(&L.hist[4 * lane]);
        const uint32_t c[4] = { h.x, h.y, h.z, h.w };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t sy = 4u * (uint32_t)lane + (uint32_t)j;
            key[j] = (sy <= maxSym && c[j]) ? ((c[j] << 8) | (255u - sy)) : 0xFFFFFFFFu;
        }
    }
    for (int i = lane; i < (HUF_MAX_BITS + 1) * 16; i += WAVE) (&K.isPkg[0][0])[i] = 0;
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int d = k >> 1; d > 0; d >>= 1) {
            if (d >= 4) {
                const int m = d >> 2;  // the partner is m lanes away, same j
                const bool up = ((4 * lane) & k) == 0, lower = (lane & m) == 0;
                const bool keep_min = lower == up;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)key[j], m, 64);
                    const uint32_t mn = o < key[j] ? o : key[j], mx = o < key[j] ? key[j] : o;
                    key[j] = keep_min ? mn : mx;
                }
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int bb = a ^ d;
                    if (bb > a) {
                        const bool up = ((4 * lane + a) & k) == 0;
                        const uint32_t mn = key[a] < key[bb] ? key[a] : key[bb], mx = key[a] < key[bb] ? key[bb] : key[a];
                        key[a] = up ? mn : mx;
                        key[bb] = up ? mx : mn;
                    }
                }
            }
        }
    }
    uint32_t n = 0;
    {
        uint32_t w4 = 0;
        uint4 lf;
        uint32_t* lfp = &lf.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool present = key[j] != 0xFFFFFFFFu;
            n += (uint32_t)__popcll(__ballot(present));
            lfp[j] = present ? key[j] >> 8 : 0xFFFFFFFFu;
            w4 |= (255u - (key[j] & 255u)) << (8 * j);
        }
        *reinterpret_cast<uint4*>(&K.leaf[4 * lane]) = lf;
        *reinterpret_cast<uint32_t*>(&K.sym[4 * lane]) = w4;
    }
    wave_lds_sync();
    SUB(6);
    // --- package-merge (see huf_package_merge in zstd_entropy.h: the same lists, the same order of equal weights), IN REGISTERS since
    // round 5: a level's list is the bitonic merge of the leaves (ascending, positions 0 .. 255: eight consecutive per lane in lanes
    // 0 .. 31, padded with "infinity") and the packages of the level below (DESCENDING from position 511: lane 63 holds the eight
    // lightest) -- nine compare-exchange stages, six of them across lanes (xor 32 .. 1), three inside a lane.  Keys are weight << 1 |
    // is-package: a leaf comes in front of a package of its weight, as in the serial merge (a weight is at most 11 x 2^24: a leaf
    // counts once per level).  The pairs of the merged list are in one lane each, so the next level's packages (package t in lane
    // t / 4) cost no traffic; they move to their place at the top of the next list with eight shuffles.  What a level leaves behind is
    // which of its items are packages: one ballot per register (bit l of mask r = item 8 l + r).  The first version searched every
    // leaf's and every package's place in the other list by bisection in LDS: nine dependent round trips per level, 54 k cycles per
    // table for a lone wavefront against 14 k here (profiles/r05_experiments.md).
    // Element e = lane + 64*j from "which leaves" on: the trip count nj is wave-uniform.
    const uint32_t X0 = 2u * n - 2u;
    const uint32_t nj = (n + 63u) >> 6;
    constexpr uint32_t PM_INF = 0xFFFFFFFFu;
    uint32_t ln[4] = { 0, 0, 0, 0 };
    uint32_t lk[8];   // the leaves as keys
    {
        uint4 a = make_uint4(PM_INF, PM_INF, PM_INF, PM_INF), c = a;
        if (lane < 32) {
            a = *reinterpret_cast<const uint4*>(&K.leaf[8 * lane]);
            c = *reinterpret_cast<const uint4*>(&K.leaf[8 * lane + 4]);
        }
        const uint32_t w8[8] = { a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w };
#pragma unroll
        for (int r = 0; r < 8; ++r) lk[r] = w8[r] == PM_INF ? PM_INF : w8[r] << 1;
    }
    uint32_t m = n >> 1;
    uint32_t pk4[4];  // the packages of the level below: package t = 4 * lane + q
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // level 1: pairs of leaves (m <= 128: lanes 0 .. 31)
        const uint32_t t = 4u * (uint32_t)lane + (uint32_t)q;
        pk4[q] = t < m ? ((((lk[2 * q] >> 1) + (lk[2 * q + 1] >> 1)) << 1) | 1u) : PM_INF;
    }
    wave_lds_sync();   // (the leaves have been read: the masks below reuse the workspace's list area)
    uint64_t* const masks = reinterpret_cast<uint64_t*>(K.merged);   // [level][8]
    uint32_t levSame = maxNbBits;   // levels above this one equal it
    for (uint32_t lev = 2; lev <= maxNbBits; ++lev) {
        const bool last = lev == maxNbBits;
        uint32_t it[8];
        {
            // lane l >= 32, register r: position 8 l + r holds package 511 - (8 l + r) = 8 (63 - l) + (7 - r), which lives in lane
            // 2 (63 - l) + ((7 - r) >> 2), register (7 - r) & 3
            const int src = 2 * (63 - lane);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t p = (uint32_t)__shfl((int)pk4[(7 - r) & 3], (src + ((7 - r) >> 2)) & 63, 64);
                it[r] = lane < 32 ? lk[r] : p;
            }
        }
#pragma unroll
        for (int dl = 32; dl >= 1; dl >>= 1) {
            const bool lower = (lane & dl) == 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t o = (uint32_t)__shfl_xor((int)it[r], dl, 64);
                const uint32_t mn = o < it[r] ? o : it[r], mx = o < it[r] ? it[r] : o;
                it[r] = lower ? mn : mx;
            }
        }
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if ((r & d) == 0) {
                    const uint32_t x = it[r], y = it[r | d];
                    it[r] = x < y ? x : y;
                    it[r | d] = x < y ? y : x;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint64_t bm = __ballot((it[r] & 1u) != 0 && it[r] != PM_INF);
            if (lane == 0) masks[8 * lev + (uint32_t)r] = bm;
        }
        if (last) break;
        const uint32_t size = n + m < X0 ? n + m : X0;
        m = size >> 1;
        bool same = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t t = 4u * (uint32_t)lane + (uint32_t)q;
            const uint32_t v = t < m ? ((((it[2 * q] >> 1) + (it[2 * q + 1] >> 1)) << 1) | 1u) : PM_INF;
            same = same && v == pk4[q];
            pk4[q] = v;
        }
        if (__ballot(!same) == 0) {   // the same packages as one level down: every further level repeats this one
            levSame = lev;
            break;
        }
    }
    wave_lds_sync();
    SUB(7);
    // --- which leaves each level takes: a prefix, known from the number of packages among the level's first X items
    {
        uint32_t X = X0;
        for (uint32_t lev = maxNbBits; lev >= 2 && X; --lev) {
            const uint64_t* mk = masks + 8u * (lev < levSame ? lev : levSame);
            uint32_t pk = 0;
#pragma unroll
            for (uint32_t r = 0; r < 8; ++r) {
                const uint32_t cnt = X > r ? (X - r + 7u) >> 3 : 0u;   // lanes whose item 8 l + r is among the first X
                const uint64_t lm = cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull);
                pk += (uint32_t)__popcll(mk[r] & lm);
            }
            const uint32_t nl = X - pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) ln[j] += ((uint32_t)lane + 64u * j) < nl ? 1u : 0u;
            X = 2u * pk;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ln[j] += ((uint32_t)lane + 64u * j) < X ? 1u : 0u;  // level 1 holds leaves only
    }
    SUB(8);
    for (int i = lane; i < 256; i += WAVE) L.nbBits[i] = 0;
    if (lane < 16) L.rankcnt[lane] = 0;
    wave_lds_sync();
    maxNbBits = (uint32_t)__builtin_amdgcn_readfirstlane((int)ln[0]);  // the lightest symbol has the longest code
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = (uint32_t)lane + 64u * j;
        if ((uint32_t)j < nj && e < n) {
            L.nbBits[K.sym[e]] = (uint8_t)ln[j];
            atomicAdd(&L.rankcnt[ln[j]], 1u);
        }
    }
    SUB(9);
    wave_lds_sync();
    // first code of every length (zstd: longest codes get the smallest values)
    uint32_t val[HUF_ABS_MAX_BITS + 1];
    {
        uint32_t minv = 0;
#pragma unroll
        for (int n = HUF_ABS_MAX_BITS; n > 0; --n) {
            const uint32_t cnt = (uint32_t)n <= maxNbBits ? L.rankcnt[n] : 0u;
            val[n] = minv;
            minv = (uint32_t)n <= maxNbBits ? ((minv + cnt) >> 1) : 0u;
        }
        val[0] = 0;
    }
    // codes in symbol order: within one length, consecutive values by increasing symbol
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (uint32_t)lane + 64u * j;
        const uint32_t nb = L.nbBits[s];
        uint32_t code = 0;
#pragma unroll
        for (int l = 1; l <= HUF_ABS_MAX_BITS; ++l) {
            const uint64_t m = __ballot(nb == (uint32_t)l);
            if (nb == (uint32_t)l) code = val[l] + (uint32_t)__popcll(m & below);
            val[l] += (uint32_t)__popcll(m);
        }
        L.set_code(s, code, nb);
    }
    wave_lds_sync();
    return maxNbBits;
}

// all lanes: the Huffman tree description (huf_write_tree of zstd_entropy.h, byte for byte) with the wavefront on everything
// that is not a chain: the weights' distribution is normalised one symbol per lane, the FSE table is spread and ranked with
// ballots, the two interleaved state chains run on two lanes side by side and only note what they put out, and the bits are
// packed from a wave prefix sum.  The table description (<= 13 symbols, data-dependent widths) stays on lane 0.  Anything
// unusual -- a weight list too short or too uniform to compress, the second normalisation method, a description that does
// not pay -- is left to the serial function.  L.weights[0..maxSym) and L.fw.count[] are filled by the caller.
// (__forceinline__: as a called function this faulted on hardware with ROCm 7.2)
template <class LDS>
__device__ __forceinline__ int huf_write_tree_wave(LDS& L, uint32_t maxSym, uint32_t huffLog, int lane)
{
    FseWeightWksp& W = L.fw;
    const uint32_t wtSize = maxSym;
    const uint32_t cnt = lane <= HUF_ABS_MAX_BITS ? W.count[lane] : 0u;
    const uint64_t nz = __ballot(cnt != 0);
    const uint32_t maxCount = wave_max_u32(cnt);
    bool fast = wtSize > 1 && nz != 0 && maxCount != wtSize && maxCount != 1;
    uint32_t tableLog = 0, maxSV = 0;
    int32_t nm = 0;
    if (fast) {   // FSE_normalizeCount, first method
        maxSV = 63u - (uint32_t)__builtin_clzll(nz);
        tableLog = optimal_table_log(6, wtSize, maxSV, 2);
        const uint32_t total = wtSize;
        const uint64_t scale = 62 - tableLog, step = (1ull << 62) / total, vStep = 1ull << (scale - 20);
        const uint32_t lowThreshold = total >> tableLog;
        int32_t pm = 0;
        if ((uint32_t)lane <= maxSV && cnt) {
            if (cnt <= lowThreshold) nm = 1;
            else {
                int32_t proba = (int32_t)(int16_t)(((uint64_t)cnt * step) >> scale);
                if (proba < 8) {
                    uint32_t rtb = 830000u;   // { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 }
                    rtb = proba == 0 ? 0u : proba == 1 ? 473195u : proba == 2 ? 504333u : proba == 3 ? 520860u
                        : proba == 4 ? 550000u : proba == 5 ? 700000u : proba == 6 ? 750000u : rtb;
                    const uint64_t restToBeat = vStep * rtb;
                    proba += (((uint64_t)cnt * step) - ((uint64_t)proba << scale)) > restToBeat ? 1 : 0;
                }
                nm = proba;
                pm = proba;
            }
        }
        const int32_t still = (int32_t)(1u << tableLog) - (int32_t)wave_sum_u32((uint32_t)nm);
        const uint32_t maxP = wave_max_u32((uint32_t)pm);
        const uint64_t atMax = __ballot(pm > 0 && (uint32_t)pm == maxP);
        const int largest = atMax ? (int)__builtin_ctzll(atMax) : 0;
        const int32_t normLargest = __shfl(nm, largest, 64);
        if (-still >= (normLargest >> 1)) fast = false;   // second method: serial
        else if (lane == largest) nm += still;
    }
    if (fast) {
        if (lane < 16) W.norm[lane] = (int16_t)nm;
        for (int i = lane; i < 52; i += WAVE) W.bitbuf[i] = 0;
        wave_lds_sync();
        if (lane == 0) W.verdict = fse_write_ncount(L.tree + 1, 133, W.norm, maxSV, tableLog);
        // FSE_buildCTable: cell (i * step) & mask holds the symbol of occurrence i; a symbol's states in cell order
        const uint32_t tableSize = 1u << tableLog, tableMask = tableSize - 1u;
        const uint32_t cum = wave_incl_scan_u32((uint32_t)nm) - (uint32_t)nm;   // occurrences in front of symbol `lane`
        if (lane < 16) W.cumul[lane] = cum;
        wave_lds_sync();
        const int hsz = W.verdict;
        if (hsz < 0) fast = false;
        else {
            if ((uint32_t)lane < tableSize) {
                uint32_t sy = 0;
                for (uint32_t u = 1; u <= maxSV; ++u) sy += W.cumul[u] <= (uint32_t)lane ? 1u : 0u;
                W.tableSymbol[((uint32_t)lane * ((tableSize >> 1) + (tableSize >> 3) + 3u)) & tableMask] = (uint8_t)sy;
            }
            if ((uint32_t)lane <= maxSV) {
                uint32_t dnb, dfs = 0;
                if (nm == 0) dnb = ((tableLog + 1) << 16) - tableSize;
                else if (nm == 1) { dnb = (tableLog << 16) - tableSize; dfs = cum - 1u; }
                else {
                    const uint32_t maxBitsOut = tableLog - (uint32_t)hb32((uint32_t)nm - 1u);
                    dnb = (maxBitsOut << 16) - ((uint32_t)nm << maxBitsOut);
                    dfs = cum - (uint32_t)nm;
                }
                W.deltaNbBits[lane] = dnb;
                W.deltaFindState[lane] = (int32_t)dfs;
            }
            wave_lds_sync();
            {
                const bool cell = (uint32_t)lane < tableSize;
                const uint32_t sy = cell ? W.tableSymbol[lane] : 0xFFu;
                uint32_t rank = 0;
                const uint64_t below = (1ull << lane) - 1ull;
                for (uint32_t u = 0; u <= maxSV; ++u) {
                    const uint64_t same = __ballot(sy == u);
                    if (sy == u) rank = (uint32_t)__popcll(same & below);
                }
                if (cell) W.stateTable[W.cumul[sy] + rank] = (uint16_t)(tableSize + (uint32_t)lane);
            }
            wave_lds_sync();
            // FSE_compress_usingCTable: the chain that starts with the last weight takes the even steps, the other the odd
            const uint32_t ne = wtSize - 2u;
            if (lane < 2) {
                const uint32_t s0 = L.weights[wtSize - 1u - (uint32_t)lane];
                const uint32_t d0 = W.deltaNbBits[s0], nb0 = (d0 + (1u << 15)) >> 16;
                uint32_t st = W.stateTable[(int32_t)(((nb0 << 16) - d0) >> nb0) + W.deltaFindState[s0]];
                for (uint32_t k = (uint32_t)lane; k < ne; k += 2) {
                    const uint32_t sy = L.weights[wtSize - 3u - k];
                    const uint32_t nbo = (st + W.deltaNbBits[sy]) >> 16;
                    W.emit[k] = (uint16_t)((st & ((1u << nbo) - 1u)) | (nbo << 8));
                    st = W.stateTable[(int32_t)(st >> nbo) + W.deltaFindState[sy]];
                }
                // the final states: "state 2" first -- the odd chain for an odd number of weights, the even chain otherwise
                const uint32_t slot = ne + (((wtSize & 1u) != 0) == (lane == 1) ? 0u : 1u);
                W.emit[slot] = (uint16_t)((st & tableMask) | (tableLog << 8));
                if (lane == 0) W.emit[ne + 2u] = (uint16_t)(1u | (1u << 8));   // end mark
            }
            wave_lds_sync();
            uint32_t acc = 0, tb = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t k = 4u * (uint32_t)lane + (uint32_t)q;
                const uint32_t e = k < ne + 3u ? W.emit[k] : 0u;
                acc |= (e & 0xFFu) << tb;
                tb += e >> 8;
            }
            const uint32_t incl = wave_incl_scan_u32(tb);
            const uint32_t off = incl - tb, sh = off & 31u;
            if (tb) {
                atomicOr(&W.bitbuf[off >> 5], acc << sh);
                if (sh + tb > 32u) atomicOr(&W.bitbuf[(off >> 5) + 1u], acc >> (32u - sh));
            }
            const uint32_t nbytes = ((uint32_t)__builtin_amdgcn_readlane((int)incl, 63) + 7u) >> 3;
            wave_lds_sync();
            const uint32_t hSize = (uint32_t)hsz + nbytes;
            if (hSize >= 133u || !(hSize > 1u && hSize < maxSym / 2u)) fast = false;
            else {
                for (uint32_t i = lane; i < nbytes; i += WAVE) L.tree[1u + (uint32_t)hsz + i] = (uint8_t)(W.bitbuf[i >> 2] >> (8u * (i & 3u)));
                if (lane == 0) L.tree[0] = (uint8_t)hSize;
                wave_lds_sync();
                return (int)hSize + 1;
            }
        }
    }
    // the serial statement decides everything else (it starts over: the histogram of the weights is still there)
    wave_lds_sync();
    if (lane == 0) W.verdict = huf_write_tree(L.tree, 134, L.nbBits, maxSym, huffLog, L.weights, &W, true);
    wave_lds_sync();
    return W.verdict;
}

// all lanes: choose the coding mode of a region from its histogram and, for Huffman, build the table
// S: bytes of the region; Sh: bytes its histogram counts (S, or the size of the sample: region_histogram)
template <class LDS>
__device__ __forceinline__ void region_plan(LDS& L, uint32_t S, uint32_t Sh, uint32_t nblk, int lane, unsigned long long* tsub = nullptr, unsigned long long* tl = nullptr)
{
    uint32_t mx = 0, msym = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (uint32_t)lane + 64u * j;
        const uint32_t h = L.hist[s];
        mx = h > mx ? h : mx;
        msym = h ? s : msym;
    }
    const uint32_t maxCount = wave_max_u32(mx);
    const uint32_t maxSym = wave_max_u32(msym);
    if (lane == 0) {
        L.treeSize = 0;
        L.mode = maxCount == Sh ? 1u : 0u;     // (a sample is only used when it shows nearly all byte values: never here)
    }
    wave_lds_sync();
    if (maxCount == Sh) return;
    if (S <= 63) return;                       // libzstd stores such literals raw (minLitSize)
    if (maxCount <= (Sh >> 7) + 4) return;     // libzstd's "probably not compressible" heuristic
    // the sort keys hold a count in 24 bits: scale the histogram of a larger region down (still a valid code)
    uint32_t shift = 0;
    while ((Sh >> shift) >= (1u << 24)) shift++;
    if (shift) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t s = (uint32_t)lane + 64u * j;
            const uint32_t h = L.hist[s];
            if (h) { const uint32_t cc = h >> shift; L.hist[s] = cc ? cc : 1u; }
        }
        wave_lds_sync();
    }
    const uint32_t logSrc = S < BLOCK_MAX ? S : BLOCK_MAX;
    uint32_t huffLog = optimal_table_log(HUF_MAX_BITS, logSrc, maxSym, 1);
    SUB(6);
    huffLog = huf_build_wave(L, maxSym, huffLog, lane, tsub, tl);
    // weights (all but the last symbol's) and their histogram, in parallel
    if (lane < 16) L.fw.count[lane] = 0;
    wave_lds_sync();
    uint64_t mybits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (uint32_t)lane + 64u * j;
        const uint32_t nb = L.nbBits[s];
        if (s < maxSym) {
            const uint32_t wt = nb ? huffLog + 1 - nb : 0u;
            L.weights[s] = (uint8_t)wt;
            atomicAdd(&L.fw.count[wt], 1u);
        }
        mybits += (uint64_t)L.hist[s] * nb;
    }
    // (a lane's share in two limbs of 20 bits, each summed over the wavefront in 32 bits)
    const uint64_t bits = ((uint64_t)wave_sum_u32((uint32_t)(mybits >> 20)) << 20) + wave_sum_u32((uint32_t)mybits & 0xFFFFFu);
    wave_lds_sync();
    SUB(10);
    const int ts = huf_write_tree_wave(L, maxSym, huffLog, lane);
    if (lane == 0) {
        bool ok = ts >= 0;
        if (ok) {
            const uint64_t counted = (bits << shift) >> 3;   // bytes the counted part of the region would take
            const uint64_t est = (Sh == S ? counted : counted * S / Sh) + (uint64_t)ts + 14ull * nblk;
            const uint64_t minGain = (S >> 6) + 2;  // ZSTD_minGain
            ok = est + minGain < S;
        }
        if (ok) {
            L.treeSize = ts;
            L.huffLog = huffLog;
            L.mode = 2;
        }
    }
    wave_lds_sync();
    SUB(11);
}

// ---- run sequences for the control-byte region (and for frames too small to have one) ----------------------------------
// Control bytes are mostly 0x00; a Huffman code cannot spend less than one bit on each of them, libzstd's
// LZ stage does.  The device equivalent keeps everything data-parallel: every run of >= RMIN zero bytes
// becomes one zstd sequence "copy run-1 bytes from offset 1" (offset 1 = repeat offset 1 of a fresh frame,
// so the offset costs no bits: OF table in RLE mode, code 0), its first zero stays a literal, and the
// literals (everything outside the run tails) are Huffman coded as before.  Which bytes are run tails is a
// morphological open of the zero mask, computed with shifts on a 64-bit window per lane; positions come from
// wave prefix sums; the FSE state chains of the sequences section run on all lanes too (encode_zero_run_sequences).
#ifndef VBZ_DATA_BLOCKS
#define VBZ_DATA_BLOCKS 15
#endif
constexpr uint32_t DATA_BLOCKS = VBZ_DATA_BLOCKS;  // blocks of the data-byte region when the control bytes take one block
// (RMIN, the shortest run that becomes a match: vbz_kernels.h -- the svb encoder's tokeniser uses it too)
static_assert(RMIN >= 8 && RMIN <= 24, "the tokeniser handles at most two run ends per 16 positions and a 64-bit window");
constexpr uint32_t TOK_PAYLOAD = 60 * 16;

__device__ __forceinline__ uint64_t erode_right(uint64_t w, uint32_t r)  // bit j = AND of bits j .. j+r-1
{
    uint32_t span = 1;
    while (span * 2 <= r) { w &= w >> span; span *= 2; }
    if (span < r) w &= w >> (r - span);
    return w;
}

__device__ __forceinline__ uint64_t dilate_left(uint64_t w, uint32_t r)  // bit j = OR of bits j-r+1 .. j
{
    uint32_t span = 1;
    while (span * 2 <= r) { w |= w << span; span *= 2; }
    if (span < r) w |= w << (r - span);
    return w;
}

// all lanes.  Compacts the literals of k[0..K) in place (k[0..Lit)) and writes one record per qualifying
// run, in order: rec[j] = (end position of the run, literals before that position).
// PERIOD: the mask "byte p equals byte p - D" was computed beforehand (period_mask) and is read from `mask16` (one 16-bit
// word per 16 positions); a run of RPER or more such positions is one match, none of its bytes stays a literal.
constexpr uint32_t RPER = 16;
// COUNT_ONLY: nothing is written or moved; nrec is all that is wanted.
template <bool PERIOD, bool COUNT_ONLY = false, bool TSUB = false>
__device__ void tokenise_runs(uint8_t* k, uint32_t K, uint2* rec, uint32_t& Lit, uint32_t& nrec, const uint16_t* mask16, int lane,
                              unsigned long long* tsub_ = nullptr, unsigned long long* tl = nullptr)
{
    unsigned long long* const tsub = TSUB ? tsub_ : nullptr;   // (sub-phase timers: the timed build of the planning launch only)
    uint32_t lit_total = 0, rec_total = 0;
    uint32_t carry60 = 0, carry61 = 0;  // masks of the 32 positions in front of the payload
    uint32_t carryw = 0;                // the dword that ends in front of the payload (its top byte precedes position pb)
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t pb = 0; pb < K; pb += TOK_PAYLOAD) {
        const int64_t lo = (int64_t)pb + 16 * ((int64_t)lane - 2);
        uint32_t w[4] = { 0, 0, 0, 0 };
        uint32_t zm = 0, valid = 0;
        if (lane >= 2 && lo < (int64_t)K) {
            uint4 v;
            __builtin_memcpy(&v, k + lo, 16);  // the slot has 16+ bytes of slack behind the stream
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            const uint32_t nvalid = (K - (uint32_t)lo) >= 16 ? 16u : (K - (uint32_t)lo);
            valid = nvalid >= 16 ? 0xFFFFu : ((1u << nvalid) - 1u);
        }
        if (PERIOD) {
            zm = (lane >= 2 && lo < (int64_t)K) ? (uint32_t)mask16[(uint32_t)lo >> 4] & valid : 0u;
        } else {
            // bit i of the mask: byte i equals the byte in front of it (a run of r equal bytes is r - 1 ones)
            uint32_t pw = wave_prev_lane_u32(w[3]);   // (DPP: a shuffle through the LDS crossbar is a round trip on this loop's dependent chain)
            if (lane == 2) pw = carryw;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = w[q] ^ __builtin_amdgcn_alignbyte(w[q], q ? w[q - 1] : pw, 3);  // zero byte <=> equal neighbours
                // "this byte is zero" for the four bytes at once (bit 7 of each), then the four bits side by side: the multiplication
                // moves bit 8 j to bit 21 + j (no two terms of the product meet)
                const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
                zm |= ((((z >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * q);
            }
            zm &= valid;
            if (lane == 2 && pb == 0) zm &= ~1u;  // nothing in front of the first byte
        }
        if (!PERIOD) {
            if (lane == 0) zm = carry60;
            if (lane == 1) zm = carry61;
        } else if (lane < 2) {  // the 32 positions in front of the payload: their masks are in memory too
            const int64_t lb = (int64_t)pb + 16 * ((int64_t)lane - 2);
            zm = lb >= 0 ? (uint32_t)mask16[(uint32_t)lb >> 4] : 0u;
        }
        SUB(3);   // (timed build: the trip's load and the equal-neighbour mask)
        // the masks of the two lanes below and above (DPP wave shifts: the lanes at the ends get 0)
        const uint32_t m1 = wave_prev_lane_u32(zm), m2 = wave_prev_lane_u32(m1);
        const uint32_t p1 = wave_next_lane_u32(zm), p2 = wave_next_lane_u32(p1);
        // window bit j <-> position lo - 24 + j
        const uint64_t W = (uint64_t)((m2 >> 8) & 0xFF) | ((uint64_t)m1 << 8) | ((uint64_t)zm << 24) | ((uint64_t)p1 << 40) |
                           ((uint64_t)(p2 & 0xFF) << 56);
        // d = 1: runs of >= RMIN equal bytes, without their first byte; period: runs of >= RPER matching positions
        const uint64_t RM = PERIOD ? dilate_left(erode_right(W, RPER), RPER) : dilate_left(erode_right(W, RMIN - 1), RMIN - 1);
        const uint64_t END = RM & ~(RM >> 1);                                  // last position of such a run
        const bool payload = lane >= 2 && lane < 62;
        const uint32_t r16 = payload ? (uint32_t)(RM >> 24) & 0xFFFFu : 0u;
        const uint32_t end16 = payload ? (uint32_t)(END >> 24) & 0xFFFFu & valid : 0u;
        const uint32_t kept16 = payload ? (~r16 & valid) : 0u;
        const uint32_t c = (uint32_t)__popc(kept16);
        const uint32_t incl = wave_incl_scan_u32(c);
        const uint32_t excl = incl - c;
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // with RMIN <= 8 two qualifying runs can end inside one lane's 16 positions (never three)
        const uint32_t second = end16 & (end16 - 1);
        const uint64_t endmask1 = __ballot(end16 != 0), endmask2 = __ballot(second != 0);
        SUB(5);   // (the window, the runs, the scan and the ballots)
        if (!COUNT_ONLY && end16) {
            const uint32_t idx = rec_total + (uint32_t)__popcll(endmask1 & below) + (uint32_t)__popcll(endmask2 & below);
            const uint32_t i = (uint32_t)__ffs((int)end16) - 1u;
            rec[idx] = make_uint2((uint32_t)lo + i + 1u, lit_total + excl + (uint32_t)__popc(kept16 & ((2u << i) - 1u)));
            if (second) {
                const uint32_t i2 = (uint32_t)__ffs((int)second) - 1u;
                rec[idx + 1] = make_uint2((uint32_t)lo + i2 + 1u, lit_total + excl + (uint32_t)__popc(kept16 & ((2u << i2) - 1u)));
            }
        }
        if (!COUNT_ONLY) {
            uint32_t off = lit_total + excl;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((kept16 >> i) & 1u) k[off++] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        }
        SUB(2);   // (the records and the literals' stores)
        rec_total += (uint32_t)__popcll(endmask1) + (uint32_t)__popcll(endmask2);
        lit_total += tot;
        carry60 = (uint32_t)__builtin_amdgcn_readlane((int)zm, 60);
        carry61 = (uint32_t)__builtin_amdgcn_readlane((int)zm, 61);
        carryw = (uint32_t)__builtin_amdgcn_readlane((int)w[3], 61);
    }
    Lit = lit_total;
    nrec = rec_total;
}

// ---- long repeats at ONE distance ---------------------------------------------------------------------------------------
// Signal that repeats a template (the reference's own perf generator cycles a 15 643-sample read: vbz/perf/
// test_data_generator.h:61-67) makes the data bytes of the svb stream periodic; libzstd's match finder -- at every level,
// vbz/vbz.cpp:194-207 hands the caller's level through -- turns every period after the first into one long match.  The
// device equivalent uses ONE distance D per read: the svb encoder proposes it (svb_kernels.hip: PeriodProbe -- it has every
// data byte in LDS once and looks the first dword of every 16-byte chunk up in a table of sixteen probes, which costs a read
// without a period next to nothing), period_holds() checks the proposal, every position is then marked "equals the byte D
// before" (period_mask) and runs of RPER or more marked positions become sequences with the explicit offset D (OF table in
// RLE mode with code floor(log2(D + 3))).
constexpr uint32_t PERIOD_MIN_D = 64;   // (short distances: the d = 1 tokeniser and the Huffman code do better)
// ... with ONE exception (round 6): distance 1, "a run".  Data bytes that are one long run -- iota of any integer type is one, the
// reference's own perf and plugin-test input (vbz/perf/test_data_generator.h:12-23, vbz_hdf_plugin_test.cpp:15-48): every delta is 1 --
// cost a Huffman code one bit per byte, and libzstd a few bytes per block: 1 MB of int16 iota came out as 64 696 bytes against 47
// (profiles/r06_ratio_sweep.md).  A read whose probes cannot be placed because the bytes under them are all alike is proposed the
// distance 1 (svb_kernels.hip, period_probe_kernel); it goes the way of every other distance: period_holds asks eight places spread
// over the data bytes, the matcher marks "equals the byte in front of it" and runs of RPER and more become matches with the explicit
// offset 1.
constexpr uint32_t RUN_D = 1;

// all lanes: eight places spread over the region are asked (one lane each); most of them must repeat what stands D bytes
// before -- a template repeated with a few changed samples still qualifies, a chance hit of the probe does not.
__device__ bool period_holds(const uint8_t* in, uint32_t S, uint32_t D, int lane)
{
    if ((D < PERIOD_MIN_D && D != RUN_D) || S < 8192 || (uint64_t)D + 32u > S) return false;
    bool agree = false;
    if (lane < 8) {
        const uint32_t t = D + 8u + (uint32_t)(((uint64_t)(S - D - 16u) * (uint32_t)lane) / 8u);  // spread over what has a predecessor
        if (t + 8 <= S) {
            uint32_t x[2], y[2];
            __builtin_memcpy(x, in + t, 8);
            __builtin_memcpy(y, in + t - D, 8);
            agree = x[0] == y[0] && x[1] == y[1];
        }
    }
    return __popcll(__ballot(agree)) >= 5;
}

// all lanes.  A read that was proposed the distance 1 (its probes could not be placed: runs) may be runs that REPEAT -- iota of int8 is 255
// equal data bytes and a two-byte value, over and over (libzstd matches it at distance 257 and writes 2.9 KB where runs alone cost 20) --:
// the places where a byte differs from the one in front of it are looked up in the 2 KB behind data byte 64, and the distance from the
// first such place to the next one that does not belong to the same cluster is proposed instead, if it holds (period_holds); else the run.
__device__ uint32_t run_period(const uint8_t* d, uint32_t S, int lane)
{
    constexpr uint32_t FROM = 64, SPAN = 32u * WAVE;
    if (S < FROM + SPAN + 16u) return RUN_D;
    const uint8_t* p = d + FROM + 32u * (uint32_t)lane;
    uint32_t m = 0;
    {
        uint32_t w[9];
        __builtin_memcpy(&w[1], p, 32);
        w[0] = (uint32_t)p[-1] << 24;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t x = w[q + 1] ^ __builtin_amdgcn_alignbyte(w[q + 1], w[q], 3);   // zero byte <=> equals the byte in front of it
            const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
            m |= (((((z >> 7) * 0x00204081u) >> 21) & 0xFu) ^ 0xFu) << (4 * q);
        }
    }
    uint64_t who = __ballot(m != 0);
    uint32_t first = 0xFFFFFFFFu, cand = 0;
    while (who != 0 && cand == 0) {
        const int l = __ffsll((long long)who) - 1;
        uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)m, l);
        while (mm != 0 && cand == 0) {
            const uint32_t at = 32u * (uint32_t)l + (uint32_t)__ffs((int)mm) - 1u;
            if (first == 0xFFFFFFFFu) first = at;
            else if (at >= first + PERIOD_MIN_D) cand = at - first;
            mm &= mm - 1u;
        }
        who &= who - 1ull;
    }
    return (cand != 0 && period_holds(d, S, cand, lane)) ? cand : RUN_D;
}

// Where the long-repeat coder keeps its workspace -- one mask bit per data byte and one 8-byte record per match.  The mask
// goes to the spare room of the library's scratch slot behind the stream (below the control-byte region's run records) if
// it fits there, the records to the top of the read's DESTINATION slot, which the caller sizes for the worst case
// (vbz_max_compressed_size: 4 bytes per value and more) while a frame is never longer than its stream plus a little; the
// frame may then use the slot up to `cap` only.  Room for every match the data could possibly hold (one per RPER + 1 bytes)
// is not asked for: real repeats are few and long, so the records get what room there is (rec_cap) and the matches are
// counted before anything is moved.  false: no room at all (a stream near its worst case): no matcher for this read.
struct DeepLayout { uint8_t* recs; uint16_t* mask; uint16_t* maskK; uint32_t cap, rec_cap, nch, nchK; };
__device__ __forceinline__ bool deep_layout(uint32_t N, uint32_t K, const uint8_t* in, uint32_t slot, uint8_t* out, uint32_t dst_cap, uint32_t hdr,
                                            DeepLayout& d)
{
    const uint32_t SD = N - K;
    d.nch = (SD + (BLOCK_MAX - 16u) - 1u) / (BLOCK_MAX - 16u);
    d.nchK = (K + (BLOCK_MAX - 16u) - 1u) / (BLOCK_MAX - 16u);
    // (one mask bit per data byte and, behind them, one per control byte: the control bytes of a read that cycles repeat too)
    const uint32_t maskD_bytes = 2u * ((SD + 15u) >> 4) + 16u;
    const uint32_t mask_bytes = maskD_bytes + 2u * ((K + 15u) >> 4) + 16u;
    const uint64_t frame_worst = (uint64_t)hdr + N + (N >> 6) + 1024u;   // (raw blocks are the worst a region can do)
    if (frame_worst + 64 > dst_cap) return false;
    uint64_t room = dst_cap - frame_worst - 48;                           // of the destination slot, above the frame
    const uintptr_t top = (uintptr_t)(out + dst_cap) & ~(uintptr_t)15;
    const uint64_t keyrecs = 8ull * (K / RMIN + 4u + 2u * d.nchK) + 8;    // the control-byte region's records (of every chunk of it), at the top of the scratch slot
    uintptr_t mask;
    if ((uint64_t)N + 32 + mask_bytes + 16 + keyrecs <= slot) {
        mask = ((uintptr_t)(in + slot) - keyrecs - mask_bytes) & ~(uintptr_t)15;
    } else {
        if (room < mask_bytes + 16u) return false;
        room -= mask_bytes + 16u;
        mask = top - mask_bytes - 16u;
    }
    const uint32_t worst = N / (RPER + 1u) + 2u * (d.nch + d.nchK) + 4u;  // more matches than this the stream cannot hold
    const uint64_t fit = room / 8u;
    d.rec_cap = fit < worst ? (uint32_t)fit : worst;
    if (d.rec_cap < 16u) return false;
    const uintptr_t recs = ((mask < top && mask >= (uintptr_t)out ? mask : top) - 8ull * d.rec_cap) & ~(uintptr_t)15;
    d.recs = reinterpret_cast<uint8_t*>(recs);
    d.mask = reinterpret_cast<uint16_t*>(mask);
    d.maskK = reinterpret_cast<uint16_t*>(mask + maskD_bytes);
    d.cap = (uint32_t)(recs - (uintptr_t)out);
    return true;
}

// all lanes.  The control bytes of a read whose data bytes repeat at distance D: how many VALUES make D data bytes?  Counted from control
// byte `from` on (the read's very first value is a delta from nothing and may be longer than its later copies): the lengths the control
// bytes announce are added up until they reach D exactly.  Returns that number of values, 0 if D falls inside a value or behind the region.
__device__ uint32_t control_period_values(const uint8_t* keys, uint32_t K, uint32_t from, uint32_t D, int lane)
{
    uint32_t done = 0;   // data bytes of the control bytes in front of this round
    for (uint32_t k0 = from; k0 + 4u <= K; k0 += 4u * WAVE) {
        const uint32_t at = k0 + 4u * (uint32_t)lane;
        uint32_t w = 0;
        if (at + 4u <= K) __builtin_memcpy(&w, keys + at, 4);
        // sixteen 2-bit codes: length = code + 1
        const uint32_t lo = w & 0x33333333u, hi = (w >> 2) & 0x33333333u;
        uint32_t t = lo + hi;                                  // nibbles: two codes each
        t = (t & 0x0F0F0F0Fu) + ((t >> 4) & 0x0F0F0F0Fu);
        const uint32_t len = at + 4u <= K ? ((t * 0x01010101u) >> 24) + 16u : 0u;
        const uint32_t incl = wave_incl_scan_u32(len);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (done + total >= D) {
            // the lane whose sixteen values contain the D-th byte's end
            const uint32_t before = done + incl - len;
            const bool mine = len != 0 && before < D && before + len >= D;
            uint32_t vals = 0;
            if (mine) {
                uint32_t acc = before;
                for (uint32_t i = 0; i < 16u && acc < D; ++i) {
                    acc += ((w >> (2u * i)) & 3u) + 1u;
                    vals = acc == D ? (at - from) * 4u + i + 1u : 0u;
                }
            }
            const uint64_t who = __ballot(mine);
            if (!who) return 0;
            return (uint32_t)__shfl((int)vals, __ffsll((long long)who) - 1, 64);
        }
        done += total;
    }
    return 0;
}

// all lanes: mask16[p >> 4] bit (p & 15) = byte p equals byte p - D (0 for p < D)
__device__ void period_mask(const uint8_t* in, uint32_t S, uint32_t D, uint16_t* mask16, int lane)
{
    for (uint32_t q0 = 0; q0 < S; q0 += 16 * WAVE) {
        const uint32_t q = q0 + 16u * (uint32_t)lane;
        if (q >= S) continue;
        uint32_t m = 0;
        if (q >= D) {
            uint4 a, c;
            __builtin_memcpy(&a, in + q, 16);       // (16+ bytes of slack behind the stream)
            __builtin_memcpy(&c, in + q - D, 16);
            const uint32_t x[4] = { a.x ^ c.x, a.y ^ c.y, a.z ^ c.z, a.w ^ c.w };
#pragma unroll
            for (int i = 0; i < 16; ++i) m |= (((x[i >> 2] >> (8 * (i & 3))) & 0xFF) == 0 ? 1u : 0u) << i;
        } else if (q + 16 > D) {
            for (uint32_t i = D - q; i < 16; ++i) m |= (in[q + i] == in[q + i - D] ? 1u : 0u) << i;
        }
        const uint32_t nvalid = (S - q) >= 16 ? 16u : (S - q);
        mask16[q >> 4] = (uint16_t)(m & (nvalid >= 16 ? 0xFFFFu : ((1u << nvalid) - 1u)));
    }
}

// Decoder checkpoints.  The LL / ML state chain of a sequences section is serial for a decoder that starts at
// the top of the bit stream; the encoder knows every intermediate state, so it publishes one checkpoint per
// CP spacing sequences -- (unread bits, LL state, ML state) before sequence k * spacing -- in a zstd *skippable
// frame* behind the frame (RFC 8878 3.1.2: decoders skip it; libzstd's ZSTD_decompress and
// ZSTD_getFrameContentSize are unaffected).  zstd_decode.hip walks the segments in parallel, one lane each, and
// accepts the result only if every segment ends exactly in the next checkpoint, so a wrong or missing trailer
// costs speed, never correctness.  Layout: magic 0x184D2A5B, u32 size, { u16 spacing, u16 count,
// count x u32 (unread bits | LL state << 20 | ML state << 26), u32 total trailer bytes }.
constexpr uint32_t CP_MAGIC = 0x184D2A5Bu;
constexpr uint32_t CP_MIN_SPACING = 32;

// all lanes.  Sequences section (RFC 8878 3.1.1.3.2) for the records of tokenise_runs: LL and ML with the predefined
// distributions, OF in RLE mode; same bit order as libzstd's ZSTD_encodeSequences (last sequence first).
//
// The two FSE state chains (match length, literal length) look serial -- a step's output is the low bits of the state the
// previous step left -- but a step forgets almost everything: a symbol with c cells in the table (c <= 4 for both predefined
// distributions) leaves one of c states, chosen by the top bits of the state before it, and the four states 64, 80, 96, 112
// between them reach every cell of every symbol.  So a lane that is to encode sequences [a, b) walks the SEQ_WARM sequences
// in front of a from those four states at once; when the four walks have met (a symbol with one cell joins them at once,
// one with two cells every other time) the state before a is known without the history, and all lanes encode their
// SEQ_PER_LANE sequences side by side.  A lane whose walks have not met waits for its neighbour's final state (one more pass
// of the lanes concerned; the first lane always knows its state), so the result is the serial chain's, bit for bit, whatever
// the data.  Per round of 64 x SEQ_PER_LANE sequences: codes and extra bits of every sequence (one per lane and chunk), the
// chains, then chunk by chunk every lane assembles the <= 44 + 17 bits of its sequence, a wave prefix sum places them, the
// chunk leaves as dwords.  Returns the bytes written.
// of_dist: 0 = every sequence copies from repeat offset 1 (runs; OF code 0, checkpoints recorded); otherwise every sequence
// carries the explicit distance of_dist (OF code floor(log2(of_dist + 3)), that many extra bits; no checkpoints).
constexpr int SEQ_PER_LANE = 4, SEQ_ROUND = WAVE * SEQ_PER_LANE, SEQ_WARM = 8;

struct SeqStep { uint32_t st; uint32_t piece; };   // piece: bits | count << 6

__device__ __forceinline__ SeqStep seq_fse_init(uint32_t dnb, int32_t dfs, const uint16_t* stab)  // FSE_initCState2
{
    const uint32_t nbo = (dnb + (1u << 15)) >> 16;
    return { stab[(int32_t)(((nbo << 16) - dnb) >> nbo) + dfs], 0u };
}

__device__ __forceinline__ SeqStep seq_fse_step(uint32_t st, uint32_t dnb, int32_t dfs, const uint16_t* stab)  // FSE_encodeSymbol
{
    const uint32_t nb = (st + dnb) >> 16;
    return { stab[(int32_t)(st >> nb) + dfs], (st & ((1u << nb) - 1u)) | (nb << 6) };
}

template <class LDS>
__device__ __forceinline__ uint32_t encode_zero_run_sequences(LDS& L, uint8_t* dst, const uint2* rec, uint32_t nseq, int lane, uint32_t of_dist = 0)
{
    const uint32_t of_code = of_dist ? (uint32_t)hb32(of_dist + 3u) : 0u;
    const uint32_t of_extra = of_dist ? (of_dist + 3u) - (1u << of_code) : 0u;
    uint32_t hdr = 0;
    if (lane == 0) {
        uint8_t* op = dst;
        if (nseq < 128) { *op++ = (uint8_t)nseq; }
        else if (nseq < 0x7F00) { *op++ = (uint8_t)((nseq >> 8) + 128); *op++ = (uint8_t)nseq; }
        else { *op++ = 255; *op++ = (uint8_t)(nseq - 0x7F00); *op++ = (uint8_t)((nseq - 0x7F00) >> 8); }
        *op++ = 0x10;  // LL predefined | OF RLE | ML predefined
        *op++ = (uint8_t)of_code;
        hdr = (uint32_t)(op - dst);
    }
    hdr = (uint32_t)__builtin_amdgcn_readlane((int)hdr, 0);
    uint8_t* out = dst + hdr;
    // LDS scratch inside the (idle) workspace of the stream packer
    uint32_t* bits = L.obuf;                 // 128 words: 64 sequences x (44 + 17) bits + carry
    uint16_t* code = L.seqcode;              // [SEQ_WARM + position in the round]: match-length code | literal-length code << 8
    uint32_t* piece = L.seqpiece;            // [position in the round]: per chain bits | count << 6 | state after << 9; LL chain << 15
    for (int i = lane; i < 128; i += WAVE) bits[i] = 0;
    if (lane < SEQ_WARM) code[lane] = 0;
    uint32_t carryM = 0, carryL = 0;         // states after the last sequence of the previous round (wave-uniform)
    uint32_t base_bits = 0, flushed = 0;
    uint32_t spacing = CP_MIN_SPACING;  // at most 63 checkpoints + the start = 64 decoder lanes
    while ((nseq + spacing - 1) / spacing > 64) spacing *= 2;
    if (lane == 0 && of_dist == 0) {
        L.cpSpacing = spacing;
        L.cpCount = (nseq - 1) / spacing;
    }
    for (uint32_t r0 = 0; r0 < nseq; r0 += SEQ_ROUND) {
        const uint32_t rn = (nseq - r0) < (uint32_t)SEQ_ROUND ? (nseq - r0) : (uint32_t)SEQ_ROUND;  // sequences of this round
        // --- codes and extra bits: chunk c, lane l is sequence r0 + 64 c + l of the section (= record nseq - 1 - that)
        uint32_t lexv[SEQ_PER_LANE], mexv[SEQ_PER_LANE];   // extra bits | count << 16
        wave_lds_sync();
        if (r0) {   // the walks start among the last sequences of the round before
            const uint32_t keep = lane < SEQ_WARM ? code[SEQ_ROUND + lane] : 0u;
            wave_lds_sync();
            if (lane < SEQ_WARM) code[lane] = (uint16_t)keep;
        }
#pragma unroll
        for (int c = 0; c < SEQ_PER_LANE; ++c) {
            const uint32_t i = 64u * c + (uint32_t)lane;
            uint32_t lex = 0, lnb = 0, mex = 0, mnb = 0, lc = 0, mc = 0;
            if (i < rn) {
                const uint32_t n = nseq - 1 - (r0 + i);
                const uint2 cur = rec[n];
                const uint2 prev = n ? rec[n - 1] : make_uint2(0u, 0u);
                const uint32_t ll = cur.y - prev.y;
                const uint32_t ml = (cur.x - prev.x) - ll;
                seq_ll_code(ll, &lc, &lex, &lnb);
                seq_ml_code(ml, &mc, &mex, &mnb);
            }
            lexv[c] = lex | (lnb << 16);
            mexv[c] = mex | (mnb << 16);
            code[SEQ_WARM + i] = (uint16_t)(mc | (lc << 8));
        }
        wave_lds_sync();
        // --- the chains: this lane encodes positions [a, b) of the round
        {
            const uint32_t a = (uint32_t)SEQ_PER_LANE * (uint32_t)lane;
            const uint32_t b = a + SEQ_PER_LANE < rn ? a + SEQ_PER_LANE : rn;
            uint32_t cM[4], cL[4];
            const bool first = r0 == 0 && a == 0;           // the very first sequence only selects the states
            uint32_t w = 0;                                  // walk over positions [a - w, a)
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { cM[q] = carryM; cL[q] = carryL; }
            } else {
                w = SEQ_WARM;
                if (r0 == 0 && a <= (uint32_t)SEQ_WARM) w = a;   // from the first sequence of the section: exact
#pragma unroll
                for (int q = 0; q < 4; ++q) { cM[q] = 64u + 16u * q; cL[q] = 64u + 16u * q; }
            }
            for (uint32_t k = SEQ_WARM; k > 0; --k) {       // position a - k
                if (k <= w) {
                    const uint32_t cc = code[SEQ_WARM + a - k], mc = cc & 0xFFu, lc = cc >> 8;
                    const uint32_t dm = L.seq.ml_dnb[mc], dl = L.seq.ll_dnb[lc];
                    const int32_t fm = L.seq.ml_dfs[mc], fl = L.seq.ll_dfs[lc];
                    if (r0 == 0 && a == k) {                 // position 0 of the section
                        const uint32_t sm = seq_fse_init(dm, fm, L.seq.ml_state).st, sl = seq_fse_init(dl, fl, L.seq.ll_state).st;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { cM[q] = sm; cL[q] = sl; }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            cM[q] = seq_fse_step(cM[q], dm, fm, L.seq.ml_state).st;
                            cL[q] = seq_fse_step(cL[q], dl, fl, L.seq.ll_state).st;
                        }
                    }
                }
            }
            bool known = cM[0] == cM[1] && cM[0] == cM[2] && cM[0] == cM[3] && cL[0] == cL[1] && cL[0] == cL[2] && cL[0] == cL[3];
            bool done = a >= rn;                             // nothing to encode: nobody waits for this lane
            known = known || done;
            uint32_t stM = cM[0], stL = cL[0];
            for (;;) {
                if (known && !done) {
                    for (uint32_t i = a; i < b; ++i) {
                        const uint32_t cc = code[SEQ_WARM + i], mc = cc & 0xFFu, lc = cc >> 8;
                        const uint32_t dm = L.seq.ml_dnb[mc], dl = L.seq.ll_dnb[lc];
                        const int32_t fm = L.seq.ml_dfs[mc], fl = L.seq.ll_dfs[lc];
                        const SeqStep m = (first && i == 0) ? seq_fse_init(dm, fm, L.seq.ml_state) : seq_fse_step(stM, dm, fm, L.seq.ml_state);
                        const SeqStep l = (first && i == 0) ? seq_fse_init(dl, fl, L.seq.ll_state) : seq_fse_step(stL, dl, fl, L.seq.ll_state);
                        stM = m.st;
                        stL = l.st;
                        piece[i] = (m.piece | ((stM & 63u) << 9)) | ((l.piece | ((stL & 63u) << 9)) << 15);
                    }
                    done = true;
                }
                if (__ballot(!done) == 0) break;
                const uint32_t pM = wave_prev_lane_u32(stM), pL = wave_prev_lane_u32(stL);
                const bool pdone = wave_prev_lane_u32(done ? 1u : 0u) != 0;   // (lane 0 reads 0: it never looks, below)
                if (!known && pdone) {
                    stM = pM;
                    stL = pL;
                    known = true;
                }
            }
            const int lastLane = (int)((rn - 1u) / SEQ_PER_LANE);
            carryM = (uint32_t)__shfl((int)stM, lastLane, 64);
            carryL = (uint32_t)__shfl((int)stL, lastLane, 64);
        }
        wave_lds_sync();
        // --- assembly, 64 sequences at a time
#pragma unroll
        for (int c = 0; c < SEQ_PER_LANE; ++c) {
            if (64u * c >= rn) break;
            const uint32_t i = 64u * c + (uint32_t)lane;
            const uint32_t t = r0 + i;
            uint64_t v = 0;
            uint32_t len = 0, pc = 0;
            if (i < rn) {
                pc = piece[i];
                v = (uint64_t)(pc & 63u);
                len = (pc >> 6) & 7u;
                v |= (uint64_t)((pc >> 15) & 63u) << len;
                len += (pc >> 21) & 7u;
                v |= (uint64_t)(lexv[c] & 0xFFFFu) << len;
                len += lexv[c] >> 16;
                v |= (uint64_t)(mexv[c] & 0xFFFFu) << len;
                len += mexv[c] >> 16;
                v |= (uint64_t)of_extra << len;   // (at most 44 + 17 bits)
                len += of_code;
            }
            const uint32_t incl = wave_incl_scan_u32(len);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t pos = base_bits + incl - len;
            if (i < rn) {
                const uint32_t n = nseq - 1 - t;  // checkpoint: everything up to and including this sequence's bits is unread
                if (of_dist == 0 && n != 0 && n % spacing == 0)
                    L.cp[n / spacing - 1] = (8u * flushed + base_bits + incl) | (((pc >> 24) & 63u) << 20) | (((pc >> 9) & 63u) << 26);
            }
            if (len) {
                const uint32_t w = pos >> 5, sh = pos & 31;
                const uint64_t lo = v << sh;                 // len <= 61, sh <= 31: may spill into a third word
                atomicOr(&bits[w], (uint32_t)lo);
                if (sh + len > 32) atomicOr(&bits[w + 1], (uint32_t)(lo >> 32));
                if (sh + len > 64) atomicOr(&bits[w + 2], (uint32_t)(v >> (64 - sh)));
            }
            wave_lds_sync();
            const uint32_t allbits = base_bits + total;
            const uint32_t full = allbits >> 5;
            for (uint32_t q = lane; q < full; q += WAVE) {
                const uint32_t wv = bits[q];
                __builtin_memcpy(out + flushed + 4 * q, &wv, 4);
            }
            const uint32_t carry = bits[full];
            wave_lds_sync();
            for (uint32_t q = lane; q <= full; q += WAVE) bits[q] = 0;
            wave_lds_sync();
            if (lane == 0) bits[0] = carry;
            flushed += 4 * full;
            base_bits = allbits & 31;
            wave_lds_sync();
        }
    }
    // final states (match length, then literal length) and the end mark
    uint32_t nbytes = 0;
    if (lane == 0) {
        uint64_t acc = bits[0];
        uint32_t nbit = base_bits;
        acc |= (uint64_t)(carryM & 63u) << nbit; nbit += SEQ_DEF_LOG;
        acc |= (uint64_t)(carryL & 63u) << nbit; nbit += SEQ_DEF_LOG;
        acc |= 1ull << nbit; nbit += 1;
        nbytes = (nbit + 7) >> 3;
        for (uint32_t i = 0; i < nbytes; ++i) out[flushed + i] = (uint8_t)(acc >> (8 * i));
        bits[0] = 0;
    }
    nbytes = (uint32_t)__builtin_amdgcn_readlane((int)nbytes, 0);
    wave_lds_sync();
    return hdr + flushed + nbytes;
}

__device__ __forceinline__ uint32_t span_tmp_bytes(uint32_t S, bool keyseq)
{
    const uint32_t b = S + (S >> 7) + 1024u + (keyseq ? 8u * (S / RMIN + 2u) + 512u : 0u);
    return (b + 15u) & ~15u;
}

// a shared-table span's slot holds the worst case of its table: 11 bits per byte (+ tree description, block headers, jump tables)
__device__ __forceinline__ uint32_t shspan_tmp_bytes(uint32_t S) { return (((S * 11u + 7u) >> 3) + 512u + 32u * ((S + SPAN_BLOCK - 1) / SPAN_BLOCK) + 15u) & ~15u; }

// per read: how its stream is cut.  keyN = spans of the control-byte region, dataN = of the rest.
// shspan: the launch has the shared-table kernels, and this is their span size (0: none); shared (out): this read's data spans use them
__device__ __forceinline__ void span_cut(uint32_t N, uint32_t K, uint32_t shspan, uint32_t& keyN, uint32_t& dataN, bool& shared)
{
    const uint32_t D = N - K;
    shared = shspan != 0 && K != 0 && D >= SHSPAN_MIN_REGION;
    const uint32_t SB = shared ? (N >= SPAN_LARGE_FROM ? SHSPAN_BYTES_LARGE : shspan) : span_bytes_for(N);
    // (with shared tables the call is a matter of latency, whatever the length of the control-byte region: short spans)
    const uint32_t KB = shspan != 0 ? KEYSPAN_BYTES_SHARED : keyspan_bytes_for(K);
    keyN = K == 0 ? 0u : (K + KB - 1) / KB;
    dataN = D ? (D + SB - 1) / SB : 0u;
    if (N == 0) dataN = 1;  // the empty frame
}

// ---- shared tables: the table role of zstd_encode_kernel (see SpanRegion) -----------------------------------------------------------
struct SpanShared
{
    SpanRegion* regions;     // nullptr: no shared tables in this launch
    uint32_t first_block;    // workgroups from this one on are in the table role
    uint32_t shspan;         // the span size the plan was made with
};

// the bytes [p, p + n) into hist (LDS atomics); the requests of up to 8 KB are in flight together
__device__ __forceinline__ void span_count_bytes(uint32_t* hist, const uint8_t* p, uint32_t n, int lane)
{
    uint32_t head = (uint32_t)(-(intptr_t)p) & 15u;
    head = head < n ? head : n;
    const uint32_t nch = (n - head) >> 4, tail0 = head + 16u * nch;
    if ((uint32_t)lane < head) atomicAdd(&hist[p[lane]], 1u);
    if (tail0 + (uint32_t)lane < n) atomicAdd(&hist[p[tail0 + lane]], 1u);
    for (uint32_t c0 = 0; c0 < nch; c0 += 8u * WAVE) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t c = c0 + (uint32_t)lane + 64u * i;
            v[i] = c < nch ? *reinterpret_cast<const uint4*>(p + head + 16u * c) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t c = c0 + (uint32_t)lane + 64u * i;
            if (c < nch) {
                const uint32_t w[4] = { v[i].x, v[i].y, v[i].z, v[i].w };
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(&hist[(w[q] >> (8 * j)) & 0xFFu], 1u);
                }
            }
        }
    }
}

// One workgroup per 4 * D consecutive spans (D = 1 for a single read ... 8 for a batch of large buffers: the fewer workgroups, the
// fewer additions to the reads' histograms -- atomics on 256 words per read, which serialise).  Its four wavefronts count a span each,
// D times, into LDS histograms of their own; the workgroup adds them up per read (nearly always ONE read) and adds the sum to the
// read's histogram.
__global__ __launch_bounds__(256) void zstd_span_count_kernel(ReadBatch b, const EncSpan* spans, const uint32_t* span_count, SpanRegion* regions)
{
    __shared__ uint32_t h[4][256];
    __shared__ uint32_t rd[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t nsp = *span_count;
    uint32_t D = nsp / 512u;
    D = D < 1u ? 1u : (D > 8u ? 8u : D);
    const uint32_t g0 = blockIdx.x * 4u * D;
    if (g0 >= nsp) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j][tid] = 0;
    if (tid < 4) rd[tid] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t d = 0; d < D; ++d) {
        // wavefront wv: span g0 + 4 d + wv
        const uint32_t k = g0 + 4u * d + (uint32_t)wv;
        EncSpan sp = {};
        bool sh = false;
        if (k < nsp) {
            sp = spans[k];
            sh = (sp.flags & (SPAN_SHARED | SPAN_SKIP)) == SPAN_SHARED;
        }
        const uint32_t mine = sh ? sp.read : 0xFFFFFFFFu;
        if (mine != rd[wv]) {   // (wave-uniform) the wavefront moves on to another read: what it has counted so far goes out
            if (rd[wv] != 0xFFFFFFFFu) {
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t v = h[wv][lane + 64 * j];
                    h[wv][lane + 64 * j] = 0;
                    if (v) atomicAdd(&regions[rd[wv]].hist[lane + 64 * j], v);
                }
            }
            wave_lds_sync();
            if (lane == 0) rd[wv] = mine;
            wave_lds_sync();
        }
        if (sh) span_count_bytes(h[wv], b.src + b.src_off[sp.read] + sp.r0, sp.r1 - sp.r0, lane);
    }
    __syncthreads();
    // the wavefronts that ended on the same read as the first are added up and go out together; the others one by one
    const uint32_t r0 = rd[0];
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (rd[j] == r0 && r0 != 0xFFFFFFFFu) v += h[j][tid];
    if (v) atomicAdd(&regions[r0].hist[tid], v);
    for (int j = 1; j < 4; ++j) {
        if (rd[j] == r0 || rd[j] == 0xFFFFFFFFu) continue;
        const uint32_t u = h[j][tid];
        if (u) atomicAdd(&regions[rd[j]].hist[tid], u);
    }
}

// the table role of zstd_encode_kernel<.., TABLES>: one wavefront per read builds the table of its data bytes from the count
__device__ __forceinline__ void span_table_role(EncLds& L, SpanRegion* R, int lane)
{
    const uint32_t ns = R->nshared;
    if (ns == 0) return;
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = R->hist[lane + 64 * j];
        L.hist[lane + 64 * j] = c;
        mine += c;
    }
    const uint32_t D = wave_sum_u32(mine);   // the bytes of the region
    wave_lds_sync();
    const uint32_t per = (D + ns - 1) / ns;
    region_plan(L, D, D, ns * ((per + SPAN_BLOCK - 1) / SPAN_BLOCK), lane);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint2 e = L.ctable[lane + 64 * j];
        R->ctable[lane + 64 * j] = e.x | (e.y << 16);
    }
    if (lane < 34) R->tree[lane] = reinterpret_cast<const uint32_t*>(L.tree)[lane];
    if (lane == 0) {
        R->mode = L.mode;
        R->treeSize = (uint32_t)L.treeSize;
        R->huffLog = L.huffLog;
    }
}

#ifndef VBZ_ENC_PACK_WAVES
#define VBZ_ENC_PACK_WAVES 5   // waves per SIMD the packing launch is compiled for (6: 80 registers with 20 spilled, slower)
#endif
#ifndef VBZ_PACK_PREFETCH
#define VBZ_PACK_PREFETCH 1    // the next step's bytes are requested before the current step is packed
#endif
#ifndef VBZ_ENC_WAVES
#define VBZ_ENC_WAVES 4   // measured: 2.2 ms (4 waves/SIMD, 16 symbols/lane) vs 2.85 ms (2 waves, 32 symbols/lane)
#endif
// TIMED: per-phase shader-clock counters (VBZ_HIP_PHASE_TIMING); a separate instantiation, the counters cost
// dozens of registers in the production kernel otherwise
// DEEP: with the long-repeat matcher (level >= 4); a separate instantiation so that the ordinary kernel does not carry its registers
// This kernel codes a whole frame in one launch: span mode, the long-repeat matcher, phase timers, and on the one-wavefront path every
// read the staged launches below (zstd_plan_kernel, zstd_pack_kernel) leave in redo[] -- then only those; it
// takes the tokeniser's result from the read's plan, because that step works in place.
// (EncRegionPlan / EncPlan: vbz_kernels.h -- the svb encoder fills part of them)

// TABLES: the span-mode instantiation with the shared tables' role (extra wavefronts behind the spans': see SpanRegion)
template <bool TIMED, bool DEEP, bool TABLES = false>
__global__ __launch_bounds__(WAVE, VBZ_ENC_WAVES) void zstd_encode_kernel(ReadBatch b, const uint32_t* orig_size, uint32_t key_elem,
                                                           const uint32_t* key_bytes, uint32_t hdr, unsigned long long* dbg,
                                                           const uint32_t* src_cap, const SeqCTables* seqtab, const EncSpan* spans,
                                                           const uint32_t* span_count, uint8_t* span_tmp, uint32_t* span_size,
                                                           uint32_t* span_trail, uint32_t trailers, uint32_t* deep_d, EncPlan* plans, uint32_t* redo,
                                                           SpanShared shared)
{
    __shared__ __attribute__((aligned(16))) EncLds L;
    if (TABLES && shared.regions && blockIdx.x >= shared.first_block) {   // span mode, shared tables: the wavefronts behind the spans' build the tables
        span_table_role(L, shared.regions + (blockIdx.x - shared.first_block), threadIdx.x);
        return;
    }
    unsigned long long tph[PHASE_SLOTS] = {};
    unsigned long long tlast = TIMED ? __builtin_readcyclecounter() : 0;
#define PHASE(k) do { if (TIMED) { unsigned long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } } while (0)
    const int lane = threadIdx.x;
    // classic mode: workgroup = read, the wave codes the whole frame into the read's destination slot.
    // span mode:    workgroup = span, the wave codes bytes [r0, r1) of the read's stream into the span's temporary slot.
    const bool span_mode = spans != nullptr;
    EncSpan sp = {};
    if (span_mode) {
        if (blockIdx.x >= *span_count) return;
        sp = spans[blockIdx.x];
        if (sp.flags & SPAN_SHARED) return;  // zstd_span_pack_kernel's, in the next launch
        if (lane == 0) { span_size[blockIdx.x] = 0; span_trail[blockIdx.x] = 0; }
        if (sp.flags & SPAN_SKIP) return;  // the read failed earlier: zstd_span_finish_kernel reports it
    }
    const uint32_t r = span_mode ? sp.read : blockIdx.x;
    // the bytes produced (or an error code): the read's result, or the span's size
#define FINISH(v) do { if (lane == 0) { if (span_mode) span_size[blockIdx.x] = (v); else b.result[r] = (v); } } while (0)
    if (redo && !span_mode && redo[r] == PLAN_READY) return;   // the launch behind the staged ones: only what they left (redo[]: their pstate[])
    if (!span_mode && b.gate && b.gate[r] >= GATE_SKIP) {
        if (b.gate[r] != GATE_SKIP) FINISH(b.gate[r]);
        return;
    }
    const uint32_t N = b.src_size[r];
    if (!span_mode && N >= E_FIRST) {  // the svb stage reported an error for this read
        FINISH(N);
        return;
    }
    uint32_t cap = span_mode ? sp.tmp_cap : b.dst_cap[r];
    const uint8_t* in = b.src + b.src_off[r];
    uint8_t* out = span_mode ? span_tmp + sp.tmp_off : b.dst + b.dst_off[r];
    uint32_t K = 0;
    if (N >= SPLIT_MIN) {
        if (key_bytes) K = key_bytes[r];
        else if (key_elem) K = (orig_size[r] / key_elem + 3u) >> 2;
        if (K >= N) K = 0;
    }
#define NEED(bytes)                                              \
    do {                                                         \
        if ((uint64_t)opos + (uint64_t)(bytes) > cap) {          \
            FINISH(E_ZSTD);                                      \
            return;                                              \
        }                                                        \
    } while (0)
    // Long repeats are used in every read that has them, at every level (the reference hands its level to libzstd, whose
    // match finder is on at all of them), in two launches: this kernel without the matcher looks at the distance the svb
    // encoder proposed (deep_d[r]) before it touches anything; a read whose distance holds is left to the second launch, the
    // DEEP instantiation, which runs for those reads only (deep_d[r] stays the distance; 0 = coded here).
    if (deep_d && !span_mode) {
        if (DEEP) {
            if (deep_d[r] == 0) return;
        } else {
            uint32_t hint = deep_d[r];
            if (hint) {
                bool ok = false;
                if (K != 0 && src_cap && seqtab && N - K >= 8192) {
                    DeepLayout dl;
                    if (hint == RUN_D) hint = run_period(in + K, N - K, lane);   // (runs that repeat: their distance)
                    ok = deep_layout(N, K, in, src_cap[r], out, cap, hdr, dl) && period_holds(in + K, N - K, hint, lane);
                }
                if (lane == 0) deep_d[r] = ok ? hint : 0u;
                if (ok) return;
            }
        }
    }
    if (seqtab) {  // 968 bytes of encoding tables, copied once per frame
        const uint32_t* g = reinterpret_cast<const uint32_t*>(seqtab);
        uint32_t* l = reinterpret_cast<uint32_t*>(&L.seq);
        for (uint32_t i = lane; i < sizeof(SeqCTables) / 4; i += WAVE) l[i] = g[i];
    }
    uint32_t opos = 0;
    bool frame_cp = false;  // a sequences section with checkpoints was written
    if (!span_mode || (sp.flags & SPAN_FIRST)) {
        NEED(hdr + 9 + (N == 0 ? 3 : 0));
        if (lane == 0) {
            if (hdr) put_le(out, orig_size[r], 4);
            uint8_t* p = out + hdr;
            put_le(p, 0xFD2FB528u, 4);
            if (N < 256) { p[4] = 0x20; p[5] = (uint8_t)N; }
            else if (N < 65536 + 256) { p[4] = 0x60; put_le(p + 5, N - 256, 2); }
            else { p[4] = 0xA0; put_le(p + 5, N, 4); }
        }
        opos = hdr + 5 + (N < 256 ? 1 : (N < 65536 + 256 ? 2 : 4));
    }
    if (N == 0) {
        if (lane == 0) put_le(out + opos, 1, 3);
        FINISH(opos + 3);
        return;
    }
    // (a Huffman block may come out LONGER than its content -- 11-bit codes from a sampled or region-wide table on a block
    // that does not look like the rest -- and a compressed block above Block_Maximum_Size is one no decoder accepts: blocks
    // are cut so that even 11 bits per byte stay below it)
    uint32_t T = (N + 13) / 14;
    T = T < MIN_BLOCK ? MIN_BLOCK : (T > HUF_BLOCK_MAX ? HUF_BLOCK_MAX : T);
    if (span_mode) T = SPAN_BLOCK;

    bool keys_one_block = false;
    // level >= 4 (one wavefront per frame only): look for one repeat distance in the data bytes.  Behind the stream, below
    // the control-byte region's records: the data bytes' records, and below them one mask bit per data byte.  A data
    // region of more than a block is then coded in chunks of at most a block (a match may not cross a block boundary).
    uint32_t deepD = 0, nunit = 2, chunk = 0;
    // ... and the control bytes of such a read repeat too, at a distance of their own (deepDk): the number of values that make one
    // period of data bytes, in control bytes (times 2 or 4 if that is not a whole number of them).  Their region is then cut into
    // chunks of at most a block like the data bytes', each one block of literals and matches; without it (no period, too few
    // periods, no room) the region is coded as ever -- zero runs in one block, or plain Huffman blocks beyond a block's length.
    uint32_t deepDk = 0, kunits = 1, chunkK = 0;
    bool kzero_chunks = false;   // the control bytes in chunks of at most a block, each with zero-run sequences of its own
    uint16_t* mask16K = nullptr;
    uint16_t* mask16 = nullptr;
    uint8_t* deep_recs = nullptr;
    uint32_t deep_used = 0;   // records of the chunks coded so far
    if (DEEP && !span_mode && deep_d && K != 0 && src_cap && seqtab && N - K >= 8192) {
        const uint32_t SD = N - K, D = deep_d[r];   // the first launch has checked the distance
        DeepLayout dl;
        if ((D >= PERIOD_MIN_D || D == RUN_D) && (uint64_t)D + 32u <= SD && deep_layout(N, K, in, src_cap[r], out, cap, hdr, dl)) {
            period_mask(in + K, SD, D, dl.mask, lane);
            __syncthreads();
            // the matches are counted before anything is moved: they must fit the room their records have
            const uint32_t ch = ((SD + dl.nch - 1u) / dl.nch + 15u) & ~15u;
            uint32_t total = 0;
            for (uint32_t c0 = 0; c0 < SD; c0 += ch) {
                uint32_t lit = 0, n1 = 0;
                tokenise_runs<true, true>(const_cast<uint8_t*>(in + K + c0), (SD - c0) < ch ? (SD - c0) : ch, nullptr, lit, n1, dl.mask + (c0 >> 4), lane);
                total += n1;
            }
            if (total != 0 && total <= dl.rec_cap) {
                mask16 = dl.mask;
                deep_recs = dl.recs;
                cap = dl.cap;   // the frame stays below the workspace
                deepD = D;
                chunk = ch;
                nunit = 1u + dl.nch;
                // the control bytes' own distance
                if (K >= 4096u) {
                    const uint32_t from = K >= 1024u ? 64u : 0u;
                    const uint32_t P = control_period_values(in, K, from, D, lane);
                    const uint32_t Dk = P == 0 ? 0u : ((P & 3u) == 0 ? P >> 2 : ((P & 1u) == 0 ? P >> 1 : P));
                    if (Dk >= PERIOD_MIN_D && (uint64_t)4u * Dk + 64u <= K) {
                        period_mask(in, K, Dk, dl.maskK, lane);
                        __syncthreads();
                        const uint32_t chK = ((K + dl.nchK - 1u) / dl.nchK + 15u) & ~15u;
                        uint32_t totalK = 0, litK = 0;
                        for (uint32_t c0 = 0; c0 < K; c0 += chK) {
                            uint32_t lit = 0, n1 = 0;
                            tokenise_runs<true, true>(const_cast<uint8_t*>(in + c0), (K - c0) < chK ? (K - c0) : chK, nullptr, lit, n1, dl.maskK + (c0 >> 4), lane);
                            totalK += n1;
                            litK += lit;
                        }
                        // (it must pay: at most a quarter of the control bytes stay literals -- the first period and what differs)
                        if (totalK != 0 && total + totalK <= dl.rec_cap && litK <= K / 4u) {
                            mask16K = dl.maskK;
                            deepDk = Dk;
                            chunkK = chK;
                            kunits = dl.nchK;
                            nunit = kunits + dl.nch;
                        }
                    }
                }
                // Control bytes without a distance of their own that are longer than a block (a long read: only the matcher's wavefront
                // sees one whole): chunks of at most a block, each ONE block whose zero runs become sequences -- as plain Huffman blocks a
                // region of zeros cost a bit per byte (iota of 1 M int8 values: 31 KB of its 44 KB; profiles/r06_ratio_sweep.md)
                if (deepDk == 0 && K > BLOCK_MAX) {
                    kzero_chunks = true;
                    chunkK = ((K + dl.nchK - 1u) / dl.nchK + 15u) & ~15u;
                    kunits = dl.nchK;
                    nunit = kunits + dl.nch;
                }
            }
        }
    }
    // A table built from a sampled histogram (region_histogram) gives every byte a word, but on data whose sampled kilobytes do
    // not resemble the rest it can code a region longer than it is, or call compressible data incompressible.  A region
    // coded from a sample that outgrows its own size (or the destination), or that the sample says does not pay, is done
    // again from the exact histogram, which knows what it costs (exact_hist; only regions without sequences are sampled,
    // so nothing that was tokenised is touched twice).
    bool exact_hist = false;
    for (int region = 0; region < (span_mode ? 1 : (int)nunit);) {
        // units: the control bytes (kunits chunks if they have a distance of their own, else one), then the data bytes (chunks if long
        // repeats were found, else one)
        const bool isK = (uint32_t)region < kunits;
        const uint32_t cidx = isK ? (uint32_t)region : (uint32_t)region - kunits;
        const uint32_t r0 = span_mode ? sp.r0 : (isK ? cidx * chunkK : K + cidx * chunk);
        const uint32_t r1 = span_mode ? sp.r1 : (isK ? (((deepDk || kzero_chunks) && K - r0 > chunkK) ? r0 + chunkK : (K ? K : N)) : ((deepD && N - r0 > chunk) ? r0 + chunk : N));
        if (!isK && K == 0) break;
        uint32_t S = r1 - r0;
        const bool lastRegion = span_mode ? (sp.flags & SPAN_LAST) != 0 : (r1 == N);
        const uint8_t* rin = in + r0;
        uint32_t nblk = (S + T - 1) / T;
        if (!isK && keys_one_block) {
            // the control bytes took one block (4 streams): the data bytes get the other 15 x 4 lanes of the decoder
            uint32_t Td = (S + DATA_BLOCKS - 1) / DATA_BLOCKS;
            Td = Td < MIN_BLOCK ? MIN_BLOCK : (Td > HUF_BLOCK_MAX ? HUF_BLOCK_MAX : Td);
            nblk = (S + Td - 1) / Td;
        }
        // control-byte region of a library-owned svb stream: turn long zero runs into sequences.  The
        // literals are compacted in place, the run records go to the unused tail of the scratch slot.
        bool seqmode = false;
        uint32_t nrec = 0;
        const uint2* rec = nullptr;
        if ((span_mode ? (sp.flags & SPAN_KEYSEQ) != 0 : (isK && deepDk == 0)) && src_cap && seqtab && S >= 256 && S <= BLOCK_MAX) {
            const uint32_t slot = src_cap[r];
            // records of all control-byte spans of the frame live behind the stream, each span's at its own offset
            uint32_t keyN = 1, dataN = 0, ord = 0;
            if (span_mode) {
                bool shared_;
                span_cut(N, K, TABLES ? shared.shspan : 0u, keyN, dataN, shared_);   // (only keyN is wanted)
                ord = sp.ord;
            } else if (kzero_chunks) {
                keyN = kunits;
                ord = cidx;
            }
            const uint32_t recs_all = (K ? K : N) / RMIN + 2u * keyN + 2u;  // (a frame without a control-byte region is tokenised as a whole)
            const uint64_t need = (uint64_t)N + 16 + 8ull * recs_all;
            if (need <= slot) {
                uint8_t* ws = const_cast<uint8_t*>(in) + ((slot - 8u * recs_all) & ~7u) + 8u * (r0 / RMIN + 2u * ord);
                uint32_t Lit = 0;
                if (plans && !span_mode && plans[r].tok_done) {
                    // the launch behind the staged ones: zstd_plan_kernel has tokenised this region already
                    nrec = plans[r].tok_nrec;
                    Lit = plans[r].tok_lit;
                } else {
                    tokenise_runs<false, false, TIMED>(const_cast<uint8_t*>(rin), S, reinterpret_cast<uint2*>(ws), Lit, nrec, nullptr, lane, tph, &tlast);
                    __syncthreads();  // the compacted literals and the records are re-read below (vmcnt drain)
                }
                if (nrec) {
                    seqmode = true;
                    rec = reinterpret_cast<const uint2*>(ws);
                    S = Lit;      // from here on the region is its literal stream: one block
                    nblk = 1;
                    keys_one_block = true;
                }
            }
        }
        uint32_t of_dist = 0;  // != 0: this region's sequences carry this explicit distance (long repeats)
        if (DEEP && !span_mode && (isK ? deepDk != 0 : deepD != 0)) {
            uint32_t Lit = 0;
            uint2* recs = reinterpret_cast<uint2*>(deep_recs + 8u * deep_used);
            tokenise_runs<true>(const_cast<uint8_t*>(rin), S, recs, Lit, nrec, isK ? mask16K + (r0 >> 4) : mask16 + ((r0 - K) >> 4), lane);
            deep_used += nrec;
            __syncthreads();
            if (nrec) {  // (without a single match nothing was moved)
                seqmode = true;
                of_dist = isK ? deepDk : deepD;
                rec = recs;
                S = Lit;
                nblk = 1;
            }
        }
        PHASE(0);
        const uint32_t opos_region = opos;
        bool need_redo = false;
        {
        const uint32_t Sh = region_histogram(L, rin, S, lane, !exact_hist && !seqmode);
#define OVERRUN()                                       \
    do {                                                \
        if (Sh != S) {                                  \
            need_redo = true;                           \
            goto region_done;                           \
        }                                               \
        FINISH(E_ZSTD);                                 \
        return;                                         \
    } while (0)
        PHASE(1);
        region_plan(L, S, Sh, nblk, lane, TIMED ? tph : nullptr, &tlast);
        wave_lds_sync();
        PHASE(2);
        if (Sh != S && L.mode != 2u) {   // "does not pay" is not taken from a sample's word
            need_redo = true;
            goto region_done;
        }
        const uint32_t mode = L.mode;
        if (seqmode && mode != 2) {
            // literals do not pay for a Huffman table: Raw_Literals_Block + sequences in one compressed block
            const uint32_t lh = S < 32 ? 1u : (S < 4096 ? 2u : 3u);
            NEED(3ull + lh + S + 8 + 8ull * nrec);
            uint8_t* bp = out + opos;
            if (lane == 0) {
                if (lh == 1) bp[3] = (uint8_t)(S << 3);
                else if (lh == 2) put_le(bp + 3, (S << 4) | 4u, 2);
                else put_le(bp + 3, (S << 4) | 12u, 3);
            }
            for (uint32_t i = lane; i < S; i += WAVE) bp[3 + lh + i] = rin[i];
            const uint32_t sb = encode_zero_run_sequences(L, bp + 3 + lh + S, rec, nrec, lane, of_dist);
            frame_cp = frame_cp || of_dist == 0;
            if (lane == 0) put_le(bp, ((lh + S + sb) << 3) | (2u << 1) | (lastRegion ? 1u : 0u), 3);
            opos += 3 + lh + S + sb;
            goto region_done;
        }
        if (mode == 1) {
            // RLE blocks (Block_Type 1): Block_Size = run length, one byte of content
            const uint32_t nb = (S + BLOCK_MAX - 1) / BLOCK_MAX;
            NEED(4ull * nb);
            if (lane == 0) {
                uint32_t left = S;
                for (uint32_t j = 0; j < nb; ++j) {
                    const uint32_t bs = left > BLOCK_MAX ? BLOCK_MAX : left;
                    left -= bs;
                    const uint32_t last = (lastRegion && j + 1 == nb) ? 1u : 0u;
                    put_le(out + opos + 4 * j, (bs << 3) | (1u << 1) | last, 3);
                    out[opos + 4 * j + 3] = rin[0];
                }
            }
            opos += 4 * nb;
            goto region_done;
        }
        if (mode == 0) {
            // raw blocks (Block_Type 0), copied by the whole wave
            const uint32_t nb = (S + BLOCK_MAX - 1) / BLOCK_MAX;
            NEED(3ull * nb + S);
            uint32_t done = 0;
            for (uint32_t j = 0; j < nb; ++j) {
                const uint32_t bs = (S - done) > BLOCK_MAX ? BLOCK_MAX : (S - done);
                const uint32_t last = (lastRegion && j + 1 == nb) ? 1u : 0u;
                if (lane == 0) put_le(out + opos, (bs << 3) | last, 3);
                opos += 3;
                for (uint32_t i = lane; i < bs; i += WAVE) out[opos + i] = rin[done + i];
                opos += bs;
                done += bs;
            }
            goto region_done;
        }
        // ---- Huffman blocks: passes of up to 16 blocks = 64 streams
        const uint32_t base = S / nblk, extra = S % nblk;
        const uint32_t treeSize = (uint32_t)L.treeSize;
        for (uint32_t b0 = 0; b0 < nblk; b0 += MAXBLK) {
            const uint32_t nb = (nblk - b0) < (uint32_t)MAXBLK ? (nblk - b0) : (uint32_t)MAXBLK;
            const uint32_t bj = b0 + (uint32_t)(lane >> 2);   // this lane describes stream q of block bj
            const int q = lane & 3;
            const bool active = (uint32_t)(lane >> 2) < nb;
            uint32_t bs = 0, boff = 0;
            if (active) {
                bs = base + (bj < extra ? 1u : 0u);
                boff = bj * base + (bj < extra ? bj : extra);
            }
            const bool single = bs < 256;
            const uint32_t seg = single ? bs : (bs + 3) >> 2;
            uint32_t cnt = 0;
            if (active) {
                if (single) cnt = q == 0 ? bs : 0;
                else cnt = q < 3 ? seg : bs - 3 * seg;
            }
            // byte range of the pass inside the region, and every stream's [begin, begin+count)
            const uint32_t pb0 = b0 * base + (b0 < extra ? b0 : extra);
            const uint32_t pbe = (b0 + nb) * base + ((b0 + nb) < extra ? (b0 + nb) : extra);
            // empty streams (single-stream blocks, lanes past the last block) sit at the end of their block
            L.sbeg[lane] = active ? (cnt ? boff + (uint32_t)q * seg : boff + bs) : pbe;
            L.scnt[lane] = cnt;
            L.ssize[lane] = 0;
            wave_lds_sync();
            (void)pb0;
            PHASE(3);
            // --- encode: the wave packs one stream at a time, in frame order, STEP_SYMS symbols per step, from
            // the end of the stream to its start (RFC 8878 4.2.2: the last symbol is written first).  Lane l takes
            // the STEP_LANE symbols that end STEP_LANE*l before the step's end; a wave prefix sum of the lanes'
            // bit counts gives every lane its bit offset; the bits are OR-ed into an LDS buffer that is written
            // out as coalesced dwords.  Nothing is sized in advance: a stream starts where the previous one
            // ended, and a block's headers (whose length only depends on its uncompressed size) are filled in
            // when its last stream is done.
            {
                for (int i = lane; i < OBUF_WORDS; i += WAVE) L.obuf[i] = 0;
                wave_lds_sync();
                // work items = (stream, step) pairs in order; the chunk of the next item is requested
                // before the current one is packed, so its latency is hidden behind the packing
                uint32_t st = 0;
                while (st < 4 * nb && L.scnt[st] == 0) ++st;
                uint32_t done = 0;
                uint32_t cur[STEP_DW], nxt[STEP_DW];
                auto load_chunk = [&](uint32_t sbeg, uint32_t scount, uint32_t dn, uint32_t (&w)[STEP_DW]) {
#pragma unroll
                    for (int k = 0; k < STEP_DW; ++k) w[k] = 0;
                    // symbols of the stream at or below the end of this lane's chunk (streams are at most 128 KB:
                    // 32-bit arithmetic relative to the stream's start)
                    const int32_t room = (int32_t)(scount - dn) - STEP_LANE * lane;
                    if (room > 0) {
                        const uint8_t* p = rin + sbeg + room - STEP_LANE;  // may start before the stream:
                        const bool headroom = (uint64_t)r0 + sbeg >= (uint32_t)STEP_LANE;  // ... but not before the input buffer
                        if (room >= STEP_LANE || headroom) {
                            uint4 v0;
                            __builtin_memcpy(&v0, p, 16);              // those bytes are masked when used
                            w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
                            if (STEP_DW == 8) {
                                uint4 v1;
                                __builtin_memcpy(&v1, p + 16, 16);
                                w[4 % STEP_DW] = v1.x; w[5 % STEP_DW] = v1.y; w[6 % STEP_DW] = v1.z; w[7 % STEP_DW] = v1.w;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < STEP_LANE; ++k) {
                                const uint32_t byte = (room - STEP_LANE + k >= 0) ? (uint32_t)p[k] : 0u;
                                w[k >> 2] |= byte << (8 * (k & 3));
                            }
                        }
                    }
                };
                // block geometry (wave-uniform): header length from the block's uncompressed size alone
                auto blk_bs = [&](uint32_t j) { return base + ((b0 + j) < extra ? 1u : 0u); };
                auto blk_lh = [&](uint32_t j) {
                    const uint32_t jbs = blk_bs(j);
                    const uint32_t worst = ((b0 + j) == 0 ? treeSize : 0u) + 6u + ((jbs * 11u + 7u) >> 3) + 4u;  // every code <= 11 bits
                    const uint32_t big = jbs > worst ? jbs : worst;
                    return 3u + (big >= 1024u ? 1u : 0u) + (big >= 16384u ? 1u : 0u);
                };
                // headers of block j of this pass, which starts at `at` (lane 0 writes; sizes of its streams are in L.ssize)
                auto block_headers = [&](uint32_t j, uint32_t at, uint32_t seqBytes) {
                    if (lane == 0) {
                        const uint32_t jb = b0 + j, jbs = blk_bs(j);
                        const bool jsingle = jbs < 256u;
                        const uint32_t lh = blk_lh(j), tsz = jb == 0 ? treeSize : 0u;
                        const uint32_t s0 = L.ssize[4 * j], s1 = L.ssize[4 * j + 1], s2 = L.ssize[4 * j + 2], s3 = L.ssize[4 * j + 3];
                        const uint32_t lit = tsz + (jsingle ? 0u : 6u) + s0 + s1 + s2 + s3;
                        uint8_t* bp = out + at;
                        const uint32_t last = (lastRegion && jb + 1 == nblk) ? 1u : 0u;
                        put_le(bp, ((lh + lit + seqBytes) << 3) | (2u << 1) | last, 3);
                        const uint64_t type = jb == 0 ? 2 : 3;  // Compressed_Literals_Block / Treeless
                        if (lh == 3) put_le(bp + 3, type | ((jsingle ? 0ull : 1ull) << 2) | ((uint64_t)jbs << 4) | ((uint64_t)lit << 14), 3);
                        else if (lh == 4) put_le(bp + 3, type | (2ull << 2) | ((uint64_t)jbs << 4) | ((uint64_t)lit << 18), 4);
                        else put_le(bp + 3, type | (3ull << 2) | ((uint64_t)jbs << 4) | ((uint64_t)lit << 22), 5);
                        if (!jsingle) {
                            uint8_t* tp = bp + 3 + lh + tsz;
                            put_le(tp, s0, 2);
                            put_le(tp + 2, s1, 2);
                            put_le(tp + 4, s2, 2);
                        }
                    }
                };
                uint32_t ocur = opos;        // where the current block starts
                uint32_t spos = 0;           // where the current stream starts
                uint32_t curblk = 0xFFFFFFFFu;
                if (st < 4 * nb) load_chunk(L.sbeg[st], L.scnt[st], 0, cur);
                uint32_t base_bits = 0;   // bits already in obuf (the partial word carried over)
                uint32_t flushed = 0;     // bytes of the stream already written to memory
#ifdef VBZ_X_NOPACK
                st = 4 * nb;
#endif
                while (st < 4 * nb) {
                    if ((st >> 2) != curblk) {  // first stream of a block: reserve its headers, place the tree
                        curblk = st >> 2;
                        const uint32_t tsz = (b0 + curblk) == 0 ? treeSize : 0u;
                        const uint32_t hl = 3u + blk_lh(curblk) + tsz + (blk_bs(curblk) < 256u ? 0u : 6u);
                        if ((uint64_t)ocur + hl > cap) OVERRUN();
                        for (uint32_t i = lane; i < tsz; i += WAVE) out[ocur + 3u + blk_lh(curblk) + i] = L.tree[i];
                        spos = ocur + hl;
                    }
                    const uint32_t scnt = L.scnt[st];
                    uint8_t* sop = out + spos;
                    // next work item
                    uint32_t nst = st, ndone = done + STEP_SYMS;
                    if (ndone >= scnt) {
                        ndone = 0;
                        ++nst;
                        while (nst < 4 * nb && L.scnt[nst] == 0) ++nst;
                    }
                    if (nst < 4 * nb) load_chunk(L.sbeg[nst], L.scnt[nst], ndone, nxt);
                    // this lane's symbols: positions [lo, hi) of the region, packed from hi-1 down to lo;
                    // the first `skip` of its 32 bytes lie before the stream (or the lane is past its start)
                    const int32_t room = (int32_t)(scnt - done) - STEP_LANE * lane;
                    const int skip = room >= STEP_LANE ? 0 : (room <= 0 ? STEP_LANE : (int)(STEP_LANE - room));
                    uint2 ent[STEP_LANE];   // { code, length } of the lane's symbols
#pragma unroll
                    for (int k = 0; k < STEP_LANE; ++k) ent[k] = L.ctable[(cur[k >> 2] >> (8 * (k & 3))) & 0xFF];
                    if (skip != 0) {  // only the last step of a stream has lanes in front of its start
#pragma unroll
                        for (int k = 0; k < STEP_LANE; ++k) ent[k] = k >= skip ? ent[k] : make_uint2(0u, 0u);
                    }
                    uint32_t Tb = 0;  // bits of this lane's codes
#pragma unroll
                    for (int k = 0; k < STEP_LANE; ++k) Tb += ent[k].y;
                    const uint32_t incl = wave_incl_scan_u32(Tb);
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    const uint32_t allbits = base_bits + total;   // base_bits: bits carried over in quad 0 of the buffer (< 128)
                    const uint32_t fq = allbits >> 7;             // complete 16-byte quads
                    if ((uint64_t)spos + flushed + 16ull * fq + 24 > cap) OVERRUN();
                    {
                        const uint32_t pos = base_bits + incl - Tb;
                        uint32_t word = pos >> 5;
                        uint32_t accbits = pos & 31;
                        uint64_t acc = 0;
#pragma unroll
                        for (int k = STEP_LANE - 1; k >= 0; k -= 2) {
                            // two symbols (at most 22 bits) per flush check: accbits < 32 before, < 54 after
                            const uint2 e1 = ent[k], e0 = ent[k - 1];
                            const uint64_t pair = (uint64_t)(e1.x | (e0.x << e1.y));
                            acc |= pair << accbits;
                            accbits += e1.y + e0.y;
                            if (accbits >= 32) {
                                atomicOr(&L.obuf[word], (uint32_t)acc);
                                acc >>= 32;
                                accbits -= 32;
                                ++word;
                            }
                        }
                        if (acc) atomicOr(&L.obuf[word], (uint32_t)acc);
                    }
                    wave_lds_sync();
                    {   // complete quads leave as 16-byte stores and are cleared on the way; the rest moves to the front
                        uint4* obq = reinterpret_cast<uint4*>(L.obuf);
                        for (uint32_t q = lane; q < fq; q += WAVE) {
                            const uint4 v = obq[q];
                            obq[q] = make_uint4(0u, 0u, 0u, 0u);
                            __builtin_memcpy(sop + flushed + 16u * q, &v, 16);
                        }
                        if (fq) {
                            const uint4 c = obq[fq];
                            wave_lds_sync();
                            if (lane == 0) {
                                obq[fq] = make_uint4(0u, 0u, 0u, 0u);
                                obq[0] = c;
                            }
                        }
                    }
                    flushed += 16u * fq;
                    base_bits = allbits & 127u;
                    wave_lds_sync();
                    if (nst != st) {
                        // stream finished: end mark and the bits still in quad 0 (lane k writes byte k)
                        const uint32_t nbytes = (base_bits + 1 + 7) >> 3;  // <= 16
                        {
                            uint4* obq = reinterpret_cast<uint4*>(L.obuf);
                            const uint4 c = obq[0];
                            const uint32_t cw[4] = { c.x, c.y, c.z, c.w };
                            uint32_t mine = cw[(lane >> 2) & 3];
                            if ((uint32_t)(lane >> 2) == (base_bits >> 5)) mine |= 1u << (base_bits & 31u);
                            if ((uint32_t)lane < nbytes) sop[flushed + lane] = (uint8_t)(mine >> (8 * (lane & 3)));
                            wave_lds_sync();
                            if (lane == 0) {
                                obq[0] = make_uint4(0u, 0u, 0u, 0u);
                                L.ssize[st] = flushed + nbytes;
                            }
                        }
                        spos += flushed + nbytes;
                        base_bits = 0;
                        flushed = 0;
                        if ((nst >> 2) != curblk && !seqmode) {
                            // block finished: a plain block ends with Number_of_Sequences = 0; then the headers
                            wave_lds_sync();
                            if ((uint64_t)spos + 1 > cap) OVERRUN();
                            if (lane == 0) out[spos] = 0;
                            block_headers(curblk, ocur, 1u);
                            ocur = spos + 1u;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < STEP_DW; ++k) cur[k] = nxt[k];
                    st = nst;
                    done = ndone;
                }
                if (seqmode && curblk != 0xFFFFFFFFu) {
                    // a block with sequences is alone in its region: its sequences section and headers are written here, outside
                    // the packing loop (the loop then carries none of the section's registers)
                    wave_lds_sync();
                    if ((uint64_t)spos + 8 + 8ull * nrec > cap) { FINISH(E_ZSTD); return; }
                    const uint32_t seqBytes = encode_zero_run_sequences(L, out + spos, rec, nrec, lane, of_dist);
                    frame_cp = frame_cp || of_dist == 0;
                    wave_lds_sync();
                    for (int i = lane; i < OBUF_WORDS; i += WAVE) L.obuf[i] = 0;  // it used the bit buffer
                    wave_lds_sync();
                    block_headers(curblk, ocur, seqBytes);
                    ocur = spos + seqBytes;
                }
                opos = ocur;
            }
            if (Sh != S && opos - opos_region > S + (S >> 6) + 256u) need_redo = true;
            wave_lds_sync();
            PHASE(5);
        }
#undef OVERRUN
        }
    region_done:
        if (need_redo) {
            exact_hist = true;
            opos = opos_region;
            wave_lds_sync();
        } else {
            exact_hist = false;
            ++region;
        }
    }
    const uint32_t main_bytes = opos;
    if (frame_cp && (trailers & 1u) && (!span_mode || (sp.flags & SPAN_FIRST))) {  // the skippable frame with the decoder checkpoints (optional: only if it fits)
        wave_lds_sync();
        const uint32_t count = L.cpCount, tb = 8u + 4u + 4u * count + 4u;
        if (count != 0 && (uint64_t)opos + tb <= cap) {
            uint8_t* tp = out + opos;
            if (lane == 0) {
                put_le(tp, CP_MAGIC, 4);
                put_le(tp + 4, tb - 8u, 4);
                put_le(tp + 8, L.cpSpacing | (count << 16), 4);
                put_le(tp + 12 + 4 * count, tb, 4);
            }
            if ((uint32_t)lane < count) put_le(tp + 12 + 4 * lane, L.cp[lane], 4);
            opos += tb;
        }
    }
    if (span_mode) {  // the trailer stays behind the span's blocks in its slot; the compaction pass moves it behind the frame
        if (lane == 0) {
            span_size[blockIdx.x] = main_bytes;
            span_trail[blockIdx.x] = opos - main_bytes;
        }
    } else if (lane == 0) {
        b.result[r] = opos;
    }
    if (TIMED && lane == 0 && !span_mode)
        for (int k = 0; k < PHASE_SLOTS; ++k) dbg[(size_t)r * PHASE_SLOTS + k] = tph[k];
#undef PHASE
#undef NEED
#undef FINISH
}

// ---- the planning launches of the staged encoder ---------------------------------------------------------------------------------
// Round 4 planned a read (tokeniser, two histograms, two table constructions, the sequences section) in one launch of the big kernel
// above: 128 registers and 10 KB of LDS per wavefront, 16 wavefronts per CU (profiles/r04_experiments.md: 505 k cycles per frame).  Now
// every read is planned by TWO wavefronts of 8 KB of LDS and 71 registers each (zstd_plan_kernel below: 20 per CU), the data bytes'
// histogram comes from the svb encoder (svb_kernels.hip CNT), and zstd_pack_kernel packs the reads whose plans are complete; everything
// else stays in redo[] for the kernel above.  What round 5 measured on the way (profiles/r05_experiments.md): the launch is bound by
// instruction issue (VALU 65 % busy, LDS 46 % with half of it bank conflicts), not by latency alone -- each phase as a launch of its
// own, at up to 32 wavefronts per CU, was SLOWER (the tables alone, 20 LDS-heavy wavefronts per CU: 1.55 ms per 32 768 reads);
// wavefronts of different phases sharing a CU is what pays.
// Byte for byte the frames the one-launch form writes (tests/test_gpu_soak_slice.py holds the two against each other).
#ifndef VBZ_TABLE_WAVES
#define VBZ_TABLE_WAVES 5
#endif

// per read: block size target, the control-byte region (0: none)
__device__ __forceinline__ uint32_t enc_key_bytes(uint32_t N, uint32_t r, const uint32_t* orig_size, uint32_t key_elem, const uint32_t* key_bytes)
{
    uint32_t K = 0;
    if (N >= SPLIT_MIN) {
        if (key_bytes) K = key_bytes[r];
        else if (key_elem) K = (orig_size[r] / key_elem + 3u) >> 2;
        if (K >= N) K = 0;
    }
    return K;
}
__device__ __forceinline__ uint32_t enc_block_target(uint32_t N)
{
    const uint32_t T = (N + 13) / 14;
    return T < MIN_BLOCK ? MIN_BLOCK : (T > HUF_BLOCK_MAX ? HUF_BLOCK_MAX : T);
}

// One launch, two wavefronts per read (blockIdx = 2 * read + role), so that wavefronts in memory-bound phases (the tokeniser) and in
// LDS-bound phases (sort, package-merge) share a CU at any time -- as separate launches the two lose that overlap (round 5 measured
// tokeniser + sequences 0.59 ms and tables 1.55 ms per 32 768 reads against 1.88 ms for the one-wavefront planning launch of round 4).
//   role 0: the control bytes' tokeniser, their sequences section, the table of region 0 (the literals);
//   role 1: the table of region 1 (the data bytes) from the histogram the svb encoder left -- it cannot wait for role 0, so it GUESSES
//           what only the tokeniser knows (did the control bytes get run sequences: then they are one block and the data bytes get 15),
//           and role 0, which knows, opens the plan only if the guess was right.
// pstate[read] (zeroed before the launch) collects PLAN_OPEN | PLAN_REG0 from role 0 and PLAN_REG1 from role 1; the packing launch packs
// the reads that reach PLAN_READY (and takes a read's state back to 0 if its packing overruns), the one-launch kernel codes the others.
__device__ __forceinline__ bool plan_guess_tokenised(uint32_t N, uint32_t K) { return K != 0 ? (K >= 256 && K <= BLOCK_MAX) : (N >= 256 && N <= BLOCK_MAX); }

struct PlanLds
{
    union
    {
        TokSeqLds ts;
        TableLds tb;
    };
};

// the table of one region into the read's plan; returns whether the region is of the ordinary kind (Huffman coded, one pass of streams)
__device__ __forceinline__ bool plan_region_table(TableLds& L, EncPlan* FP, uint32_t region, const uint8_t* rin, uint32_t S, uint32_t nblk, bool seqmode,
                                                  uint32_t nrec, bool from_plan, int lane)
{
    uint32_t Sh;
    if (from_plan) Sh = region_histogram_from_plan(L, rin, S, lane, true, FP->reg[1].ctable, FP->hist_mode == 2u ? FP->histB : nullptr);
    else Sh = region_histogram(L, rin, S, lane, !seqmode);
    region_plan(L, S, Sh, nblk, lane);
    wave_lds_sync();
    // a sample that says "does not pay" is not believed (the one-launch kernel counts again, exactly); raw / RLE regions and more
    // than 16 blocks are its business too
    if (L.mode != 2u) return false;
    EncRegionPlan* P = &FP->reg[region];
#pragma unroll
    for (int j = 0; j < 4; ++j) P->ctable[lane + 64 * j] = L.ctable[lane + 64 * j];
    if (lane < 34) P->tree[lane] = reinterpret_cast<const uint32_t*>(L.tree)[lane];
    if (lane == 0) {
        P->S = S;
        P->nblk = nblk;
        P->nrec = nrec;
        P->seqmode = seqmode ? 1u : 0u;
        P->Sh = Sh;
        P->treeSize = (uint32_t)L.treeSize;
        P->huffLog = L.huffLog;
    }
    return nblk <= (uint32_t)MAXBLK;
}

__global__ __launch_bounds__(WAVE, VBZ_TABLE_WAVES) void zstd_plan_kernel(ReadBatch b, const uint32_t* orig_size, uint32_t key_elem, const uint32_t* key_bytes,
                                                                          uint32_t hdr, const uint32_t* src_cap, const SeqCTables* seqtab, uint32_t* deep_d,
                                                                          EncPlan* plans, uint32_t* pstate, uint32_t pre_filled)
{
    __shared__ __attribute__((aligned(16))) PlanLds LL;
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x >> 1, role = blockIdx.x & 1u;
    EncPlan* FP = &plans[r];
    if (role == 0 && lane == 0) {
        FP->tok_done = 0;
        FP->seqBytes = 0;
        FP->seqOff = 0;
        FP->cpCount = 0;
        FP->cpSpacing = 0;
    }
    if (b.gate && b.gate[r] >= GATE_SKIP) return;
    const uint32_t N = b.src_size[r];
    if (N >= E_FIRST || N == 0) return;   // (an error of the svb stage, the empty frame: the one-launch kernel reports / writes them)
    const uint8_t* in = b.src + b.src_off[r];
    const uint32_t K = enc_key_bytes(N, r, orig_size, key_elem, key_bytes);
    const uint32_t T = enc_block_target(N);
    const bool guess = src_cap && seqtab && plan_guess_tokenised(N, K);
    if (role == 1) {
        if (K == 0) return;   // (a frame of one region: role 0 says so)
        uint32_t S = N - K;
        uint32_t nblk = (S + T - 1) / T;
        if (guess) {
            // the control bytes took one block (4 streams): the data bytes get the other 15 x 4 lanes of the decoder
            uint32_t Td = (S + DATA_BLOCKS - 1) / DATA_BLOCKS;
            Td = Td < MIN_BLOCK ? MIN_BLOCK : (Td > HUF_BLOCK_MAX ? HUF_BLOCK_MAX : Td);
            nblk = (S + Td - 1) / Td;
        }
        const bool ok = plan_region_table(LL.tb, FP, 1, in + K, S, nblk, false, 0, pre_filled && FP->hist_mode != 0, lane);
        __syncthreads();   // the plan is in memory before the bit that says so
        if (ok && lane == 0) atomicOr(&pstate[r], PLAN_REG1);
        return;
    }
    const uint32_t cap = b.dst_cap[r];
    uint8_t* out = b.dst + b.dst_off[r];
    // a read whose repeat distance holds is the matcher's (the check the one-launch kernel makes at its top; deep_d[r] keeps the verdict)
    if (deep_d) {
        uint32_t hint = deep_d[r];
        if (hint) {
            // (both roles of a read come here and reach the same verdict: whichever value of deep_d[r] the other has left -- the
            // proposal, or the distance it has settled on --, the distance this one settles on is the same)
            bool ok = false;
            if (K != 0 && src_cap && seqtab && N - K >= 8192) {
                DeepLayout dl;
                if (hint == RUN_D) hint = run_period(in + K, N - K, lane);
                ok = deep_layout(N, K, in, src_cap[r], out, cap, hdr, dl) && period_holds(in + K, N - K, hint, lane);
            }
            if (lane == 0) deep_d[r] = ok ? hint : 0u;
            if (ok) return;
        }
    }
    if ((uint64_t)hdr + 9 > cap) return;
    // the control-byte region (a stream too short to be cut in two: the whole stream): long runs become sequences, the literals are
    // compacted in place, the run records go to the unused tail of the scratch slot
    const uint32_t S0 = K ? K : N;
    uint32_t nrec = 0, Lit = 0;
    const uint2* rec = nullptr;
    if (src_cap && seqtab && S0 >= 256 && S0 <= BLOCK_MAX) {
        const uint32_t slot = src_cap[r];
        const uint32_t recs_all = S0 / RMIN + 2u + 2u;
        const uint64_t need = (uint64_t)N + 16 + 8ull * recs_all;
        if (need <= slot) {
            uint8_t* ws = const_cast<uint8_t*>(in) + ((slot - 8u * recs_all) & ~7u);
            tokenise_runs<false>(const_cast<uint8_t*>(in), S0, reinterpret_cast<uint2*>(ws), Lit, nrec, nullptr, lane);
            __syncthreads();  // the compacted literals and the records are re-read below (vmcnt drain)
            rec = reinterpret_cast<const uint2*>(ws);
            if (lane == 0) {
                FP->tok_nrec = nrec;
                FP->tok_lit = Lit;
                FP->tok_done = 1;
            }
        }
    }
    if ((nrec != 0) != guess) return;   // role 1 has planned the data bytes for the other case: the one-launch kernel takes the read
    if (nrec) {
        // the sequences section does not depend on the packing: it is coded here, at the top of the read's destination slot, and the
        // packing launch moves it behind the block's literals (it checks that the frame stays below it)
        const uint64_t room = 24ull + 8ull * nrec;
        if (room + 4096u > cap) return;
        const uint32_t so = (uint32_t)((cap - room) & ~15ull);
        TokSeqLds& L = LL.ts;
        {
            const uint32_t* g = reinterpret_cast<const uint32_t*>(seqtab);   // 968 bytes of encoding tables
            uint32_t* l = reinterpret_cast<uint32_t*>(&L.seq);
            for (uint32_t i = lane; i < sizeof(SeqCTables) / 4; i += WAVE) l[i] = g[i];
        }
        wave_lds_sync();
        const uint32_t sb = encode_zero_run_sequences(L, out + so, rec, nrec, lane, 0u);
        wave_lds_sync();
        FP->cp[lane] = L.cp[lane];
        if (lane == 0) {
            FP->seqBytes = sb;
            FP->seqOff = so;
            FP->cpCount = L.cpCount;
            FP->cpSpacing = L.cpSpacing;
        }
        wave_lds_sync();   // (the table workspace below shares this LDS)
    }
    const uint32_t S = nrec ? Lit : S0;
    const uint32_t nblk = nrec ? 1u : (S + T - 1) / T;
    const bool ok = plan_region_table(LL.tb, FP, 0, in, S, nblk, nrec != 0, nrec, false, lane);
    __syncthreads();
    if (ok && lane == 0) atomicOr(&pstate[r], PLAN_OPEN | PLAN_REG0 | (K == 0 ? PLAN_REG1 : 0u));
}

// ---- the packing launch of the staged encoder ---------------------------------------------------------------------------------
// What zstd_encode_kernel does behind its table constructions, for the reads whose plans the launches above have completed: frame
// header; per region the tree description, the streams packed one after the other in frame order (the same loop: 16 symbols per lane
// and step, a wave prefix sum of the bit counts, bits OR-ed into an LDS buffer, 16-byte quads out), the block headers; the sequences
// section of the first block moved into place; the checkpoint trailer.  Byte for byte the fused kernel's frame.  80 registers and
// 5 KB of LDS instead of 128 (+ spills) and 10 KB: 24 waves per CU instead of 16 (profiles/r04_experiments.md).
#ifndef VBZ_PACK_TABLE_COPIES
#define VBZ_PACK_TABLE_COPIES 1
#endif
constexpr int PACK_COPIES = VBZ_PACK_TABLE_COPIES;   // copies of the code table, one per group of 64 / PACK_COPIES lanes (fewer lanes per bank).
// Measured (round 6; the kernel's LDS pipe is 75 % busy, a fifth of it bank conflicts): two copies +- 0 (6.75 - 6.85 against 6.77 - 6.82 ms for
// the encoder), four copies + 0.75 ms (7.9 KB of LDS: four wavefronts per CU fewer).  One copy stays.
struct PackLds
{
    uint32_t ctable[PACK_COPIES * 256];   // code | length << 16
    uint32_t tree[34];
    uint32_t ssize[WAVE], sbeg[WAVE], scnt[WAVE];
    // Two bit buffers, used in turn: a step's complete quads stay in its buffer and are stored at the top of the NEXT step, in front
    // of the request for the step after that -- the wait for a step's input then never covers a store (one in-order counter)
    uint32_t obuf[2][OBUF_WORDS];
};

// TIMED (VBZ_HIP_PHASE_TIMING=3, a separate instantiation): shader-clock counters per frame -- 0 set-up, 1 a region's table and
// stream geometry, 2 the packing steps' table look-ups and prefix sum, 3 their bits into the LDS buffer, 4 the buffer's quads to
// memory, 5 stream ends and block headers, 6 the sequences section moved into place, 7 trailer and result.
// (dbg: the TIMED instantiation's counters -- an argument of the launch, nothing shared between contexts.  redo[] = pstate[].)
// One region packed with one table, in frame order: blocks of S / nblk bytes, the tree description (treeSize != 0) in the first of them, the
// others treeless.  out + opos is where the first block goes; the frame may grow to `limit`.  false: it does not fit (nothing of the
// caller's state is touched then).  Used by zstd_pack_kernel (a read's two regions) and zstd_span_pack_kernel (a span of a large read).
template <bool TIMED>
__device__ __forceinline__ bool pack_region(PackLds& L, const uint8_t* rin, uint32_t r0, uint8_t* out, uint32_t& opos, uint32_t limit, const uint32_t* ctable_g,
                                            const uint32_t* tree_g, uint32_t S, uint32_t nblk, uint32_t treeSize, bool seqmode, bool lastRegion,
                                            uint32_t seqBytes, uint32_t seqOff, int lane, unsigned long long (&tph)[PHASE_SLOTS], unsigned long long& tlast)
{
#define PPHASE(k) do { if (TIMED) { unsigned long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } } while (0)
#define REDO() return false
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = ctable_g[lane + 64 * j];
#pragma unroll
        for (int cpy = 0; cpy < PACK_COPIES; ++cpy) L.ctable[256 * cpy + lane + 64 * j] = e;
    }
    const uint32_t* const mytable = L.ctable + 256 * (lane / (WAVE / PACK_COPIES));
    if (lane < 34) L.tree[lane] = tree_g[lane];
    const uint32_t base = S / nblk, extra = S % nblk;
    {
        const uint32_t bj = (uint32_t)(lane >> 2);   // this lane describes stream q of block bj
        const int q = lane & 3;
        const bool active = bj < nblk;
        uint32_t bs = 0, boff = 0;
        if (active) {
            bs = base + (bj < extra ? 1u : 0u);
            boff = bj * base + (bj < extra ? bj : extra);
        }
        const bool single = bs < 256;
        const uint32_t seg = single ? bs : (bs + 3) >> 2;
        uint32_t cnt = 0;
        if (active) {
            if (single) cnt = q == 0 ? bs : 0;
            else cnt = q < 3 ? seg : bs - 3 * seg;
        }
        L.sbeg[lane] = active ? (cnt ? boff + (uint32_t)q * seg : boff + bs) : S;
        L.scnt[lane] = cnt;
        L.ssize[lane] = 0;
    }
    for (int i = lane; i < 2 * OBUF_WORDS; i += WAVE) (&L.obuf[0][0])[i] = 0;
    wave_lds_sync();
    // (the stream table is the same for every lane: its entries are read into scalar registers, so that the loop's bookkeeping --
    // stream, step, positions, the limit checks -- runs on the scalar unit beside the lanes' symbol work)
    auto s_cnt = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)L.scnt[i]); };
    auto s_beg = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)L.sbeg[i]); };
    const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nblk);
    uint32_t st = 0;
    while (st < 4 * nb && s_cnt(st) == 0) ++st;
    uint32_t done = 0;
#if VBZ_PACK_PREFETCH
    uint32_t cur[STEP_DW], nxt[STEP_DW];
#else
    uint32_t cur[STEP_DW];
#endif
    auto load_chunk = [&](uint32_t sbeg, uint32_t scount, uint32_t dn, uint32_t (&w)[STEP_DW]) {
#pragma unroll
        for (int k = 0; k < STEP_DW; ++k) w[k] = 0;
        const int32_t room = (int32_t)(scount - dn) - STEP_LANE * lane;
        if (room > 0) {
            const uint8_t* p = rin + sbeg + room - STEP_LANE;  // may start before the stream: those bytes are masked when used
            const bool headroom = (uint64_t)r0 + sbeg >= (uint32_t)STEP_LANE;  // ... but not before the input buffer
            if (room >= STEP_LANE || headroom) {
                uint4 v0;
                __builtin_memcpy(&v0, p, 16);
                w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
            } else {
#pragma unroll
                for (int k = 0; k < STEP_LANE; ++k) {
                    const uint32_t byte = (room - STEP_LANE + k >= 0) ? (uint32_t)p[k] : 0u;
                    w[k >> 2] |= byte << (8 * (k & 3));
                }
            }
        }
    };
    auto blk_bs = [&](uint32_t j) { return base + (j < extra ? 1u : 0u); };
    auto blk_lh = [&](uint32_t j) {
        const uint32_t jbs = blk_bs(j);
        const uint32_t worst = (j == 0 ? treeSize : 0u) + 6u + ((jbs * 11u + 7u) >> 3) + 4u;  // every code <= 11 bits
        const uint32_t big = jbs > worst ? jbs : worst;
        return 3u + (big >= 1024u ? 1u : 0u) + (big >= 16384u ? 1u : 0u);
    };
    auto block_headers = [&](uint32_t j, uint32_t at, uint32_t seqB) {
        if (lane == 0) {
            const uint32_t jbs = blk_bs(j);
            const bool jsingle = jbs < 256u;
            const uint32_t lh = blk_lh(j), tsz = j == 0 ? treeSize : 0u;
            const uint32_t s0 = L.ssize[4 * j], s1 = L.ssize[4 * j + 1], s2 = L.ssize[4 * j + 2], s3 = L.ssize[4 * j + 3];
            const uint32_t lit = tsz + (jsingle ? 0u : 6u) + s0 + s1 + s2 + s3;
            uint8_t* bp = out + at;
            const uint32_t last = (lastRegion && j + 1 == nblk) ? 1u : 0u;
            put_le(bp, ((lh + lit + seqB) << 3) | (2u << 1) | last, 3);
            const uint64_t type = (j == 0 && treeSize) ? 2 : 3;  // Compressed_Literals_Block / Treeless
            if (lh == 3) put_le(bp + 3, type | ((jsingle ? 0ull : 1ull) << 2) | ((uint64_t)jbs << 4) | ((uint64_t)lit << 14), 3);
            else if (lh == 4) put_le(bp + 3, type | (2ull << 2) | ((uint64_t)jbs << 4) | ((uint64_t)lit << 18), 4);
            else put_le(bp + 3, type | (3ull << 2) | ((uint64_t)jbs << 4) | ((uint64_t)lit << 22), 5);
            if (!jsingle) {
                uint8_t* tp = bp + 3 + lh + tsz;
                put_le(tp, s0, 2);
                put_le(tp + 2, s1, 2);
                put_le(tp + 4, s2, 2);
            }
        }
    };
    uint32_t ocur = opos;        // where the current block starts
    uint32_t spos = 0;           // where the current stream starts
    uint32_t curblk = 0xFFFFFFFFu;
    if (st < 4 * nb) load_chunk(s_beg(st), s_cnt(st), 0, cur);
    PPHASE(1);
    uint32_t base_bits = 0;   // bits already in obuf (the partial word carried over)
    uint32_t flushed = 0;     // bytes of the stream already written to memory (or waiting in the other buffer for it)
    uint32_t cb = 0;          // the buffer this step fills
    uint32_t pend_fq = 0;     // quads of the step before that wait in the other buffer ...
    uint8_t* pend_dst = nullptr;  // ... for this address
    auto flush_pending = [&]() {
        uint4* pq = reinterpret_cast<uint4*>(L.obuf[cb ^ 1u]);
        for (uint32_t q = lane; q < pend_fq; q += WAVE) {
            const uint4 v = pq[q];
            pq[q] = make_uint4(0u, 0u, 0u, 0u);
            __builtin_memcpy(pend_dst + 16u * q, &v, 16);
        }
        pend_fq = 0;
    };
    while (st < 4 * nb) {
        if ((st >> 2) != curblk) {  // first stream of a block: reserve its headers, place the tree
            curblk = st >> 2;
            const uint32_t tsz = curblk == 0 ? treeSize : 0u;
            const uint32_t hl = 3u + blk_lh(curblk) + tsz + (blk_bs(curblk) < 256u ? 0u : 6u);
            if ((uint64_t)ocur + hl > limit) REDO();
            for (uint32_t i = lane; i < tsz; i += WAVE) out[ocur + 3u + blk_lh(curblk) + i] = reinterpret_cast<const uint8_t*>(L.tree)[i];
            spos = ocur + hl;
        }
        const uint32_t scnt = s_cnt(st);
        uint8_t* sop = out + spos;
        uint32_t nst = st, ndone = done + STEP_SYMS;
        if (ndone >= scnt) {
            ndone = 0;
            ++nst;
            while (nst < 4 * nb && s_cnt(nst) == 0) ++nst;
        }
        flush_pending();   // (the step before: its quads, in front of the request below)
#if VBZ_PACK_PREFETCH
        if (nst < 4 * nb) load_chunk(s_beg(nst), s_cnt(nst), ndone, nxt);
#endif
        const int32_t room = (int32_t)(scnt - done) - STEP_LANE * lane;
        const int skip = room >= STEP_LANE ? 0 : (room <= 0 ? STEP_LANE : (int)(STEP_LANE - room));
        uint32_t ent[STEP_LANE];   // code | length << 16 of the lane's symbols
#pragma unroll
        for (int k = 0; k < STEP_LANE; ++k) ent[k] = mytable[(cur[k >> 2] >> (8 * (k & 3))) & 0xFF];
        if (skip != 0) {  // only the last step of a stream has lanes in front of its start
#pragma unroll
            for (int k = 0; k < STEP_LANE; ++k) ent[k] = k >= skip ? ent[k] : 0u;
        }
        uint32_t Tb = 0;  // bits of this lane's codes: the entries are code | length << 16 and sixteen codes of at most 11 bits add up to less than 2^16
#pragma unroll
        for (int k = 0; k < STEP_LANE; ++k) Tb += ent[k];
        Tb >>= 16;
        const uint32_t incl = wave_incl_scan_u32(Tb);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t allbits = base_bits + total;   // base_bits: bits carried over in quad 0 of the buffer (< 128)
        const uint32_t fq = allbits >> 7;             // complete 16-byte quads
        if ((uint64_t)spos + flushed + 16ull * fq + 24 > limit) REDO();
        PPHASE(2);
        {
            const uint32_t pos = base_bits + incl - Tb;
            uint32_t word = pos >> 5;
            uint32_t accbits = pos & 31;
            uint64_t acc = 0;
#pragma unroll
            for (int k = STEP_LANE - 1; k >= 0; k -= 2) {
                // two symbols (at most 22 bits) per flush check: accbits < 32 before, < 54 after
                const uint32_t e1 = ent[k], e0 = ent[k - 1];
                const uint32_t l1 = e1 >> 16, l0 = e0 >> 16;
                const uint64_t pair = (uint64_t)((e1 & 0xFFFFu) | ((e0 & 0xFFFFu) << l1));
                acc |= pair << accbits;
                accbits += l1 + l0;
                if (accbits >= 32) {
                    atomicOr(&L.obuf[cb][word], (uint32_t)acc);
                    acc >>= 32;
                    accbits -= 32;
                    ++word;
                }
            }
            if (acc) atomicOr(&L.obuf[cb][word], (uint32_t)acc);
        }
        wave_lds_sync();
        PPHASE(3);
        {   // complete quads wait in this buffer for the top of the next step; the rest moves to the front of the other buffer
            uint4* obq = reinterpret_cast<uint4*>(L.obuf[cb]);
            uint4* nbq = reinterpret_cast<uint4*>(L.obuf[cb ^ 1u]);   // (stored and cleared at the top of this step)
            if (lane == 0) {
                const uint4 c = obq[fq];
                obq[fq] = make_uint4(0u, 0u, 0u, 0u);
                nbq[0] = c;
            }
            pend_fq = fq;
            pend_dst = sop + flushed;
            cb ^= 1u;
        }
        flushed += 16u * fq;
        base_bits = allbits & 127u;
        wave_lds_sync();
        PPHASE(4);
        if (nst != st) {
            // stream finished: end mark and the bits still in quad 0 (lane k writes byte k)
            const uint32_t nbytes = (base_bits + 1 + 7) >> 3;  // <= 16
            {
                uint4* obq = reinterpret_cast<uint4*>(L.obuf[cb]);
                const uint4 c = obq[0];
                const uint32_t cw[4] = { c.x, c.y, c.z, c.w };
                uint32_t mine = cw[(lane >> 2) & 3];
                if ((uint32_t)(lane >> 2) == (base_bits >> 5)) mine |= 1u << (base_bits & 31u);
                if ((uint32_t)lane < nbytes) sop[flushed + lane] = (uint8_t)(mine >> (8 * (lane & 3)));
                wave_lds_sync();
                if (lane == 0) {
                    obq[0] = make_uint4(0u, 0u, 0u, 0u);
                    L.ssize[st] = flushed + nbytes;
                }
            }
            spos += flushed + nbytes;
            base_bits = 0;
            flushed = 0;
            if ((nst >> 2) != curblk && !seqmode) {
                // block finished: a plain block ends with Number_of_Sequences = 0; then the headers
                wave_lds_sync();
                if ((uint64_t)spos + 1 > limit) REDO();
                if (lane == 0) out[spos] = 0;
                block_headers(curblk, ocur, 1u);
                ocur = spos + 1u;
            }
        }
#if VBZ_PACK_PREFETCH
#pragma unroll
        for (int k = 0; k < STEP_DW; ++k) cur[k] = nxt[k];
#else
        if (nst < 4 * nb) load_chunk(s_beg(nst), s_cnt(nst), ndone, cur);
#endif
        st = nst;
        done = ndone;
        PPHASE(5);
    }
    flush_pending();
    if (seqmode && curblk != 0xFFFFFFFFu) {
        // the block with the run sequences: its sequences section, coded by the planning launch above the frame, moves behind
        // the literals (upwards in memory never: the frame has stayed below it)
        wave_lds_sync();
        if ((uint64_t)spos + seqBytes > seqOff) REDO();
        for (uint32_t i = 16u * (uint32_t)lane; i < seqBytes; i += 16u * WAVE) {
            uint4 v;
            __builtin_memcpy(&v, out + seqOff + i, 16);   // (reads up to 15 bytes past the section: inside the slot)
            if (i + 16u <= seqBytes) __builtin_memcpy(out + spos + i, &v, 16);
            else {
                const uint32_t w[4] = { v.x, v.y, v.z, v.w };
                for (uint32_t k = i; k < seqBytes; ++k) out[spos + k] = (uint8_t)(w[(k - i) >> 2] >> (8 * ((k - i) & 3)));
            }
        }
        wave_lds_sync();
        block_headers(curblk, ocur, seqBytes);
        ocur = spos + seqBytes;
    }
    opos = ocur;
    return true;
#undef PPHASE
#undef REDO
}

template <bool TIMED>
__global__ __launch_bounds__(WAVE, VBZ_ENC_PACK_WAVES) void zstd_pack_kernel(ReadBatch b, const uint32_t* orig_size, uint32_t key_elem, const uint32_t* key_bytes,
                                                                             uint32_t hdr, uint32_t trailers, const EncPlan* plans, uint32_t* redo,
                                                                             unsigned long long* dbg_)
{
    unsigned long long* const dbg = TIMED ? dbg_ : nullptr;
    __shared__ __attribute__((aligned(16))) PackLds L;
    unsigned long long tph[PHASE_SLOTS] = {};
    unsigned long long tlast = TIMED ? __builtin_readcyclecounter() : 0;
#define PPHASE(k) do { if (TIMED) { unsigned long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } } while (0)
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x;
    if (redo[r] != PLAN_READY) return;   // (redo[] = pstate[]: how far zstd_plan_kernel has brought the read)
#define REDO()                          \
    do {                                \
        if (lane == 0) redo[r] = 0;     \
        return;                         \
    } while (0)
    const uint32_t N = b.src_size[r];
    const uint32_t cap = b.dst_cap[r];
    const uint8_t* in = b.src + b.src_off[r];
    uint8_t* out = b.dst + b.dst_off[r];
    const EncPlan* FP = &plans[r];
    uint32_t K = 0;
    if (N >= SPLIT_MIN) {
        if (key_bytes) K = key_bytes[r];
        else if (key_elem) K = (orig_size[r] / key_elem + 3u) >> 2;
        if (K >= N) K = 0;
    }
    const uint32_t seqBytes = FP->seqBytes, seqOff = FP->seqOff;
    const uint32_t limit = seqBytes ? seqOff : cap;   // the frame grows below the staged sequences section
    uint32_t opos = 0;
    if ((uint64_t)hdr + 9 > limit) REDO();
    if (lane == 0) {
        if (hdr) put_le(out, orig_size[r], 4);
        uint8_t* p = out + hdr;
        put_le(p, 0xFD2FB528u, 4);
        if (N < 256) { p[4] = 0x20; p[5] = (uint8_t)N; }
        else if (N < 65536 + 256) { p[4] = 0x60; put_le(p + 5, N - 256, 2); }
        else { p[4] = 0xA0; put_le(p + 5, N, 4); }
    }
    opos = hdr + 5 + (N < 256 ? 1 : (N < 65536 + 256 ? 2 : 4));
    PPHASE(0);
    for (int region = 0; region < 2; ++region) {
        if (region == 1 && K == 0) break;
        const uint32_t r0 = region == 0 ? 0u : K;
        const uint32_t r1 = region == 0 ? (K ? K : N) : N;
        const EncRegionPlan* P = &FP->reg[region];
        const uint32_t S = P->S, Sh = P->Sh;
        const uint32_t opos_region = opos;
        if (!pack_region<TIMED>(L, in + r0, r0, out, opos, limit, P->ctable, P->tree, S, P->nblk, P->treeSize, P->seqmode != 0, r1 == N, seqBytes, seqOff, lane,
                                tph, tlast))
            REDO();
        if (Sh != S && opos - opos_region > S + (S >> 6) + 256u) REDO();   // a sample that misled: coded again from the exact histogram
        PPHASE(6);
    }
    if (FP->cpCount != 0 && (trailers & 1u)) {  // the skippable frame with the decoder checkpoints (optional: only if it fits)
        const uint32_t count = FP->cpCount, tb = 8u + 4u + 4u * count + 4u;
        if ((uint64_t)opos + tb <= cap) {
            __syncthreads();   // (the section has left the top of the slot)
            uint8_t* tp = out + opos;
            if (lane == 0) {
                put_le(tp, CP_MAGIC, 4);
                put_le(tp + 4, tb - 8u, 4);
                put_le(tp + 8, FP->cpSpacing | (count << 16), 4);
                put_le(tp + 12 + 4 * count, tb, 4);
            }
            if ((uint32_t)lane < count) put_le(tp + 12 + 4 * lane, FP->cp[lane], 4);
            opos += tb;
        }
    }
    if (lane == 0) b.result[r] = opos;
    PPHASE(7);
    if (TIMED && lane == 0)
        for (int k = 0; k < PHASE_SLOTS; ++k) dbg[(size_t)r * PHASE_SLOTS + k] = tph[k];
#undef PPHASE
#undef REDO
}

// ---- span mode: plan, finish, compaction --------------------------------------------------------------------------------
// Span mode serves batches of few, large reads (a 10 M-element buffer, one 400 k-sample read): the svb stream of a read is cut
// into spans of at most span_bytes_for(N) (the control-byte region and the data-byte region separately; a control-byte region of up
// to 128 KB stays ONE span so that its zero runs can become sequences), one wavefront codes one span into a temporary slot,
// then the spans are strung together behind the frame header.  Behind the frame (and behind the checkpoint trailer, if
// any) goes an INDEX of the spans in a second skippable frame -- where each span's first block starts in the frame and in
// the content -- with which zstd_decode.hip decodes the spans on different wavefronts; like the checkpoints it is verified,
// never trusted (a decoder without it, e.g. libzstd, walks the blocks one after the other).
//   layout: magic 0x184D2A5C, u32 payload bytes, { u32 nspans (bit 31: some spans begin with a treeless block), nspans x { u32 frame offset,
//           u32 content offset }, u32 total bytes }
constexpr uint32_t IDX_MAGIC = 0x184D2A5Cu;

__global__ __launch_bounds__(1024) void zstd_span_plan_kernel(uint32_t n, const uint32_t* svb_size, const uint32_t* orig_size, uint32_t key_elem,
                                                              const uint32_t* gate, uint32_t seq_enabled, uint32_t max_spans, uint64_t tmp_limit,
                                                              EncSpan* spans, uint32_t* span_first, uint32_t* span_count, SpanRegion* regions, uint32_t shspan)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint32_t wcnt[16];
    __shared__ uint64_t carry_b;
    __shared__ uint32_t carry_c;
    // the plan of the reads of one round (1024 reads), read back by all threads when the span descriptors are written
    __shared__ uint32_t q_N[1024], q_K[1024], q_keyN[1024], q_dataN[1024], q_first[1024];
    __shared__ uint64_t q_off[1024];
    __shared__ uint8_t q_shared[1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { carry_b = 0; carry_c = 0; }
    if (regions) {   // histograms and tickets of the shared-table launch start at zero
        for (uint32_t r = (uint32_t)tid >> 8; r < n; r += 4)   // (a quarter of the workgroup per read: no divisions)
            for (uint32_t wd = (uint32_t)tid & 255u; wd < SPANREGION_ZEROED; wd += 256) reinterpret_cast<uint32_t*>(regions + r)[wd] = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        uint32_t cnt = 0, N = 0, K = 0, keyN = 0, dataN = 0;
        uint64_t bytes = 0;
        bool skip = false, shared = false;
        if (i < n) {
            N = svb_size[i];
            skip = (gate && gate[i] >= GATE_SKIP) || N >= E_FIRST;
            if (skip) {
                cnt = 1;
            } else {
                if (N >= SPLIT_MIN && key_elem) {
                    K = (orig_size[i] / key_elem + 3u) >> 2;
                    if (K >= N) K = 0;
                }
                span_cut(N, K, regions ? shspan : 0u, keyN, dataN, shared);
                cnt = keyN + dataN;
                // every span of a region gets the slot of the region's largest span
                if (keyN) bytes += (uint64_t)keyN * span_tmp_bytes((K + keyN - 1) / keyN, seq_enabled != 0);
                // (a small frame without a control-byte region is one span that looks for runs, like the one-wavefront path)
                if (dataN) bytes += (uint64_t)dataN * (shared ? shspan_tmp_bytes((N - K + dataN - 1) / dataN)
                                                              : span_tmp_bytes((N - K + dataN - 1) / dataN, keyN == 0 && dataN == 1 && seq_enabled != 0));
            }
        }
        uint32_t ci = cnt;
        uint64_t bi = bytes;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t tc = (uint32_t)__shfl_up((int)ci, d, 64);
            const uint64_t tb = __shfl_up(bi, d, 64);
            if (lane >= d) { ci += tc; bi += tb; }
        }
        if (lane == 63) { wcnt[w] = ci; wsum[w] = bi; }
        __syncthreads();
        uint32_t pc = carry_c;
        uint64_t pb = carry_b;
        for (int k = 0; k < w; ++k) { pc += wcnt[k]; pb += wsum[k]; }
        if (i < n) {
            const uint32_t si = pc + ci - cnt;
            const uint64_t off = pb + bi - bytes;
            span_first[i] = si;
            // a plan that does not fit the arrays it was given (cannot happen with the host's bounds) skips the read
            const bool fits = si + cnt <= max_spans && off + bytes <= tmp_limit;
            q_first[tid] = si;
            q_off[tid] = off;
            q_N[tid] = N;
            q_K[tid] = K;
            q_keyN[tid] = (skip || !fits) ? 0xFFFFFFFFu : keyN;   // marks a skipped read
            q_dataN[tid] = (skip || !fits) ? cnt : dataN;
            q_shared[tid] = (shared && !skip && fits) ? 1 : 0;
            if (regions && shared && !skip && fits) regions[i].nshared = dataN;
        }
        __syncthreads();
        const uint32_t here = (n - base) < 1024u ? (n - base) : 1024u;
        for (uint32_t q = 0; q < here; ++q) {  // all threads write the spans of read base + q
            const uint32_t si = q_first[q];
            if (q_keyN[q] == 0xFFFFFFFFu) {
                for (uint32_t j = tid; j < q_dataN[q] && si + j < max_spans; j += 1024) {
                    EncSpan e = {};
                    e.read = base + q;
                    e.flags = SPAN_SKIP | (j == 0 ? (SPAN_FIRST | SPAN_LAST) : 0u);
                    spans[si + j] = e;
                }
                continue;
            }
            const uint32_t qN = q_N[q], qK = q_K[q], kN = q_keyN[q], dN = q_dataN[q], D = qN - qK, cntq = kN + dN;
            const bool keyseq1 = seq_enabled != 0;
            const uint32_t slotK = kN ? span_tmp_bytes((qK + kN - 1) / kN, keyseq1) : 0u;
            const bool dataseq = kN == 0 && dN == 1 && seq_enabled != 0 && D >= 256;
            const uint32_t slotD = dN ? (q_shared[q] ? shspan_tmp_bytes((D + dN - 1) / dN) : span_tmp_bytes((D + dN - 1) / dN, kN == 0 && dN == 1 && seq_enabled != 0)) : 0u;
            for (uint32_t j = tid; j < cntq; j += 1024) {
                EncSpan e = {};
                e.read = base + q;
                bool keyseq = false;
                if (j < kN) {
                    e.r0 = (uint32_t)((uint64_t)qK * j / kN);
                    e.r1 = (uint32_t)((uint64_t)qK * (j + 1) / kN);
                    keyseq = keyseq1;
                    e.tmp_off = q_off[q] + (uint64_t)j * slotK;
                    e.tmp_cap = slotK;
                } else {
                    const uint32_t t = j - kN;
                    e.r0 = qK + (uint32_t)((uint64_t)D * t / dN);
                    e.r1 = qK + (uint32_t)((uint64_t)D * (t + 1) / dN);
                    e.tmp_off = q_off[q] + (uint64_t)kN * slotK + (uint64_t)t * slotD;
                    e.tmp_cap = slotD;
                    keyseq = dataseq;
                    if (q_shared[q]) e.flags = SPAN_SHARED | (t == 0 ? SPAN_TREE : 0u);
                }
                e.flags |= (j == 0 ? SPAN_FIRST : 0u) | (j + 1 == cntq ? SPAN_LAST : 0u) | (keyseq ? SPAN_KEYSEQ : 0u);
                e.ord = j;
                spans[si + j] = e;
            }
        }
        __syncthreads();
        if (tid == 1023) { carry_c = pc + ci; carry_b = pb + bi; }
        __syncthreads();
    }
    if (tid == 0) {
        span_first[n] = carry_c;
        *span_count = carry_c < max_spans ? carry_c : max_spans;
    }
}

// ---- shared tables: pack the spans (the count and the table: zstd_encode_kernel's table role, see SpanRegion) --------------------------
// One wavefront per shared span: its bytes as blocks of at most SPAN_BLOCK, packed with the region's table (pack_region: the packing
// launch's loop) into the span's temporary slot -- which holds the worst case, 11 bits per byte, so that the span with the tree
// description can never fail; any other span that comes out larger than raw blocks is stored as raw blocks, like a region whose
// count said that a table does not pay (mode 0); a region of one byte value (mode 1) becomes RLE blocks.
__global__ __launch_bounds__(WAVE, VBZ_ENC_PACK_WAVES) void zstd_span_pack_kernel(ReadBatch b, const EncSpan* spans, const uint32_t* span_count, uint8_t* span_tmp,
                                                                                  uint32_t* span_size, uint32_t* span_trail, const SpanRegion* regions)
{
    __shared__ __attribute__((aligned(16))) PackLds L;
    const int lane = threadIdx.x;
    if (blockIdx.x >= *span_count) return;
    const EncSpan sp = spans[blockIdx.x];
    if ((sp.flags & (SPAN_SHARED | SPAN_SKIP)) != SPAN_SHARED) return;
    const SpanRegion* R = &regions[sp.read];
    const uint8_t* in = b.src + b.src_off[sp.read];
    uint8_t* out = span_tmp + sp.tmp_off;
    const uint32_t S = sp.r1 - sp.r0, cap = sp.tmp_cap;
    const uint32_t nblk = (S + SPAN_BLOCK - 1) / SPAN_BLOCK;
    const bool last = (sp.flags & SPAN_LAST) != 0, tree = (sp.flags & SPAN_TREE) != 0;
    const uint32_t mode = R->mode;
    uint32_t opos = 0;
    bool packed = false;
    if (mode == 2u) {
        unsigned long long tph[PHASE_SLOTS] = {}, tlast = 0;
        packed = pack_region<false>(L, in + sp.r0, sp.r0, out, opos, cap, R->ctable, R->tree, S, nblk, tree ? R->treeSize : 0u, false, last, 0u, 0u, lane, tph, tlast);
        if (packed && !tree && opos > S + 3u * nblk) packed = false;
        if (!packed && tree) {   // (cannot happen: the slot holds 11 bits per byte)
            if (lane == 0) { span_size[blockIdx.x] = E_ZSTD; span_trail[blockIdx.x] = 0; }
            return;
        }
    }
    if (!packed) {
        if ((uint64_t)S + 3ull * nblk > cap) {
            if (lane == 0) { span_size[blockIdx.x] = E_ZSTD; span_trail[blockIdx.x] = 0; }
            return;
        }
        wave_lds_sync();
        const uint32_t base = S / nblk, extra = S % nblk;
        const bool rle = mode == 1u;
        opos = 0;
        for (uint32_t j = 0; j < nblk; ++j) {
            const uint32_t bs = base + (j < extra ? 1u : 0u), boff = j * base + (j < extra ? j : extra);
            const uint8_t* q = in + sp.r0 + boff;
            if (lane == 0) put_le(out + opos, (bs << 3) | (rle ? 2u : 0u) | ((last && j + 1 == nblk) ? 1u : 0u), 3);
            if (rle) {
                if (lane == 0) out[opos + 3] = q[0];
                opos += 4;
            } else {
                for (uint32_t i = lane; i < bs; i += WAVE) out[opos + 3 + i] = q[i];
                opos += 3 + bs;
            }
        }
    }
    if (lane == 0) {
        span_size[blockIdx.x] = opos;
        span_trail[blockIdx.x] = 0;
    }
}

// one workgroup per read: where its spans go in the destination slot, the trailers, the read's result
__global__ __launch_bounds__(256) void zstd_span_finish_kernel(ReadBatch b, uint32_t hdr, const EncSpan* spans, const uint32_t* span_first,
                                                               uint32_t max_spans, const uint32_t* span_size, const uint32_t* span_trail,
                                                               uint32_t* span_dst, uint32_t index_enabled)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t bad_s;
    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t s0 = span_first[r], s1 = span_first[r + 1];
    if (tid == 0) bad_s = 0;
    __syncthreads();
    if (s1 > max_spans || s0 >= s1) {
        if (tid == 0) b.result[r] = E_OOM;
        return;
    }
    if (spans[s0].flags & SPAN_SKIP) {
        if (tid == 0 && !(b.gate && b.gate[r] == GATE_SKIP)) {
            const uint32_t N = b.src_size[r];
            b.result[r] = (b.gate && b.gate[r] >= E_FIRST) ? b.gate[r] : (N >= E_FIRST ? N : E_OOM);
        }
        return;
    }
    uint64_t carry = 0;
    for (uint32_t base = s0; base < s1; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        uint32_t v = i < s1 ? span_size[i] : 0u;
        if (v >= E_FIRST) { atomicOr(&bad_s, 1u); v = 0; }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(v, wsum, tot);
        if (i < s1) span_dst[i] = (uint32_t)(carry + ex);
        carry += tot;
    }
    __syncthreads();
    const uint32_t cap = b.dst_cap[r];
    if (bad_s || carry > cap) {
        if (tid == 0) b.result[r] = E_ZSTD;
        // the compaction pass must not copy anything of this read
        for (uint32_t i = s0 + tid; i < s1; i += 256) span_dst[i] = 0xFFFFFFFFu;
        return;
    }
    uint32_t total = (uint32_t)carry;
    uint8_t* out = b.dst + b.dst_off[r];
    const uint32_t nsp = s1 - s0;
    const uint32_t cp = span_trail[s0];           // the checkpoint trailer of the first span, moved by the compaction pass
    if ((uint64_t)total + cp <= cap) total += cp;
    else if (tid == 0) span_dst[s0] |= 0x80000000u;  // ... which drops it when this bit is set
    const uint32_t ib = 8u + 4u + 8u * nsp + 4u;
    if (index_enabled && nsp > 1 && (uint64_t)total + ib <= cap) {
        uint8_t* tp = out + total;
        if (tid == 0) {
            put_le(tp, IDX_MAGIC, 4);
            put_le(tp + 4, ib - 8u, 4);
            // (bit 31: the frame has treeless spans -- the decoder's plan then looks for the spans that bring trees: IDX_TREELESS)
            put_le(tp + 8, nsp | ((spans[s1 - 1].flags & SPAN_SHARED) ? 0x80000000u : 0u), 4);
            put_le(tp + 12 + 8 * nsp, ib, 4);
        }
        __syncthreads();  // span_dst of every span is written
        const uint32_t fhl = 5u + (b.src_size[r] < 256 ? 1u : (b.src_size[r] < 65536 + 256 ? 2u : 4u));  // frame header length
        for (uint32_t j = tid; j < nsp; j += 256) {
            const uint32_t fo = j == 0 ? fhl : (span_dst[s0 + j] & 0x7FFFFFFFu) - hdr;  // first block of the span, from the frame's magic
            put_le(tp + 12 + 8 * j, fo, 4);
            put_le(tp + 16 + 8 * j, spans[s0 + j].r0, 4);
        }
        total += ib;
    }
    if (tid == 0) b.result[r] = total;
}

// one workgroup per span: move it from its temporary slot to its place in the frame
__global__ __launch_bounds__(256) void zstd_span_compact_kernel(ReadBatch b, const EncSpan* spans, const uint32_t* span_count, const uint8_t* span_tmp,
                                                                const uint32_t* span_size, const uint32_t* span_trail, const uint32_t* span_dst,
                                                                const uint32_t* span_first)
{
    if (blockIdx.x >= *span_count) return;
    const EncSpan sp = spans[blockIdx.x];
    if (sp.flags & SPAN_SKIP) return;
    const uint32_t d = span_dst[blockIdx.x];
    if (d == 0xFFFFFFFFu) return;
    const uint32_t size = span_size[blockIdx.x];
    const uint8_t* s = span_tmp + sp.tmp_off;
    uint8_t* o = b.dst + b.dst_off[sp.read] + (d & 0x7FFFFFFFu);
    const uint32_t nv = size >> 4;  // 16 bytes per lane and trip (the slot is 16-byte aligned, the destination need not be)
    for (uint32_t i = threadIdx.x; i < nv; i += 256) {
        uint4 v;
        __builtin_memcpy(&v, s + 16ull * i, 16);
        __builtin_memcpy(o + 16ull * i, &v, 16);
    }
    for (uint32_t i = (nv << 4) + threadIdx.x; i < size; i += 256) o[i] = s[i];
    if ((sp.flags & SPAN_FIRST) && !(d & 0x80000000u)) {
        // the checkpoint trailer goes behind the last span
        const uint32_t last = span_first[sp.read + 1] - 1;
        const uint32_t end = (span_dst[last] & 0x7FFFFFFFu) + span_size[last];
        const uint32_t tb = span_trail[blockIdx.x];
        uint8_t* t = b.dst + b.dst_off[sp.read] + end;
        for (uint32_t i = threadIdx.x; i < tb; i += 256) t[i] = s[size + i];
    }
}

// ---- the long-repeat matcher for batches on the large-read path ------------------------------------------------------------------
// A batch too small to fill the device runs as spans (vbz_api.hip, use_segments), where no wavefront sees a whole read.  The
// reads of such a batch that repeat at ONE distance -- template-cycling signal, which libzstd codes 10 x smaller at any level --
// are found here and coded by the one-wavefront matcher instantiation after all; the spans skip them (gate_out).
//   period_probe_kernel: svb_kernels.hip's probe as a pass of its own over the read's data bytes (one workgroup per read; the
//     segmented svb kernels have no workgroup that sees the head of the data bytes and the rest): sixteen probe dwords at data
//     bytes p0 .. p0 + 15 in a hash table, the first dword of every 16-byte chunk looked up, a hit verified on 16 bytes, the
//     smallest distance proposed in deep_d[r] (0: none).
//     Its first wavefront then makes the check zstd_encode_kernel<.., false> makes at its top (period_holds, deep_layout):
//     deep_d[r] = the distance or 0, gate_out[r] = GATE_SKIP for the reads the matcher takes.
constexpr uint32_t PP_TABLE = 512, PP_P0 = 256, PP_TRIES = 4, PP_SHIFT = 80;
constexpr uint32_t PROBE_WINDOW = 128u << 10;   // data bytes of a read that the separate probe pass looks at (distances up to ~127 KB)
__device__ __forceinline__ uint32_t pp_hash(uint32_t w) { return (w * 0x9E3779B1u) >> 23; }

__device__ __forceinline__ uint32_t matcher_region(const ReadBatch& b, uint32_t r, const uint32_t* orig_size, uint32_t key_elem, uint32_t max_raw,
                                                   const uint32_t* gate, uint32_t& N)
{
    N = 0;
    if (gate && gate[r] >= GATE_SKIP) return 0;
    N = b.src_size[r];
    if (N >= E_FIRST || N < SPLIT_MIN || !key_elem || orig_size[r] >= max_raw) return 0;
    const uint32_t K = (orig_size[r] / key_elem + 3u) >> 2;
    return (K >= N || N - K < 8192u) ? 0u : K;
}

__global__ __launch_bounds__(256) void period_probe_kernel(ReadBatch b, const uint32_t* orig_size, uint32_t key_elem, uint32_t hdr, const uint32_t* src_cap,
                                                           uint32_t max_raw, const uint32_t* gate, uint32_t* deep_d, uint32_t* gate_out)
{
    __shared__ uint32_t val[PP_TABLE];
    __shared__ uint8_t idx[PP_TABLE];
    __shared__ uint32_t best, lost;
    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    uint32_t N;
    const uint32_t K = matcher_region(b, r, orig_size, key_elem, max_raw, gate, N);
    if (K == 0) {
        if (tid == 0) {
            deep_d[r] = 0;
            gate_out[r] = gate ? gate[r] : 0u;
        }
        return;
    }
    const uint8_t* data = b.src + b.src_off[r] + K;
    const uint32_t SD = N - K;
    auto dword_at = [&](uint32_t at) { return (uint32_t)data[at] | ((uint32_t)data[at + 1] << 8) | ((uint32_t)data[at + 2] << 16) | ((uint32_t)data[at + 3] << 24); };
    for (uint32_t i = tid; i < PP_TABLE; i += 256) val[i] = i == 0 ? 1u : 0u;   // hash(0) = 0, hash(1) != 0: an empty cell matches nothing
    if (tid == 0) { best = 0xFFFFFFFFu; lost = 0; }
    __syncthreads();
    // sixteen distinct probes in sixteen different cells, else another place
    uint32_t p0 = PP_P0;
    bool placed = false;
    for (uint32_t attempt = 0; attempt < PP_TRIES && !placed; ++attempt, p0 += PP_SHIFT) {
        uint32_t pw = 0, h = 0;
        if (tid < 16) {
            pw = dword_at(p0 + tid);
            h = pp_hash(pw);
            val[h] = pw;
            idx[h] = (uint8_t)tid;
        }
        __syncthreads();
        if (tid < 16 && (val[h] != pw || idx[h] != (uint8_t)tid)) lost = 1;
        __syncthreads();
        placed = lost == 0;
        __syncthreads();
        if (!placed) {
            if (tid < 16) val[h] = h == 0 ? 1u : 0u;
            if (tid == 0) lost = 0;
            __syncthreads();
        }
    }
    if (placed) {
        p0 -= PP_SHIFT;   // (the loop's increment)
        // (a long read: the head of its data bytes is looked at -- a period shows there or the read is not periodic; period_holds
        // below asks eight places spread over ALL of it)
        const uint32_t nchunk = (SD < PROBE_WINDOW ? SD : PROBE_WINDOW) >> 4;
        auto look_up = [&](uint32_t at, uint32_t w) {
            const uint32_t h = pp_hash(w);
            if (val[h] == w) {   // rare: compare the chunk with the 16 bytes behind that probe
                const uint32_t from = p0 + idx[h];
                if (at >= from + PERIOD_MIN_D) {
                    bool same = true;
                    for (uint32_t k = 0; k < 16; ++k) same = same && data[at + k] == data[from + k];
                    if (same) atomicMin(&best, at - from);
                }
            }
        };
        // four chunks per thread and trip, their loads in flight together (the pass is a chain of memory round trips otherwise)
        for (uint32_t c = tid; c < nchunk; c += 4 * 256) {
            uint32_t w[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c + 256u * q < nchunk) __builtin_memcpy(&w[q], data + 16u * (c + 256u * q), 4);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c + 256u * q < nchunk) look_up(16u * (c + 256u * q), w[q]);
        }
    }
    __syncthreads();
    // the check zstd_encode_kernel<.., false> makes at its top, by the first wavefront: does the distance hold, is there room
    if (tid < WAVE) {
        // (probes that could not be placed at any of the places tried: the bytes there are all alike -- a run is proposed)
        const uint32_t hint = best == 0xFFFFFFFFu ? ((!placed && SD >= PP_P0 + PP_TRIES * PP_SHIFT + 48u) ? RUN_D : 0u) : best;
        uint32_t D = 0;
        if (hint) {
            const uint8_t* in = b.src + b.src_off[r];
            DeepLayout dl;
            const uint32_t h2 = hint == RUN_D ? run_period(in + K, SD, tid) : hint;
            if (deep_layout(N, K, in, src_cap[r], b.dst + b.dst_off[r], b.dst_cap[r], hdr, dl) && period_holds(in + K, SD, h2, tid)) D = h2;
        }
        if (tid == 0) {
            deep_d[r] = D;
            gate_out[r] = D ? GATE_SKIP : (gate ? gate[r] : 0u);
        }
    }
}

}  // namespace

hipError_t launch_zstd_encode(const ReadBatch& b, const uint32_t* orig_size, uint32_t key_elem, const uint32_t* key_bytes,
                              uint32_t hdr, unsigned long long* dbg, const uint32_t* src_cap, const void* seq_tables, bool trailers,
                              uint32_t* deep_d, void* plan_meta, bool staged, bool pre_filled, unsigned long long* pack_dbg, hipStream_t s)
{
    if (!plan_meta) staged = pre_filled = false;
    const uint32_t tr = (trailers ? ENC_TRAILERS : 0u) | (pre_filled ? ENC_PRE_FILLED : 0u);
    if (b.n_reads == 0) return hipSuccess;
    const SeqCTables* st = reinterpret_cast<const SeqCTables*>(seq_tables);
#ifdef VBZ_EXPERIMENTS   // the timed instantiations (phase cycle counters) are part of the experiments build only
    if (dbg && !plan_meta) {   // phase counters of the whole frame in one launch
        hipLaunchKernelGGL((zstd_encode_kernel<true, false>), dim3(b.n_reads), dim3(WAVE), 0, s, b, orig_size, key_elem, key_bytes, hdr, dbg,
                           src_cap, st, nullptr, nullptr, nullptr, nullptr, nullptr, tr, nullptr, nullptr, nullptr, SpanShared{});
        return hipGetLastError();
    }
#else
    dbg = nullptr;
    pack_dbg = nullptr;
#endif
    EncPlan* plans = reinterpret_cast<EncPlan*>(plan_meta);
    uint32_t* redo = plans ? reinterpret_cast<uint32_t*>(plans + b.n_reads) : nullptr;
    if (staged && src_cap && st) {
        // the ordinary read in stages (tokeniser + sequences section, tables per region, packing); whatever they leave in redo[] in the
        // one-launch form behind them
        (void)hipMemsetAsync(redo, 0, 4ull * b.n_reads, s);   // pstate[]
        hipLaunchKernelGGL(zstd_plan_kernel, dim3(2 * b.n_reads), dim3(WAVE), 0, s, b, orig_size, key_elem, key_bytes, hdr, src_cap, st, deep_d, plans, redo,
                           pre_filled ? 1u : 0u);
#ifdef VBZ_EXPERIMENTS
        if (pack_dbg)   // VBZ_HIP_PHASE_TIMING=3: the packing launch with its phase counters
            hipLaunchKernelGGL(zstd_pack_kernel<true>, dim3(b.n_reads), dim3(WAVE), 0, s, b, orig_size, key_elem, key_bytes, hdr, tr, plans, redo, pack_dbg);
        else
#endif
        hipLaunchKernelGGL(zstd_pack_kernel<false>, dim3(b.n_reads), dim3(WAVE), 0, s, b, orig_size, key_elem, key_bytes, hdr, tr, plans, redo, (unsigned long long*)nullptr);
    } else {
        redo = nullptr;
    }
    hipLaunchKernelGGL((zstd_encode_kernel<false, false>), dim3(b.n_reads), dim3(WAVE), 0, s, b, orig_size, key_elem, key_bytes, hdr, nullptr,
                       src_cap, st, nullptr, nullptr, nullptr, nullptr, nullptr, tr, deep_d, redo ? plans : nullptr, redo, SpanShared{});
    if (deep_d)  // the reads in which the first launch found a repeat distance (it wrote deep_d[] for every read)
        hipLaunchKernelGGL((zstd_encode_kernel<false, true>), dim3(b.n_reads), dim3(WAVE), 0, s, b, orig_size, key_elem, key_bytes, hdr, nullptr,
                           src_cap, st, nullptr, nullptr, nullptr, nullptr, nullptr, tr, deep_d, redo ? plans : nullptr, nullptr, SpanShared{});
    return hipGetLastError();
}

size_t zstd_encode_plan_bytes(uint32_t n_reads) { return (size_t)n_reads * (sizeof(EncPlan) + 4) + 256; }   // plans, pstate[]

// ---- the long-repeat matcher in front of span mode ----------------------------------------------------------------------------
hipError_t launch_zstd_encode_matcher(const ReadBatch& b, const uint32_t* orig_size, uint32_t key_elem, uint32_t hdr, const uint32_t* src_cap,
                                      const void* seq_tables, bool trailers, uint32_t max_raw, uint32_t* deep_d, const uint32_t* gate_in,
                                      uint32_t* gate_out, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    const SeqCTables* st = reinterpret_cast<const SeqCTables*>(seq_tables);
    hipLaunchKernelGGL(period_probe_kernel, dim3(b.n_reads), dim3(256), 0, s, b, orig_size, key_elem, hdr, src_cap, max_raw, gate_in, deep_d, gate_out);
    ReadBatch g = b;
    g.gate = gate_in;
    hipLaunchKernelGGL((zstd_encode_kernel<false, true>), dim3(b.n_reads), dim3(WAVE), 0, s, g, orig_size, key_elem, (const uint32_t*)nullptr, hdr,
                       (unsigned long long*)nullptr, src_cap, st, nullptr, nullptr, nullptr, nullptr, nullptr, trailers ? 1u : 0u, deep_d, nullptr, nullptr, SpanShared{});
    return hipGetLastError();
}

// ---- span mode (few, large reads) ------------------------------------------------------------------------------------------
size_t zstd_span_desc_bytes() { return sizeof(EncSpan); }
size_t zstd_span_region_bytes(uint32_t n_reads) { return (size_t)n_reads * sizeof(SpanRegion); }
uint32_t zstd_span_shared_bytes(uint64_t stream_bytes) { return stream_bytes < SHSPAN_BATCH_FROM ? SHSPAN_BYTES : 0u; }

uint32_t zstd_span_max_spans(uint64_t stream_bytes, uint32_t n_reads)
{
    const uint64_t v = stream_bytes / (KEYSPAN_BYTES_SHARED / 2) + 4ull * n_reads + 1;  // spans are cut evenly: none is below half its limit
    return v > 0x7FFFFFF0ull ? 0u : (uint32_t)v;
}

uint64_t zstd_span_tmp_bytes(uint64_t stream_bytes, uint32_t n_reads, uint32_t max_spans)
{
    // a span's slot: its bytes + 1/128 + 1 KB, and for a control-byte span 8 bytes per possible sequence (<= 8/RMIN per byte)
    return stream_bytes + (stream_bytes >> 7) + stream_bytes * 8u / RMIN + (uint64_t)max_spans * 2048u + 4096u;
}

hipError_t launch_zstd_encode_spans(const ReadBatch& b, const uint32_t* orig_size, uint32_t key_elem, uint32_t hdr, const uint32_t* src_cap,
                                    const void* seq_tables, void* span_desc, uint32_t* span_first, uint32_t* span_count, uint32_t max_spans,
                                    uint8_t* span_tmp, uint64_t span_tmp_bytes, uint32_t* span_size, uint32_t* span_trail, uint32_t* span_dst,
                                    bool index_trailer, void* shared_regions, uint32_t shared_span_bytes, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    EncSpan* spans = reinterpret_cast<EncSpan*>(span_desc);
    SpanRegion* regions = reinterpret_cast<SpanRegion*>(shared_regions);
    hipLaunchKernelGGL(zstd_span_plan_kernel, dim3(1), dim3(1024), 0, s, b.n_reads, b.src_size, orig_size, key_elem, b.gate,
                       (src_cap && seq_tables) ? 1u : 0u, max_spans, span_tmp_bytes, spans, span_first, span_count, regions, shared_span_bytes);
    // (shared tables: the wavefronts behind the spans' are in the table role; the data bytes' spans are packed by the launch behind)
    const SpanShared shared = { regions, max_spans, regions ? shared_span_bytes : 0u };
    if (regions) hipLaunchKernelGGL(zstd_span_count_kernel, dim3((max_spans + 3) / 4), dim3(256), 0, s, b, spans, span_count, regions);
    if (regions)
        hipLaunchKernelGGL((zstd_encode_kernel<false, false, true>), dim3(max_spans + b.n_reads), dim3(WAVE), 0, s, b, orig_size, key_elem, nullptr, hdr, nullptr, src_cap,
                           reinterpret_cast<const SeqCTables*>(seq_tables), spans, span_count, span_tmp, span_size, span_trail, index_trailer ? 1u : 0u,
                           nullptr, nullptr, nullptr, shared);
    else
        hipLaunchKernelGGL((zstd_encode_kernel<false, false>), dim3(max_spans), dim3(WAVE), 0, s, b, orig_size, key_elem, nullptr, hdr, nullptr, src_cap,
                           reinterpret_cast<const SeqCTables*>(seq_tables), spans, span_count, span_tmp, span_size, span_trail, index_trailer ? 1u : 0u,
                           nullptr, nullptr, nullptr, shared);
    if (regions) hipLaunchKernelGGL(zstd_span_pack_kernel, dim3(max_spans), dim3(WAVE), 0, s, b, spans, span_count, span_tmp, span_size, span_trail, regions);
    hipLaunchKernelGGL(zstd_span_finish_kernel, dim3(b.n_reads), dim3(256), 0, s, b, hdr, spans, span_first, max_spans, span_size, span_trail,
                       span_dst, index_trailer ? 1u : 0u);
    hipLaunchKernelGGL(zstd_span_compact_kernel, dim3(max_spans), dim3(256), 0, s, b, spans, span_count, span_tmp, span_size, span_trail, span_dst,
                       span_first);
    return hipGetLastError();
}

// host: the encoding tables of the predefined sequence distributions (uploaded once per context)
size_t seq_tables_bytes() { return sizeof(SeqCTables); }
void seq_tables_build(void* host_buffer) { seq_build_default_ctables(reinterpret_cast<SeqCTables*>(host_buffer)); }

}  // namespace vbzhip
