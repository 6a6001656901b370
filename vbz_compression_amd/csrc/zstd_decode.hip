// zstd_decode.hip -- zstd (RFC 8878) frame decoder for gfx950: the inverse entropy stage of VBZ.
//
// Replaces the reference's ZSTD_getFrameContentSize + ZSTD_decompress calls (vbz/vbz.cpp:236-273;
// external libzstd 1.4.8).  It decodes ANY conforming single frame with a content-size field -- the
// frames the reference wrote with libzstd (block 0 with FSE-coded sequences, 4-stream Huffman
// literals, treeless blocks, raw blocks: SURVEY.md section 8a "zstd frame features") as well as the
// frames of zstd_encode.hip.  oracle/zstd_restate.c is its serial CPU mirror.
//
// One wavefront per frame, 8 waves per CU.  Parallelism comes from the format's independent Huffman bit streams:
// the wave walks the block headers (all lanes parse the same staged bytes; the next block's header is
// requested a block ahead), queues every literals stream of literals-only blocks as a "task", and when 64
// tasks are pending (or a block with sequences needs its history) all 64 lanes decode one stream each, table
// look-ups in LDS.  Frames from zstd_encode.hip keep all 64 lanes busy; libzstd frames offer 4 streams per
// 128 KiB block.  Serial pieces of the format run as wave-uniform code on the scalar unit with their tables
// held across the lanes of vector registers (Huffman weights, FSE state chains).
// Blocks whose sequences are all "repeat offset 1" runs (the control-byte block of zstd_encode.hip) are
// decoded out of order: state chains first (in parallel segments when the encoder's checkpoints are
// present), their literal streams with everybody else's, the runs placed afterwards from prefix sums.
// Other blocks with sequences are executed in order: lane 0 walks the three FSE state machines, the wave
// copies literals and matches cooperatively.
// Algorithmic HBM bytes per svb byte: ~0.67 read + 1 written.
#include "vbz_kernels.h"
#include "svb_wave.h"
#include "zstd_runs.h"

namespace vbzhip {

namespace {

constexpr int HBUF = 768;

#ifndef VBZ_DEC_RING
#define VBZ_DEC_RING 32
#endif
#define VBZ_DEC_RING_DECL VBZ_DEC_RING
// experiment knobs (defaults are the full decoder): entries per Huffman table slot, entries per FSE table
#ifndef VBZ_DEC_LITS_GW
#define VBZ_DEC_LITS_GW 4   // stripes whose loads are in flight together when the wavefront moves literals out of stripes
#endif
#ifndef VBZ_DEC_HUF_SLOT
#define VBZ_DEC_HUF_SLOT 2048
#endif
#ifndef VBZ_DEC_FSE_SLOT
#define VBZ_DEC_FSE_SLOT 512
#endif
constexpr int HUF_SLOT = VBZ_DEC_HUF_SLOT, FSE_SLOT = VBZ_DEC_FSE_SLOT;
struct DecLds
{
    uint16_t huf[2][HUF_SLOT];   // two Huffman decoding tables: symbol | nbBits << 8 (a 12-bit table spans both)
    // The input rings of the stream decoders share their LDS with everything that is only needed while
    // headers are parsed or sequences are executed.  Frames that carry sequences (fse tables must survive
    // from block to block) park those tables in registers while the rings are in use (flush_tasks).
    union
    {
        uint32_t inbuf[VBZ_DEC_RING_DECL + 1][WAVE];  // per-lane rings of compressed input ([slot][lane]), slot 0 mirrors the last
        struct
        {
            uint32_t fse[3][FSE_SLOT];  // LL, OF, ML decoding tables: symbol | nbBits << 8 | base << 16
            uint8_t hbuf[HBUF];    // staged header bytes of the current block section
            int16_t norm[256];
            uint16_t symnext[256];
        } p;
    } u;
    uint32_t wfse[64];       // FSE table of the Huffman weights (accuracy log <= 6)
    uint8_t weights[256];
    uint32_t t_src[WAVE], t_size[WAVE], t_out[WAVE], t_cnt[WAVE], t_tab[WAVE];  // pending stream tasks
    uint32_t t_bit[WAVE], t_end[WAVE];  // split streams only: bits consumed where a lane's piece starts and ends
    uint32_t ctl[24];
};
// one wave per workgroup: the frame's LDS state.  At namespace scope so that functions that are real calls
// (not inlined, to keep the kernel's register footprint down) still address it as LDS.
__shared__ __attribute__((aligned(16))) DecLds L;
#ifdef VBZ_SPLIT_DEBUG
__shared__ uint32_t split_dbg[2];  // (measurements: passes over the pieces of split streams, lanes walking in them)
#endif

// ctl slots
enum { C_ERR = 0, C_A, C_B, C_C, C_D, C_E, C_F, C_G, C_H, C_I, C_J, C_K };

struct BitReader  // backward bit stream (RFC 8878 4.1): bits are consumed from the last byte down
{
    const uint8_t* p;
    uint32_t nextbyte;  // bytes [0, nextbyte) are not yet loaded
    uint64_t buf;       // unread bits, left aligned
    int32_t avail;      // number of valid bits in buf
    bool over;          // tried to read past the beginning

    __device__ __forceinline__ bool init(const uint8_t* ptr, uint32_t n)
    {
        over = false;
        p = ptr;
        buf = 0;
        avail = 0;
        nextbyte = 0;
        if (n == 0) return false;
        const uint32_t last = ptr[n - 1];
        if (last == 0) return false;
        const int hb = 31 - __clz((int)last);
        nextbyte = n - 1;
        buf = hb ? ((uint64_t)(last & ((1u << hb) - 1u)) << (64 - hb)) : 0ull;
        avail = hb;
        refill();
        return true;
    }
    __device__ __forceinline__ void refill()
    {
        if (avail <= 32) {
            if (nextbyte >= 4) {
                uint32_t w;
                __builtin_memcpy(&w, p + nextbyte - 4, 4);
                buf |= (uint64_t)w << (32 - avail);
                nextbyte -= 4;
                avail += 32;
            } else if (nextbyte > 0) {
                uint32_t w = 0;
                for (uint32_t k = 0; k < nextbyte; ++k) w |= (uint32_t)p[k] << (8 * k);
                const int k8 = 8 * (int)nextbyte;
                buf |= (uint64_t)w << (64 - avail - k8);
                avail += k8;
                nextbyte = 0;
            }
        }
    }
    __device__ __forceinline__ uint32_t peek(int nb) const { return nb ? (uint32_t)(buf >> (64 - nb)) : 0u; }
    __device__ __forceinline__ void skip(int nb)
    {
        if (nb > avail) {
            over = true;
            buf = 0;
            avail = 0;
        } else {
            buf <<= nb;
            avail -= nb;
        }
    }
    __device__ __forceinline__ uint32_t read(int nb)  // nb <= 32
    {
        refill();
        const uint32_t v = peek(nb);
        skip(nb);
        return v;
    }
    __device__ __forceinline__ bool finished() const { return !over && avail == 0 && nextbyte == 0; }
};

__device__ __forceinline__ uint32_t lane_put(uint32_t reg, uint32_t l, uint32_t v, int lane) { return (uint32_t)lane == l ? v : reg; }
// the same with wave-uniform l and v in scalar registers: one v_writelane_b32 (the lane select goes through m0: a VOP3
// instruction of gfx9 reads one scalar register)
// (this compiler has no __builtin_amdgcn_writelane; m0 is named as clobbered so that nothing the compiler keeps there --
// it keeps nothing there in this file: no LDS-DMA, no movrel -- is expected to survive, which clang remarks on)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ uint32_t lane_write(uint32_t reg, uint32_t l, uint32_t v)
{
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(v), "s"(l) : "m0");
    return reg;
}
#pragma clang diagnostic pop

// FSE table description (RFC 8878 4.1.1) read from LDS bytes; lane 0 only.
// returns bytes consumed or -1; fills L.u.p.norm[0..nsym)
__device__ int read_ncount(const uint8_t* p, int n, int max_symbol, int max_log, int* out_log, int* out_nsym)
{
    if (n < 1) return -1;
    auto bits = [&](uint32_t bitpos, int k) -> uint32_t {
        uint32_t v = 0;
        const uint32_t by = bitpos >> 3;
        // gather up to 4 bytes (k <= 10 + 7 shift bits)
        for (int i = 0; i < 3; ++i) {
            const uint32_t idx = by + (uint32_t)i;
            v |= (idx < (uint32_t)n ? (uint32_t)p[idx] : 0u) << (8 * i);
        }
        return (v >> (bitpos & 7u)) & ((1u << k) - 1u);
    };
    const int log = (int)(p[0] & 0xF) + 5;
    if (log > max_log) return -1;
    uint32_t bitpos = 4;
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
    bool prev0 = false;
    while (remaining > 1 && sym <= max_symbol) {
        if (prev0) {
            for (;;) {
                const uint32_t rr = bits(bitpos, 2);
                bitpos += 2;
                for (uint32_t k = 0; k < rr; ++k) {
                    if (sym > max_symbol) return -1;
                    L.u.p.norm[sym++] = 0;
                }
                if (rr != 3) break;
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
        L.u.p.norm[sym++] = (int16_t)count;
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

// FSE decoding table from L.u.p.norm (lane 0 only): RFC 8878 4.1.1
__device__ int fse_build(uint32_t* tab, int nsym, int log)
{
    const int size = 1 << log;
    int high = size - 1;
    for (int s = 0; s < nsym; ++s) {
        if (L.u.p.norm[s] == -1) {
            tab[high--] = (uint32_t)s;
            L.u.p.symnext[s] = 1;
        } else {
            L.u.p.symnext[s] = (uint16_t)L.u.p.norm[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; ++s) {
        for (int i = 0; i < L.u.p.norm[s]; ++i) {
            tab[pos] = (uint32_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    }
    if (pos != 0) return -1;
    for (int u = 0; u < size; ++u) {
        const uint32_t s = tab[u] & 0xFF;
        const uint32_t ns = L.u.p.symnext[s]++;
        const int nb = log - hbit(ns);
        tab[u] = s | ((uint32_t)nb << 8) | ((((ns << nb) - (uint32_t)size) & 0xFFFFu) << 16);
    }
    return 0;
}

// one sequence-table definition (lane 0): mode 0 predefined, 1 RLE, 2 FSE, 3 repeat.
// returns bytes consumed from p, or -1
__device__ __noinline__ int seq_table(uint32_t* tab, int* log_io, bool* have, int mode, const uint8_t* p, int n,
                         const int16_t* def, int def_n, int def_log, int max_sym, int max_log)
{
    if (mode == 0) {
        for (int i = 0; i < def_n; ++i) L.u.p.norm[i] = def[i];
        if (fse_build(tab, def_n, def_log) != 0) return -1;
        *log_io = def_log;
        *have = true;
        return 0;
    }
    if (mode == 1) {
        if (n < 1 || p[0] > max_sym) return -1;
        tab[0] = p[0];
        *log_io = 0;
        *have = true;
        return 1;
    }
    if (mode == 2) {
        int log, nsym;
        const int used = read_ncount(p, n, max_sym, max_log, &log, &nsym);
        if (used < 0) return -1;
        if (fse_build(tab, nsym, log) != 0) return -1;
        *log_io = log;
        *have = true;
        return used;
    }
    return *have ? 0 : -1;
}

// Huffman tree description -> weights in L.weights (lane 0). returns bytes consumed or -1; sets nw/log
__device__ __noinline__ int huf_read_weights(const uint8_t* p, int n, int* out_nw, int* out_log)
{
    if (n < 1) return -1;
    int nw = 0, used;
    const int hb = p[0];
    if (hb >= 128) {
        nw = hb - 127;
        used = 1 + (nw + 1) / 2;
        if (used > n) return -1;
        for (int i = 0; i < nw; ++i) L.weights[i] = (i & 1) ? (p[1 + i / 2] & 0xF) : (p[1 + i / 2] >> 4);
    } else {
        used = 1 + hb;
        if (hb == 0 || used > n) return -1;
        int log, nsym;
        // (the weights' alphabet ends at 11 = HUF_TABLELOG_MAX - 1: libzstd >= 1.4.7 refuses a description that lists a symbol beyond it)
        const int hdr = read_ncount(p + 1, hb, 11, 6, &log, &nsym);
        if (hdr < 0) return -1;
        uint32_t* tab = L.wfse;
        if (fse_build(tab, nsym, log) != 0) return -1;
        // two interleaved FSE states over an LDS-resident backward bit stream (at most 127 bytes)
        const uint8_t* q = p + 1 + hdr;
        const int qn = hb - hdr;
        if (qn < 1 || q[qn - 1] == 0) return -1;
        const int top = hbit(q[qn - 1]);
        int left = (qn - 1) * 8 + top;  // unread bits of the stream
        uint64_t buf = top ? ((uint64_t)(q[qn - 1] & ((1u << top) - 1u)) << (64 - top)) : 0ull;
        int avail = top, nextb = qn - 1;
        auto rd = [&](int nb) -> uint32_t {
            while (avail <= 56 && nextb > 0) {
                --nextb;
                buf |= (uint64_t)q[nextb] << (56 - avail);
                avail += 8;
            }
            const uint32_t v = nb ? (uint32_t)(buf >> (64 - nb)) : 0u;
            buf <<= nb;
            avail = avail > nb ? avail - nb : 0;
            left -= nb;
            return v;
        };
        uint32_t s1 = rd(log), s2 = rd(log);
        if (left < 0) return -1;
        for (;;) {
            if (nw > 253) return -1;
            uint32_t e = tab[s1];
            L.weights[nw++] = (uint8_t)e;
            s1 = (e >> 16) + rd((int)((e >> 8) & 0xFF));
            if (left < 0) { L.weights[nw++] = (uint8_t)tab[s2]; break; }
            if (nw > 253) return -1;
            e = tab[s2];
            L.weights[nw++] = (uint8_t)e;
            s2 = (e >> 16) + rd((int)((e >> 8) & 0xFF));
            if (left < 0) { L.weights[nw++] = (uint8_t)tab[s1]; break; }
        }
    }
    uint32_t total = 0;
    int r1 = 0;
    for (int i = 0; i < nw; ++i) {
        const uint32_t wt = L.weights[i];
        if (wt >= 12) return -1;
        total += wt ? (1u << (wt - 1)) : 0u;
        r1 += (wt == 1);
    }
    if (total == 0) return -1;
    const int log = hbit(total) + 1;
    if (log > 12) return -1;
    const uint32_t rest = (1u << log) - total;
    if (rest & (rest - 1)) return -1;
    const uint32_t lastw = (uint32_t)hbit(rest) + 1;
    L.weights[nw++] = (uint8_t)lastw;
    r1 += (lastw == 1);
    if (r1 < 2 || (r1 & 1)) return -1;
    *out_nw = nw;
    *out_log = log;
    return used;
}

// all lanes.  Huffman tree description at g[0..n) -> weights in L.weights, like huf_read_weights, with the description
// (at most 129 bytes) across the lanes of one register (lane j = bytes 4j..4j+3), and so the probabilities and the FSE table
// of the weights (at most 64 cells: every lane builds its own cell).  What is serial by nature -- the table description's
// variable-width fields, the two interleaved state machines over the bit stream -- runs as wave-uniform code on the scalar
// unit; the weights loop takes four weights per trip with no tests in between while neither bits nor weights can run out.
// Returns bytes consumed, -1 for a corrupt description, -2 when the description is legal but does not fit this layout (the
// caller falls back to huf_read_weights).
__device__ __noinline__ int huf_read_tree(const uint8_t* g, uint32_t n_, int lane, uint32_t* out_nw, uint32_t* out_log)
{
    const uint32_t n = uni(n_);
    if (n < 1) return -1;
    uint32_t d = 0;
    if (4u * (uint32_t)lane < n) __builtin_memcpy(&d, g + 4 * lane, 4);  // up to 3 bytes past n: inside the block / arena slack
    const uint32_t hb = lane_get(d, 0) & 0xFF;
    uint32_t nw = 0, used;
    if (hb >= 128) {  // direct representation: 4 bits per weight
        nw = hb - 127;
        used = 1 + (nw + 1) / 2;
        if (used > n) return -1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t i = (uint32_t)lane + 64u * j, b = 1 + i / 2;
            const uint32_t dw = (uint32_t)__shfl((int)d, (int)(b >> 2), 64);
            const uint32_t by = (dw >> (8 * (b & 3))) & 0xFF;
            if (i < nw) L.weights[i] = (uint8_t)((i & 1) ? (by & 0xF) : (by >> 4));
        }
    } else {
        used = 1 + hb;
        if (hb == 0 || used > n) return -1;
        {  // bytes behind the description read as zero
            const uint32_t lo = 4u * (uint32_t)lane;
            if (lo >= used) d = 0;
            else if (lo + 4 > used) d &= (1u << (8 * (used - lo))) - 1u;
        }
        auto bits = [&](uint32_t bitpos, uint32_t k) -> uint32_t {  // k <= 16 bits at absolute bit position bitpos
            const uint32_t idx = bitpos >> 5;
            const uint64_t v = (uint64_t)lane_get(d, idx & 63) | ((uint64_t)lane_get(d, (idx + 1) & 63) << 32);
            return (uint32_t)(v >> (bitpos & 31)) & ((1u << k) - 1u);
        };
        // ---- probabilities (RFC 8878 4.1.1), forward bit stream from byte 1; lane s of `nrm` = count of symbol s
        const uint32_t log = (bits(8, 4)) + 5;
        if (log > 6) return -1;
        uint32_t bitpos = 12, nrm = 0, sym = 0;
        int remaining = (1 << log) + 1, threshold = 1 << log, nbits = (int)log + 1;
        bool prev0 = false;
        // (the weights' alphabet ends at 11 = HUF_TABLELOG_MAX - 1: libzstd >= 1.4.7 refuses a description that lists a symbol beyond it)
        while (remaining > 1 && sym <= 11) {
            if (prev0) {
                for (;;) {
                    const uint32_t rr = bits(bitpos, 2);
                    bitpos += 2;
                    sym += rr;  // zero counts: nrm is already 0 there
                    if (rr != 3) break;
                }
                prev0 = false;
                if (sym > 11) break;
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
            if (sym >= 64) return -2;  // more symbols than lanes: legal, but not what a weight table looks like
            nrm = lane_put(nrm, sym, (uint32_t)count, lane);
            ++sym;
            prev0 = (count == 0);
            while (remaining < threshold) {
                nbits--;
                threshold >>= 1;
            }
        }
        if (remaining != 1) return -1;
        if (sym > 12) return -1;
        const uint32_t nsym = sym;
        const uint32_t hdr = (bitpos - 8 + 7) >> 3;  // bytes of the table description
        if (hdr > hb) return -1;
        // ---- FSE decoding table (4.1.1): lane u of `tab` = cell u (symbol | nbBits << 8 | base << 16)
        const uint32_t size = 1u << log, mask = size - 1;
        // every lane builds its own cell (lane u = cell u; lane s also holds symbol s's count in nrm):
        //   "less than one" symbols (count -1) take the cells from the top down, in symbol order;
        //   the walk k -> (k * step) & mask, k = 0 .. size-1, visits every cell once; the cells it may use (<= high) take the
        //   symbols' occurrences in order, i.e. the i-th usable cell of the walk holds the symbol whose occurrences include i;
        //   a cell's state number is its symbol's count plus the cell's rank among that symbol's cells.
        uint32_t tab;
        {
            const int cnt = (uint32_t)lane < nsym ? (int)nrm : 0;
            const uint64_t below = (1ull << lane) - 1ull;
            const uint64_t lowp = __ballot(cnt == -1);
            const uint32_t nlow = (uint32_t)__popcll(lowp);
            if (nlow >= size) return -1;
            const uint32_t high = size - 1 - nlow;
            const uint32_t pos_cnt = cnt > 0 ? (uint32_t)cnt : 0u;
            const uint32_t cum_incl = wave_incl_scan_u32(pos_cnt);           // occurrences of symbols 0 .. lane
            if ((uint32_t)__builtin_amdgcn_readlane((int)cum_incl, 63) + nlow != size) return -1;
            // (1) low-probability symbols: symbol s with count -1 sits in cell size-1 - (its rank among them)
            uint32_t sym_of_cell = 0xFFFFFFFFu;
            {
                const uint32_t my_top = size - 1u - (uint32_t)__popcll(lowp & below);   // for a lane that is such a symbol
                // hand the symbol number to the lane that is its cell
                for (uint64_t m = lowp; m; m &= m - 1) {
                    const uint32_t sl = (uint32_t)__builtin_ctzll(m);
                    const uint32_t cell = lane_get(my_top, sl);
                    if ((uint32_t)lane == cell) sym_of_cell = sl;
                }
            }
            // (2) the walk: lane k looks at cell (k * step) & mask
            {
                const uint32_t step = (size >> 1) + (size >> 3) + 3;
                const uint32_t cell = ((uint32_t)lane * step) & mask;
                const bool usable = (uint32_t)lane < size && cell <= high;
                const uint64_t um = __ballot(usable);
                const uint32_t occ = (uint32_t)__popcll(um & below);        // which occurrence this cell takes
                // symbol of occurrence occ: the first symbol whose inclusive occurrence count exceeds occ
                uint32_t sy = 0;
                for (uint32_t s2 = 0; s2 + 1 < nsym; ++s2) sy += lane_get(cum_incl, s2) <= occ ? 1u : 0u;
                // send (cell <- sy): lane `cell` must learn sy; cells are a permutation of the lanes below size
                uint32_t sent = 0xFFFFFFFFu;
                if ((uint32_t)lane < size) sent = (uint32_t)__builtin_amdgcn_ds_permute((int)(cell << 2), (int)(usable ? sy : 0xFFFFFFFFu));
                if ((uint32_t)lane < size && sym_of_cell == 0xFFFFFFFFu) sym_of_cell = sent;
            }
            if (__any((uint32_t)lane < size && sym_of_cell >= nsym)) return -1;
            // (3) state numbers in cell order
            uint32_t rank = 0;
            const uint32_t mysym = (uint32_t)lane < size ? sym_of_cell : 0xFFFFFFFFu;
            for (uint32_t s2 = 0; s2 < nsym; ++s2) {
                const uint64_t same = __ballot(mysym == s2);
                if (mysym == s2) rank = (uint32_t)__popcll(same & below);
            }
            const int c0 = (int)(uint32_t)__shfl((int)nrm, (int)(mysym & 63u), 64);
            const uint32_t ns = (c0 == -1 ? 1u : (uint32_t)c0) + rank;
            const uint32_t nb = log - (uint32_t)hbit(ns ? ns : 1u);
            tab = (uint32_t)lane < size ? (mysym | (nb << 8) | ((((ns << nb) - size) & 0xFFFFu) << 16)) : 0u;
        }
        // ---- the weights: two interleaved states over the backward bit stream in bytes [1 + hdr, 1 + hb)
        const uint32_t lowbit = 8 * (1 + hdr);
        const uint32_t lastbyte = bits(8 * hb, 8);
        if (hdr >= hb || lastbyte == 0) return -1;
        const uint32_t P = 8 * hb + (uint32_t)hbit(lastbyte);  // unread bits are [lowbit, P)
        uint32_t dw = d;  // the stream alone: bits below its start read as zero (libzstd's zero fill past the start)
        {
            const uint32_t lo = 32u * (uint32_t)lane;
            if (lo + 32 <= lowbit) dw = 0;
            else if (lo < lowbit) dw &= ~((1u << (lowbit - lo)) - 1u);
        }
        int32_t left = (int32_t)(P - lowbit);
        int32_t idx = (int32_t)((P - 1) >> 5);
        const uint32_t r0 = P - 32u * (uint32_t)idx;  // 1..32 unread bits in the top dword
        uint64_t buf = (uint64_t)(lane_get(dw, (uint32_t)idx) << (32u - r0)) << 32;
        uint32_t have = r0;
        --idx;
        auto rd = [&](uint32_t nb) -> uint32_t {  // nb <= 6
            if (have <= 32) {
                const uint32_t nd = idx >= 0 ? lane_get(dw, (uint32_t)idx) : 0u;
                --idx;
                buf |= (uint64_t)nd << (32u - have);
                have += 32;
            }
            const uint32_t v = (uint32_t)((buf >> 1) >> (63u - nb));
            buf <<= nb;
            have -= nb;
            left -= (int32_t)nb;
            return v;
        };
        uint32_t s1 = rd(log), s2 = rd(log);
        if (left < 0) return -1;
        // Four weights per trip while neither the bit stream (a weight takes at most 6 bits) nor the weight count can run
        // out inside a trip: no tests between the steps, one refill per trip, the weights collected across the lanes of a
        // register (v_writelane) and written 64 at a time.  The last few weights go through the careful loop below.
        {
            uint32_t wcur = 0;   // lane j: weight number (nw & ~63) + j
            while (left >= 24 && nw + 4 <= 252) {
                if (have <= 32) {
                    const uint32_t nd = idx >= 0 ? lane_get(dw, (uint32_t)idx) : 0u;
                    --idx;
                    buf |= (uint64_t)nd << (32u - have);
                    have += 32;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t e = lane_get(tab, (g & 1) ? s2 : s1);
                    wcur = lane_write(wcur, nw & 63u, e);
                    ++nw;
                    const uint32_t nb = (e >> 8) & 0xFF;
                    const uint32_t v = (uint32_t)((buf >> 1) >> (63u - nb));
                    buf <<= nb;
                    have -= nb;
                    left -= (int32_t)nb;
                    if (g & 1) s2 = (e >> 16) + v;
                    else s1 = (e >> 16) + v;
                }
                if ((nw & 63u) == 0) L.weights[nw - 64u + (uint32_t)lane] = (uint8_t)wcur;
            }
            if ((uint32_t)lane < (nw & 63u)) L.weights[(nw & ~63u) + (uint32_t)lane] = (uint8_t)wcur;
        }
        for (;;) {
            if (nw > 253) return -1;
            uint32_t e = lane_get(tab, s1);
            L.weights[nw++] = (uint8_t)e;
            s1 = (e >> 16) + rd((e >> 8) & 0xFF);
            if (left < 0) { L.weights[nw++] = (uint8_t)lane_get(tab, s2); break; }
            if (nw > 253) return -1;
            e = lane_get(tab, s2);
            L.weights[nw++] = (uint8_t)e;
            s2 = (e >> 16) + rd((e >> 8) & 0xFF);
            if (left < 0) { L.weights[nw++] = (uint8_t)lane_get(tab, s1); break; }
        }
    }
    __syncthreads();
    // ---- the implied last weight and the sanity rules of libzstd's HUF_readStats, four weights per lane
    uint32_t part = 0, ones = 0, badw = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = (uint32_t)lane + 64u * j;
        const uint32_t wt = i < nw ? L.weights[i] : 0u;
        badw |= wt >= 12 ? 1u : 0u;
        part += (wt && wt < 12) ? (1u << (wt - 1)) : 0u;
        ones += wt == 1 ? 1u : 0u;
    }
    if (__any(badw)) return -1;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(part), 63);
    uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(ones), 63);
    if (total == 0) return -1;
    const uint32_t tlog = (uint32_t)hbit(total) + 1;
    if (tlog > 12) return -1;
    const uint32_t rest = (1u << tlog) - total;
    if (rest & (rest - 1)) return -1;
    const uint32_t lastw = (uint32_t)hbit(rest) + 1;
    if (lane == 0) L.weights[nw] = (uint8_t)lastw;
    ++nw;
    r1 += lastw == 1 ? 1u : 0u;
    if (r1 < 2 || (r1 & 1)) return -1;
    *out_nw = nw;
    *out_log = tlog;
    __syncthreads();
    return (int)used;
}

// all lanes: fill a Huffman decoding table from L.weights[0..nw) (RFC 8878 4.2.1: increasing weight, then
// increasing symbol value).  Table start of a symbol = cells of all lighter symbols + cells of the equally
// heavy symbols before it, found with ballots in symbol order; short runs are written by the owning lane,
// long ones by the whole wave.
__device__ __noinline__ void huf_fill_table(uint16_t* T, uint32_t nw, uint32_t tlog, int lane)
{
    if (lane < 16) L.ctl[8 + lane] = 0;  // cells per weight live in ctl[8..23]
    __syncthreads();
    uint32_t wt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (uint32_t)lane + 64u * j;
        wt[j] = s < nw ? L.weights[s] : 0u;
        if (wt[j]) atomicAdd(&L.ctl[8 + wt[j]], 1u);
    }
    __syncthreads();
    uint32_t base[13];
    {
        uint32_t acc = 0;
#pragma unroll
        for (int v = 1; v <= 12; ++v) {
            base[v] = acc;
            acc += L.ctl[8 + v] << (v - 1);
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
            const uint32_t bst = (uint32_t)__shfl((int)st[j], src_lane, 64);
            const uint32_t blen = (uint32_t)__shfl((int)len[j], src_lane, 64);
            const uint32_t bent = (uint32_t)__shfl((int)ent, src_lane, 64);
            for (uint32_t i = lane; i < blen; i += WAVE) T[bst + i] = (uint16_t)bent;
        }
    }
    __syncthreads();
}

// ---- per-lane Huffman stream decoding -----------------------------------------------------------------
// Every lane walks its own bit stream, so its loads are scattered by nature.  To keep them wide and
// off the critical path each lane owns a ring of RING dwords in LDS (layout [dword][lane]: conflict
// free when the lanes advance together) that is topped up with 64-byte batches fetched one period
// ahead: at every uniform point (each PERIOD symbols) a lane whose ring has room commits the batch it
// requested a period ago and requests the next one.  A period consumes at most 11 dwords (32 symbols x
// 11 bits), so a ring that holds more than 16 dwords can always skip a top-up: no lane ever runs dry,
// and all ring traffic happens at wave-uniform points (no divergent refill code).
typedef __attribute__((address_space(1))) const uint8_t gcu8;  // global memory, not flat
typedef __attribute__((address_space(1))) uint8_t gu8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef VBZ_DEC_RING
#define VBZ_DEC_RING 32
#endif
constexpr int RING = VBZ_DEC_RING;      // dwords per lane in the LDS ring (32 or 16)
constexpr int BATCH = RING / 2;         // dwords fetched per top-up
constexpr int PERIOD = RING;            // symbols between two top-up points (a period eats <= 11*PERIOD/32 dwords)
#ifndef VBZ_DEC_BURST
#define VBZ_DEC_BURST 2
#endif
constexpr int BURST = VBZ_DEC_BURST;    // periods whose symbols a lane stores together

// next BATCH dwords below `nextbyte`, in consumption order (w[0] holds the highest bytes)
__device__ __forceinline__ void fetch_batch(gcu8* p, uint32_t& nextbyte, uint32_t (&w)[BATCH])
{
    typedef __attribute__((address_space(1), aligned(1))) const u32x4 gq4;
    if (nextbyte >= 4 * BATCH) {
        gcu8* q = p + nextbyte - 4 * BATCH;
#pragma unroll
        for (int v = 0; v < BATCH / 4; ++v) {
            const u32x4 x = *(gq4*)(q + 16 * (BATCH / 4 - 1 - v));
            w[4 * v + 0] = x.w;
            w[4 * v + 1] = x.z;
            w[4 * v + 2] = x.y;
            w[4 * v + 3] = x.x;
        }
        nextbyte -= 4 * BATCH;
    } else {
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            uint32_t v = 0;
            if (nextbyte >= 4) {
                typedef __attribute__((address_space(1), aligned(1))) const uint32_t gq1;
                v = *(gq1*)(p + nextbyte - 4);
                nextbyte -= 4;
            } else if (nextbyte > 0) {
                for (uint32_t j = 0; j < nextbyte; ++j) v |= (uint32_t)p[j] << (8 * (j + 4 - nextbyte));
                nextbyte = 0;
            }
            w[k] = v;
        }
    }
}

// all lanes: decode the queued Huffman streams, one per lane.  Returns true if any stream is corrupt.
//
// A lane's compressed bytes arrive through its LDS ring ([slot][lane], filled 64 bytes at a time one period
// ahead, at wave-uniform points).  The decoder keeps no bit buffer: with c = bits consumed so far (c >= 1, the end
// mark), the next 32 unread bits are {D[k], D[k+1]} << ((c-1) % 32 + 1) >> 32 with k = (c-1) / 32.  Holding
// n = -c makes that one v_alignbit_b32(D[k], D[k+1], n), and storing the ring upside down (dword k in slot
// RING - k % RING, slot 0 mirroring slot RING) makes the pair one ds_read2st64_b32 at slot (n >> 5) % RING.
// 32 fresh bits are good for two symbols (codes are at most 11 bits), so a pair of symbols costs one ring read,
// two table reads and about a dozen ALU operations, with no conditional refill.
//
// split: the tasks are pieces of longer streams (huf_split_plan): a lane starts with t_bit bits consumed and must end
// at exactly t_end.
__device__ __noinline__ bool flush_tasks_ring(const uint8_t* src, uint8_t* dst, uint32_t& ntask, bool split, int lane)
{
    bool bad = false;
    const bool mine = (uint32_t)lane < ntask;
    gcu8* p = (gcu8*)src;
    uint32_t nbytes = 0, cnt = 0, tab = 0;
    gu8* o = (gu8*)dst;
    if (mine) {
        p = (gcu8*)src + L.t_src[lane];
        nbytes = L.t_size[lane];
        o = (gu8*)dst + L.t_out[lane];
        cnt = L.t_cnt[lane];
        tab = L.t_tab[lane];
    }
    const uint32_t sL = 32u - (tab >> 16);  // a table index is the top `log` bits of the fresh word
    const uint16_t* T = &L.huf[0][0] + (tab & 0xFFFF);
    uint32_t* ring = &L.u.inbuf[0][0] + lane;  // slot s of this lane: ring[s * WAVE]

    int32_t n = -1;  // minus the number of bits consumed
    uint32_t nextbyte = 0, widx = 0, end_bits = 8u * nbytes;
    uint32_t pend[BATCH];
    if (split) {
        // the ring holds dwords k0, k0 + 1, ... (counted from the end of the stream), k0 a multiple of BATCH
        const uint32_t c0 = L.t_bit[lane];
        end_bits = L.t_end[lane];
        n = -(int32_t)c0;
        widx = ((c0 - 1u) >> 5) & ~(uint32_t)(BATCH - 1);
        nextbyte = mine ? nbytes - 4u * widx : 0u;
    } else if (mine) {
        const uint32_t last = nbytes ? p[nbytes - 1] : 0;
        if (last == 0) {
            bad = true;
            cnt = 0;
        } else {
            n = -(int32_t)(8 - (31 - __clz((int)last)));  // padding and end mark
            nextbyte = nbytes;
        }
    }
    // initial fill: the whole ring, a further batch in flight.  Past the start of the stream fetch_batch
    // yields zeros, so the ring never runs dry (the final value of n tells real bits from padding).
#define RING_PUT()                                                              \
    do {                                                                        \
        const uint32_t wb__ = (uint32_t)RING - (widx & (uint32_t)(RING - 1));   \
        _Pragma("unroll") for (int k = 0; k < BATCH; ++k) ring[(wb__ - (uint32_t)k) * WAVE] = pend[k]; \
        if (wb__ == (uint32_t)RING) ring[0] = pend[0];                          \
        widx += BATCH;                                                          \
    } while (0)
    for (int f = 0; f < 2; ++f) {
        fetch_batch(p, nextbyte, pend);
        RING_PUT();
    }
    fetch_batch(p, nextbyte, pend);

    // two symbols from 32 fresh bits.  The two ring dwords that hold them are kept in registers (w0 = D[k],
    // w1 = D[k+1]) with the next one (w2 = D[k+2]) requested a pair ahead: a pair consumes at most 22 bits, so k
    // advances by at most one and the ring read is off the dependent chain (two LDS round trips per pair, not three).
    int32_t tprev = n >> 5;
    uint32_t w0 = ring[((((uint32_t)tprev) & (uint32_t)(RING - 1)) + 1u) * WAVE];
    uint32_t w1 = ring[(((uint32_t)tprev) & (uint32_t)(RING - 1)) * WAVE];
    uint32_t w2 = ring[(((uint32_t)tprev - 1u) & (uint32_t)(RING - 1)) * WAVE];
#define HUF_PAIR(e1, e2)                                                               \
    do {                                                                               \
        const int32_t t__ = n >> 5;                                                    \
        const bool adv__ = t__ != tprev;                                               \
        const uint32_t a__ = adv__ ? w1 : w0, b__ = adv__ ? w2 : w1;                   \
        w0 = a__;                                                                      \
        w1 = b__;                                                                      \
        tprev = t__;                                                                   \
        w2 = ring[(((uint32_t)t__ - 1u) & (uint32_t)(RING - 1)) * WAVE];               \
        uint32_t x__ = __builtin_amdgcn_alignbit(a__, b__, (uint32_t)n);               \
        e1 = T[x__ >> sL];                                                             \
        x__ <<= (e1 >> 8);                                                             \
        e2 = T[x__ >> sL];                                                             \
    } while (0)

    while (__any(cnt > 0)) {
        // ---- top-up point (at least BATCH dwords are still unread in the ring afterwards)
        if (widx - (((uint32_t)~n) >> 5) <= (uint32_t)(RING - BATCH)) {
            RING_PUT();
            fetch_batch(p, nextbyte, pend);
        }
        if (PERIOD == 32 && cnt >= 32u * BURST) {
            // BURST periods (a top-up point between them): 64 bytes leave with four adjacent 16-byte stores.  The stream phase
            // lives on the L2 merging what a lane writes into whole lines (non-temporal stores: 2.4 x the kernel's time);
            // 64-byte bursts measured -2 % against 32-byte ones, 128-byte ones cost the second wave per SIMD (281 registers)
            uint32_t ow[8 * BURST];
#pragma unroll
            for (int h = 0; h < BURST; ++h) {
                if (h) {
                    if (widx - (((uint32_t)~n) >> 5) <= (uint32_t)(RING - BATCH)) {
                        RING_PUT();
                        fetch_batch(p, nextbyte, pend);
                    }
                }
#pragma unroll
                for (int q = 8 * h; q < 8 * h + 8; ++q) {
                    uint32_t e1, e2, e3, e4;
                    HUF_PAIR(e1, e2);
                    n -= (int32_t)((e1 >> 8) + (e2 >> 8));
                    HUF_PAIR(e3, e4);
                    n -= (int32_t)((e3 >> 8) + (e4 >> 8));
                    ow[q] = (e1 & 0xFFu) | ((e2 & 0xFFu) << 8) | ((e3 & 0xFFu) << 16) | (e4 << 24);
                }
            }
            typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
#pragma unroll
            for (int q = 0; q < 2 * BURST; ++q) {
                const u32x4 ov = { ow[4 * q], ow[4 * q + 1], ow[4 * q + 2], ow[4 * q + 3] };
                *(gs4*)(o + 16 * q) = ov;
            }
            o += 32 * BURST;
            cnt -= 32 * BURST;
            continue;
        }
        // ---- one period: groups of 16 symbols, each stored with one 16-byte write
        if (PERIOD == 32 && cnt >= 32) {
            // a whole period: 32 bytes leave with two adjacent 16-byte stores (one full 32-byte sector per lane)
            uint32_t ow[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint32_t e1, e2, e3, e4;
                HUF_PAIR(e1, e2);
                n -= (int32_t)((e1 >> 8) + (e2 >> 8));
                HUF_PAIR(e3, e4);
                n -= (int32_t)((e3 >> 8) + (e4 >> 8));
                ow[q] = (e1 & 0xFFu) | ((e2 & 0xFFu) << 8) | ((e3 & 0xFFu) << 16) | (e4 << 24);
            }
            typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
            const u32x4 ov0 = { ow[0], ow[1], ow[2], ow[3] }, ov1 = { ow[4], ow[5], ow[6], ow[7] };
            *(gs4*)o = ov0;
            *(gs4*)(o + 16) = ov1;
            o += 32;
            cnt -= 32;
            continue;
        }
#pragma unroll
        for (int g = 0; g < PERIOD / 16; ++g) {
            if (cnt >= 16) {
                uint32_t ow[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t e1, e2, e3, e4;
                    HUF_PAIR(e1, e2);
                    n -= (int32_t)((e1 >> 8) + (e2 >> 8));
                    HUF_PAIR(e3, e4);
                    n -= (int32_t)((e3 >> 8) + (e4 >> 8));
                    ow[q] = (e1 & 0xFFu) | ((e2 & 0xFFu) << 8) | ((e3 & 0xFFu) << 16) | (e4 << 24);
                }
                typedef __attribute__((address_space(1), aligned(1))) u32x4 gs4;
                const u32x4 ov = { ow[0], ow[1], ow[2], ow[3] };
                *(gs4*)o = ov;
                o += 16;
                cnt -= 16;
            } else if (cnt > 0) {
                while (cnt > 0) {
                    uint32_t e1, e2;
                    HUF_PAIR(e1, e2);
                    (void)e2;  // only the first symbol of the pair is taken
                    n -= (int32_t)(e1 >> 8);
                    *o++ = (uint8_t)e1;
                    --cnt;
                }
            }
        }
    }
#undef HUF_PAIR
#undef RING_PUT
    if (mine && !bad && n != -(int32_t)end_bits) bad = true;  // every bit of the stream must be consumed, none beyond
    ntask = 0;
    __syncthreads();  // also makes the decoded bytes visible to the whole wave (vmcnt drain)
    return __any(bad);
}

// ---- long Huffman streams, split ----------------------------------------------------------------------
// libzstd writes the literals of a block as four streams of up to 32 KB: four busy lanes, sixty idle ones, and the
// longest dependent chain of the whole frame.  Huffman codes resynchronise: a decoder started at an arbitrary bit
// falls into step with the real code boundaries after a few dozen symbols.  So every stream is cut into pieces of
// equal bit length, one lane each (16 pieces for up to 4 queued streams, 8 / 4 / 2 for up to 8 / 16 / 32):
//   pass 0  a lane starts SPLIT_RUNUP bits before its piece and walks to the first code boundary inside it: its
//           presumed start;
//   pass 1  from there it walks its piece to the first boundary inside the next one, counting symbols;
//   check   a lane's end must be the next lane's start; a lane for which it is not walks again from the right bit
//           (repeated until all agree: lane 0 starts at a known bit, so this ends, normally at once);
//   then the counts give every piece its place in the output and the pieces are decoded like 64 short streams.
// Nothing is trusted: the pieces must chain from the end mark to the first bit of the stream and their symbol
// counts must add up to the stream's regenerated size, else the streams are decoded the ordinary way (which also
// is what reports a corrupt stream).
constexpr uint32_t LANE_COPY_MAX = 256;   // sequence execution: longer literal runs and matches are moved by the whole wave
constexpr uint32_t SPLIT_MIN_PIECE = 64;    // bytes: shorter pieces are not worth the extra passes
constexpr uint32_t SPLIT_RUNUP = 768;       // bits

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// all lanes.  No output: walks from c0 bits consumed to the first code boundary at or beyond `stop`.
// Returns {that boundary, symbols walked}; boundary 0xFFFFFFFF = the table has a hole (never for a valid tree).
__device__ __noinline__ uint2 huf_dry_walk(const uint8_t* stream, uint32_t nbytes, bool act, uint32_t c0, uint32_t stop, uint32_t tab, int lane)
{
    gcu8* p = (gcu8*)stream;
    const uint32_t sL = 32u - (tab >> 16);
    const uint16_t* T = &L.huf[0][0] + (tab & 0xFFFF);
    uint32_t* ring = &L.u.inbuf[0][0] + lane;
    if (!act) {
        c0 = 1;
        stop = 0;
        nbytes = 0;
    }
    int32_t n = -(int32_t)c0;
    uint32_t widx = ((c0 - 1u) >> 5) & ~(uint32_t)(BATCH - 1);
    uint32_t nextbyte = nbytes - 4u * widx;
    uint32_t pend[BATCH];
#define RING_PUT()                                                              \
    do {                                                                        \
        const uint32_t wb__ = (uint32_t)RING - (widx & (uint32_t)(RING - 1));   \
        _Pragma("unroll") for (int k = 0; k < BATCH; ++k) ring[(wb__ - (uint32_t)k) * WAVE] = pend[k]; \
        if (wb__ == (uint32_t)RING) ring[0] = pend[0];                          \
        widx += BATCH;                                                          \
    } while (0)
    for (int f = 0; f < 2; ++f) {
        fetch_batch(p, nextbyte, pend);
        RING_PUT();
    }
    fetch_batch(p, nextbyte, pend);
    uint32_t m = 0;
    bool done = c0 >= stop;
    // a pair of symbols eats at least two bits: more trips than this means a code of length 0
    uint32_t trips = wave_max_u32(done ? 0u : stop - c0) / 32u + 2u;
    while (__any(!done)) {
        if (trips-- == 0) return make_uint2(0xFFFFFFFFu, 0u);
        if (widx - (((uint32_t)~n) >> 5) <= (uint32_t)(RING - BATCH)) {
            RING_PUT();
            fetch_batch(p, nextbyte, pend);
        }
#pragma unroll 4
        for (int q = 0; q < PERIOD / 2; ++q) {
            const int32_t t = n >> 5;
            const uint32_t a = ring[((((uint32_t)t) & (uint32_t)(RING - 1)) + 1u) * WAVE];
            const uint32_t b = ring[(((uint32_t)t) & (uint32_t)(RING - 1)) * WAVE];
            uint32_t x = __builtin_amdgcn_alignbit(a, b, (uint32_t)n);
            const uint32_t l1 = (uint32_t)T[x >> sL] >> 8;
            x <<= l1;
            const uint32_t l2 = (uint32_t)T[x >> sL] >> 8;
            const uint32_t c1 = (uint32_t)-n + l1, c2 = c1 + l2;
            const bool one = c1 >= stop;  // the first symbol already reaches the stop
            if (!done) {
                n = -(int32_t)(one ? c1 : c2);
                m += one ? 1u : 2u;
                done = one || c2 >= stop;
            }
        }
    }
#undef RING_PUT
    return make_uint2((uint32_t)-n, m);
}

// all lanes.  Turns up to four queued long streams into 64 pieces (see above).  Returns false (tasks untouched) if the
// queue does not qualify or the pieces do not fit together.
__device__ __noinline__ bool huf_split_plan(const uint8_t* src, uint32_t& ntask, int lane)
{
    const uint32_t nt = ntask;
    if (nt == 0 || nt > 32) return false;
    // 1-4 streams: 16 pieces each; up to 8: 8; up to 16: 4; up to 32: 2 (reads of several blocks queue 4 streams a block)
    const uint32_t gs = nt <= 4 ? 4u : (nt <= 8 ? 3u : (nt <= 16 ? 2u : 1u));
    const uint32_t G = 1u << gs;
    const uint32_t t = (uint32_t)lane >> gs, j = (uint32_t)lane & (G - 1u);
    const bool act = t < nt;
    uint32_t so = 0, nbytes = 0, out = 0, cnt = 0, tab = 0;
    if (act) {
        so = L.t_src[t];
        nbytes = L.t_size[t];
        out = L.t_out[t];
        cnt = L.t_cnt[t];
        tab = L.t_tab[t];
    }
    if (__any(act && nbytes < SPLIT_MIN_PIECE * G)) return false;
    const uint8_t* p = src + so;
    const uint32_t last = act ? p[nbytes - 1] : 1u;
    if (__any(last == 0)) return false;
    const uint32_t pad = 8u - (uint32_t)hbit(last), B = 8u * nbytes;
    const uint32_t seg = (B - pad + G - 1u) >> gs;
    const uint32_t c_lo = pad + j * seg, c_hi = j == G - 1u ? B : pad + (j + 1u) * seg;
    // pass 0: where this piece presumably starts
    uint32_t s_bit = pad;
    {
        const bool run = act && j != 0;
        const uint32_t from = (c_lo - pad <= SPLIT_RUNUP) ? pad : c_lo - SPLIT_RUNUP;
        const uint2 r = huf_dry_walk(p, nbytes, run, from, c_lo, tab, lane);
        if (__any(run && r.x == 0xFFFFFFFFu)) return false;
        if (run) s_bit = r.x;
    }
    // pass 1 and the repair rounds
    uint32_t e_bit = 0, m = 0;
    bool need = act;
    for (int round = 0; round < 18; ++round) {
        if (!__any(need)) break;
#ifdef VBZ_SPLIT_DEBUG
        {
            const uint32_t nn = (uint32_t)__popcll(__ballot(need));
            if (lane == 0) {
                split_dbg[0] += 1;
                split_dbg[1] += nn;
            }
        }
#endif
        const uint2 r = huf_dry_walk(p, nbytes, need, s_bit, c_hi, tab, lane);
        if (__any(need && r.x == 0xFFFFFFFFu)) return false;
        if (need) {
            e_bit = r.x;
            m = r.y;
        }
        const uint32_t e_prev = (uint32_t)__shfl_up((int)e_bit, 1, 64);
        need = act && j != 0 && e_prev != s_bit;
        if (need) s_bit = e_prev;
    }
    if (__any(need)) return false;
    // the pieces chain; do they cover the stream, symbol for symbol?
    const uint32_t incl = wave_incl_scan_u32(act ? m : 0u);
    const uint32_t before = (uint32_t)__shfl((int)incl, t ? (int)(t * G) - 1 : 0, 64) * (t ? 1u : 0u);
    const uint32_t total = (uint32_t)__shfl((int)incl, (int)((t * G + G - 1u) & 63u), 64) - before;
    if (__any(act && (total != cnt || (j == G - 1u && e_bit != B)))) return false;
    wave_lds_sync();
    L.t_src[lane] = so;
    L.t_size[lane] = nbytes;
    L.t_out[lane] = out + (incl - m - before);
    L.t_cnt[lane] = act ? m : 0u;
    L.t_tab[lane] = tab;
    L.t_bit[lane] = act ? s_bit : 1u;
    L.t_end[lane] = act ? e_bit : 1u;
    wave_lds_sync();
    ntask = WAVE;
    return true;
}

// all lanes: decode the queued Huffman streams.  keep_fse: the sequence tables in LDS (which the rings overwrite) are
// still needed -- a libzstd frame whose next block may say Repeat_Mode -- so they spend the flush in registers.
// Returns true if any stream is corrupt.
__device__ __noinline__ bool flush_tasks(const uint8_t* src, uint8_t* dst, uint32_t& ntask, bool keep_fse, int lane)
{
    if (ntask == 0) {   // (a block whose literals stood already)
        __syncthreads();
        return false;
    }
    constexpr int KEEP = 3 * FSE_SLOT / WAVE;
    uint32_t keep[KEEP];
    uint32_t* f = &L.u.p.fse[0][0];
    if (keep_fse) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) keep[k] = f[k * WAVE + lane];
        wave_lds_sync();
    }
    const bool split = huf_split_plan(src, ntask, lane);
    const bool bad = flush_tasks_ring(src, dst, ntask, split, lane);
    if (keep_fse) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) f[k * WAVE + lane] = keep[k];
        __syncthreads();
    }
    return bad;
}

// ---- byte movers for sequence execution ----------------------------------------------------------------
// Literal runs and matches are a few to a few thousand bytes at arbitrary addresses.  A loop of load-then-store is a
// chain of memory round trips (about a microsecond each); these helpers issue all the loads of a batch before its
// stores, move 16 bytes per instruction whatever the alignment, and never touch a byte outside [0, n).
typedef __attribute__((address_space(1), aligned(1))) const u32x4 gld16;
typedef __attribute__((address_space(1), aligned(1))) u32x4 gst16;
typedef __attribute__((address_space(1), aligned(1))) const uint32_t gld4;
typedef __attribute__((address_space(1), aligned(1))) uint32_t gst4;
typedef __attribute__((address_space(1), aligned(1))) const uint16_t gld2;
typedef __attribute__((address_space(1), aligned(1))) uint16_t gst2;

// one lane: n bytes from f to o.  The source of a batch (up to 79 bytes) must not be written by the same batch.
__device__ __forceinline__ void lane_copy(gu8* o, gcu8* f, uint32_t n)
{
    uint32_t k = 0;
    while (n - k >= 64) {
        const u32x4 a = *(gld16*)(f + k), b = *(gld16*)(f + k + 16), c = *(gld16*)(f + k + 32), d = *(gld16*)(f + k + 48);
        *(gst16*)(o + k) = a;
        *(gst16*)(o + k + 16) = b;
        *(gst16*)(o + k + 32) = c;
        *(gst16*)(o + k + 48) = d;
        k += 64;
    }
    const uint32_t r = n - k;
    if (n >= 16) {
        if (r) {  // whole vectors, then one that ends exactly at n (it may overlap the one before: same bytes)
            u32x4 a = {}, b = {}, c = {};
            if (r > 16) a = *(gld16*)(f + k);
            if (r > 32) b = *(gld16*)(f + k + 16);
            if (r > 48) c = *(gld16*)(f + k + 32);
            const u32x4 z = *(gld16*)(f + n - 16);
            if (r > 16) *(gst16*)(o + k) = a;
            if (r > 32) *(gst16*)(o + k + 16) = b;
            if (r > 48) *(gst16*)(o + k + 32) = c;
            *(gst16*)(o + n - 16) = z;
        }
    } else if (n) {  // 8 + 4 + 2 + 1
        const uint32_t p4 = n & 8u, p2 = n & 12u, p1 = n & 14u;
        uint32_t a0 = 0, a1 = 0, b = 0, c = 0, d = 0;
        if (n & 8u) {
            a0 = *(gld4*)f;
            a1 = *(gld4*)(f + 4);
        }
        if (n & 4u) b = *(gld4*)(f + p4);
        if (n & 2u) c = *(gld2*)(f + p2);
        if (n & 1u) d = f[p1];
        if (n & 8u) {
            *(gst4*)o = a0;
            *(gst4*)(o + 4) = a1;
        }
        if (n & 4u) *(gst4*)(o + p4) = b;
        if (n & 2u) *(gst2*)(o + p2) = (uint16_t)c;
        if (n & 1u) o[p1] = (uint8_t)d;
    }
}

// all lanes: n bytes from f to o, the regions do not overlap
__device__ __forceinline__ void wave_copy(gu8* o, gcu8* f, uint32_t n, int lane)
{
    uint32_t k = 16u * (uint32_t)lane;
    while (k + 3u * 1024u + 16u <= n) {
        const u32x4 a = *(gld16*)(f + k), b = *(gld16*)(f + k + 1024), c = *(gld16*)(f + k + 2048), d = *(gld16*)(f + k + 3072);
        *(gst16*)(o + k) = a;
        *(gst16*)(o + k + 1024) = b;
        *(gst16*)(o + k + 2048) = c;
        *(gst16*)(o + k + 3072) = d;
        k += 4096;
    }
    for (; k + 16u <= n; k += 1024u) *(gst16*)(o + k) = *(gld16*)(f + k);
    const uint32_t r = n & 15u;
    if ((uint32_t)lane < r) o[n - r + (uint32_t)lane] = f[n - r + (uint32_t)lane];
}

// all lanes: the n bytes at o repeat what stands `off` bytes in front of them (n > off >= 16: a match longer than its offset).  Bytes
// [o - off, o + done) are final and periodic in off, so they may be copied from any multiple of off back: the distance doubles with
// every round (a 128 KB match at offset 257 -- what iota of int8 is made of -- took ONE lane 2 000 dependent trips of 64 bytes, 3.5 ms;
// now nine rounds of the whole wavefront).  The barrier drains the round's stores: they are the next round's source.
__device__ __forceinline__ void wave_copy_repeat(gu8* o, uint32_t off, uint32_t n, int lane)
{
    uint32_t done = 0, d = off;
    while (done < n) {
        const uint32_t c = n - done < d ? n - done : d;
        wave_copy(o + done, (gcu8*)(o + done) - d, c, lane);
        __syncthreads();
        done += c;
        d *= 2u;   // (done = d - off now: d + d <= off + done + d ... the next source [o + done - 2 d', ...) starts at or behind o - off)
    }
}

// 16 bytes of the period-`off` pattern that starts at f (off = 1, 2, 4, 8 or 16; f[0 .. 15] must be readable)
__device__ __forceinline__ u32x4 period_expand(u32x4 v, uint32_t off);
__device__ __forceinline__ u32x4 period_pattern(gcu8* f, uint32_t off)
{
    return period_expand(*(gld16*)f, off);
}
// ... from its first `off` bytes in v
__device__ __forceinline__ u32x4 period_expand(u32x4 v, uint32_t off)
{
    if (off == 1) v.x = (v.x & 0xFFu) * 0x01010101u;
    if (off == 2) v.x = (v.x & 0xFFFFu) * 0x00010001u;
    if (off <= 4) v.y = v.x;
    if (off <= 8) {
        v.z = v.x;
        v.w = v.y;
    }
    return v;
}

// one lane: n bytes of the pattern v (phase 0 at o)
__device__ __forceinline__ void lane_fill(gu8* o, u32x4 v, uint32_t n)
{
    uint32_t k = 0;
    for (; k + 16u <= n; k += 16u) *(gst16*)(o + k) = v;
    if (n & 8u) {
        *(gst4*)(o + k) = v.x;
        *(gst4*)(o + k + 4) = v.y;
        v.x = v.z;
        v.y = v.w;
        k += 8;
    }
    if (n & 4u) {
        *(gst4*)(o + k) = v.x;
        v.x = v.y;
        k += 4;
    }
    if (n & 2u) {
        *(gst2*)(o + k) = (uint16_t)v.x;
        v.x >>= 16;
        k += 2;
    }
    if (n & 1u) o[k] = (uint8_t)v.x;
}

// all lanes: n bytes of the pattern v (phase 0 at o)
__device__ __forceinline__ void wave_fill(gu8* o, u32x4 v, uint32_t n, int lane)
{
    for (uint32_t k = 16u * (uint32_t)lane; k + 16u <= n; k += 1024u) *(gst16*)(o + k) = v;
    const uint32_t r = n & 15u;
    if ((uint32_t)lane < r) {
        const uint32_t w = (lane & 8) ? ((lane & 4) ? v.w : v.z) : ((lane & 4) ? v.y : v.x);
        o[n - r + (uint32_t)lane] = (uint8_t)(w >> (8 * (lane & 3)));
    }
}

// all lanes.  The sequences of a block with arbitrary offsets (libzstd's frames), decoded -- not executed -- into 16-byte
// records {literal length, match length, offset, 0} at `rec`, fully validated (bit stream, literal budget, offsets
// inside the output, block and frame size).  The three state machines are one dependent chain: wave-uniform code on
// the scalar unit, FSE tables in LDS (read at a uniform address), the backward bit stream in a 256-byte register
// window (lane j = the j-th dword from the end, reloaded with one coalesced load per 2048 bits), 64 records collected
// across the lanes before they leave with one store.  rep[] = the three repeat offsets (in and out).
// Returns false if the section is corrupt.
__device__ __noinline__ bool general_sequence_records(const uint8_t* bs_, uint32_t bsn_, uint4* rec, uint32_t nseq_, uint32_t log_ll_,
                                                      uint32_t log_of_, uint32_t log_ml_, uint32_t regen_, uint32_t opos_, uint32_t fcs_,
                                                      uint32_t (&rep)[3], int lane)
{
    const uint32_t bsn = uni(bsn_), nseq = uni(nseq_), log_ll = uni(log_ll_), log_of = uni(log_of_), log_ml = uni(log_ml_);
    const uint32_t regen = uni(regen_), opos0 = uni(opos_), fcs = uni(fcs_);
    uint32_t rep0 = uni(rep[0]), rep1 = uni(rep[1]), rep2 = uni(rep[2]);
    const uint8_t* bs = reinterpret_cast<const uint8_t*>(((uint64_t)uni((uint32_t)((uint64_t)bs_ >> 32)) << 32) |
                                                         uni((uint32_t)(uint64_t)bs_));
    if (bsn == 0) return false;
    // code -> baseline | extra bits << 24, one code per lane (read with v_readlane: no memory on the chain)
    const uint32_t llx = lane < 36 ? LL_BASE[lane] | ((uint32_t)LL_BITS[lane] << 24) : 0u;
    const uint32_t mlx = lane < 53 ? ML_BASE[lane] | ((uint32_t)ML_BITS[lane] << 24) : 0u;
    // the window: lane j holds dword k0 + j, counted from the end of the stream (beyond the start: zeros).  It is
    // moved only at the top of the loop (a sequence eats at most 89 bits), so the reads inside never miss.  There is no
    // bit buffer: `pos` bits are consumed, the next ones are {D[pos / 32], D[pos / 32 + 1]} << pos % 32.
    auto load_window = [&](uint32_t k0) -> uint32_t {
        const int64_t off = (int64_t)bsn - 4 * (int64_t)(k0 + (uint32_t)lane + 1);
        uint32_t v = 0;
        if (off >= 0) {
            __builtin_memcpy(&v, bs + off, 4);
        } else if (off > -4) {
            for (int b = 0; b < 4 + (int)off; ++b) v |= (uint32_t)bs[b] << (8 * (b - (int)off));
        }
        return v;
    };
    uint32_t k0 = 0;
    uint32_t win = load_window(0);
    const uint32_t top = lane_get(win, 0) >> 24;
    if (top == 0) return false;
    uint32_t pos = 8u - (uint32_t)hbit(top);
    auto take = [&](uint32_t nb) -> uint32_t {  // nb <= 32
        const uint32_t k = (pos >> 5) - k0;
        const uint64_t two = ((uint64_t)lane_get(win, k) << 32) | lane_get(win, k + 1);
        const uint32_t v = (uint32_t)(((two << (pos & 31u)) >> 1) >> (63 - nb));
        pos += nb;
        return v;
    };
    uint32_t sl = take(log_ll), so = take(log_of), sm = take(log_ml);
    // checks that cannot lead a read or write astray are collected and looked at once, behind the loop
    uint64_t sum_ll = 0, outp = opos0;
    uint32_t worst_l = 0, worst_o = 0, worst_m = 0, astray = 0;
    uint32_t r_ll = 0, r_ml = 0, r_of = 0;
    for (uint32_t i = 0; i < nseq; ++i) {
        if ((pos >> 5) - k0 > 58) {
            k0 = pos >> 5;
            win = load_window(k0);
        }
        const uint32_t el = uni(L.u.p.fse[0][sl]), eo = uni(L.u.p.fse[1][so]), em = uni(L.u.p.fse[2][sm]);
        const uint32_t lc = el & 0xFF, oc = eo & 0xFF, mc = em & 0xFF;
        worst_l = lc > worst_l ? lc : worst_l;
        worst_o = oc > worst_o ? oc : worst_o;
        worst_m = mc > worst_m ? mc : worst_m;
        const uint32_t lx = lane_get(llx, lc & 63u), mx = lane_get(mlx, mc & 63u);
        const uint32_t lb = lx >> 24, mb = mx >> 24;
        const uint32_t ofv = (1u << (oc & 31u)) + take(oc & 31u);  // RFC 8878 3.1.1.3.2.1.1: offset bits first,
        const uint32_t ex = take(mb + lb);                          // then match length, then literal length
        const uint32_t mlen = (mx & 0xFFFFFFu) + (ex >> lb);
        const uint32_t llen = (lx & 0xFFFFFFu) + (ex & ((1u << lb) - 1u));
        // repeat offsets (3.1.1.5), branch free: idx 0 = rep0 as it is, 1 = rep1, 2 = rep2, 3 = rep0 - 1
        const bool isrep = ofv <= 3;
        const uint32_t idx = ofv - 1 + (llen == 0 ? 1u : 0u);
        uint32_t cand = idx == 1 ? rep1 : (idx == 2 ? rep2 : rep0 - (idx == 3 ? 1u : 0u));
        cand = cand ? cand : 1u;  // libzstd forces an invalid 0 to 1
        const uint32_t offset = isrep ? cand : ofv - 3;
        rep2 = (isrep && idx <= 1) ? rep2 : rep1;
        rep1 = (isrep && idx == 0) ? rep1 : rep0;
        rep0 = offset;
        {  // the three states move on: LL, ML, OF (at most 9 + 9 + 8 bits; the last sequence reads none)
            const bool last = i + 1 == nseq;
            const uint32_t nl = last ? 0u : (el >> 8) & 0xFF, nm = last ? 0u : (em >> 8) & 0xFF, no = last ? 0u : (eo >> 8) & 0xFF;
            const uint32_t v = take(nl + nm + no);
            sl = (el >> 16) + (v >> (nm + no));
            sm = (em >> 16) + ((v >> no) & ((1u << nm) - 1u));
            so = (eo >> 16) + (v & ((1u << no) - 1u));
        }
        sum_ll += llen;
        outp += llen;
        astray += offset > outp ? 1u : 0u;
        outp += mlen;
        const bool mine = (uint32_t)lane == (i & 63);
        r_ll = mine ? llen : r_ll;
        r_ml = mine ? mlen : r_ml;
        r_of = mine ? offset : r_of;
        if ((i & 63) == 63) rec[(i & ~63u) + (uint32_t)lane] = make_uint4(r_ll, r_ml, r_of, 0u);
    }
    if ((uint32_t)lane < (nseq & 63)) rec[(nseq & ~63u) + (uint32_t)lane] = make_uint4(r_ll, r_ml, r_of, 0u);
    if (worst_l > 35 || worst_m > 52 || worst_o > 31 || astray) return false;
    if (sum_ll > regen || outp > fcs || outp - opos0 > BLOCK_MAX) return false;  // both only grow
    if (pos != 8u * bsn) return false;  // every bit consumed, none beyond
    rep[0] = rep0;
    rep[1] = rep1;
    rep[2] = rep2;
    return true;
}

// The same walk with the three state machines on three LANES (lane 0: offsets, lane 1: match lengths, lane 2: literal
// lengths) instead of one after the other on the scalar unit: one LDS read fetches the three table entries, the places of
// the six bit fields of a sequence -- three groups of extra bits, three groups of state bits -- come from two short DPP
// scans, and every lane cuts its two fields out of a copy of the window in LDS.  Half the instructions per sequence of the
// scalar walk (a lone wavefront: 950 -> ~500 cycles per sequence); the repeat-offset rules and the checks are the scalar
// walk's, on the three values read back from the lanes.  Works in LDS that is idle at this point: the extra-bit count of
// every state in hbuf / norm / symnext, the code -> baseline tables in wfse / weights, the window in the task arrays.
__device__ __noinline__ bool general_sequence_records_lanes(const uint8_t* bs_, uint32_t bsn_, uint4* rec, uint32_t nseq_, uint32_t log_ll_,
                                                            uint32_t log_of_, uint32_t log_ml_, uint32_t regen_, uint32_t opos_, uint32_t fcs_,
                                                            uint32_t (&rep)[3], int lane)
{
    static_assert(HBUF + 512 + 512 >= 3 * FSE_SLOT, "room for one byte per state behind the sequence tables");
    const uint32_t bsn = uni(bsn_), nseq = uni(nseq_), log_ll = uni(log_ll_), log_of = uni(log_of_), log_ml = uni(log_ml_);
    const uint32_t regen = uni(regen_), opos0 = uni(opos_), fcs = uni(fcs_);
    // (the repeat offsets and the sums live in vector registers, the same value in every lane: as scalar code the rules below
    // come out as a chain of compares and BRANCHES, which a lone wavefront pays for with a refetch each)
    // (the caller's repeat offsets are lane 0's: the other lanes' copies may be stale)
    uint32_t rep0 = uni(rep[0]), rep1 = uni(rep[1]), rep2 = uni(rep[2]);
    const uint8_t* bs = reinterpret_cast<const uint8_t*>(((uint64_t)uni((uint32_t)((uint64_t)bs_ >> 32)) << 32) |
                                                         uni((uint32_t)(uint64_t)bs_));
    if (bsn == 0) return false;
    uint8_t* NB = L.u.p.hbuf;          // [3][FSE_SLOT]: extra bits of the code of every state (role order: OF, ML, LL)
    uint32_t* XT = L.wfse;             // [2][64]: code -> baseline | extra bits << 24 (ML, LL)
    uint32_t* WIN = L.t_src;           // the window: WIN[j] = dword k0 + j counted from the end of the stream
    // role of this lane: 0 offsets, 1 match lengths, 2 literal lengths (lanes >= 3 run along with empty fields)
    const uint32_t role = (uint32_t)lane < 3u ? (uint32_t)lane : 0u;
    const bool live = (uint32_t)lane < 3u;
    const uint32_t tsel = role == 0 ? 1u : (role == 1 ? 2u : 0u);   // the role's table in L.u.p.fse (LL, OF, ML)
    wave_lds_sync();
    XT[lane] = lane < 53 ? ML_BASE[lane] | ((uint32_t)ML_BITS[lane] << 24) : 0u;
    XT[64 + lane] = lane < 36 ? LL_BASE[lane] | ((uint32_t)LL_BITS[lane] << 24) : 0u;
    wave_lds_sync();
    for (uint32_t s0 = (uint32_t)lane; s0 < (uint32_t)FSE_SLOT; s0 += WAVE) {
        NB[s0] = (uint8_t)(L.u.p.fse[1][s0] & 31u);
        NB[FSE_SLOT + s0] = (uint8_t)(XT[L.u.p.fse[2][s0] & 63u] >> 24);
        NB[2 * FSE_SLOT + s0] = (uint8_t)(XT[64 + (L.u.p.fse[0][s0] & 63u)] >> 24);
    }
    auto load_window = [&](uint32_t k0) -> uint32_t {
        const int64_t off = (int64_t)bsn - 4 * (int64_t)(k0 + (uint32_t)lane + 1);
        uint32_t v = 0;
        if (off >= 0) {
            __builtin_memcpy(&v, bs + off, 4);
        } else if (off > -4) {
            for (int b = 0; b < 4 + (int)off; ++b) v |= (uint32_t)bs[b] << (8 * (b - (int)off));
        }
        return v;
    };
    uint32_t k0 = 0;
    WIN[lane] = load_window(0);
    if (lane < 8) WIN[64 + lane] = 0;
    wave_lds_sync();
    const uint32_t top = uni(WIN[0]) >> 24;
    if (top == 0) return false;
    uint32_t pos = 8u - (uint32_t)hbit(top);   // bits consumed (wave-uniform)
    // bits [P, P + w) of the stream, w <= 32 (per lane)
    auto field = [&](uint32_t P, uint32_t w) -> uint32_t {
        const uint32_t k = (P >> 5) - k0;
        const uint64_t two = ((uint64_t)WIN[k] << 32) | WIN[k + 1];
        return (uint32_t)(((two << (P & 31u)) >> 1) >> (63u - w));
    };
    // the initial states: LL, OF, ML in that order
    uint32_t st;
    {
        const uint32_t at = role == 2 ? 0u : (role == 0 ? log_ll : log_ll + log_of);
        st = field(pos + at, role == 2 ? log_ll : (role == 0 ? log_of : log_ml));
        pos += log_ll + log_of + log_ml;
    }
    uint64_t sum_ll = 0, outp = opos0;
    uint32_t worst = 0;
    bool astray = false;
    uint32_t r_ll = 0, r_ml = 0, r_of = 0;
    const uint32_t tbase = tsel * (uint32_t)FSE_SLOT, nbase_role = role * (uint32_t)FSE_SLOT;
    const uint32_t* FSE = &L.u.p.fse[0][0];
    const uint32_t xt_role = role == 2 ? 64u : 0u;
    // the table entry and the extra-bit count of the current state are requested as soon as the state is known, one
    // sequence ahead of their use: the LDS round trip runs behind the repeat-offset rules of the sequence before
    uint32_t e = FSE[tbase + st], a_raw = NB[nbase_role + st];
    for (uint32_t i = 0; i < nseq; ++i) {
        if ((pos >> 5) - k0 > 58) {
            k0 = pos >> 5;
            wave_lds_sync();
            WIN[lane] = load_window(k0);
            wave_lds_sync();
        }
        const bool last = i + 1 == nseq;
        uint32_t a = a_raw;
        const uint32_t code = e & 0xFFu;
        uint32_t nb = last ? 0u : (e >> 8) & 0xFFu;
        a = live ? a : 0u;
        nb = live ? nb : 0u;
        worst = (live && code > worst) ? code : worst;
        // where the lane's extra bits and state bits stand: extras in the order OF, ML, LL, then the states LL, ML, OF
        const uint32_t x1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x111, 0xF, 0xF, false);   // row_shr:1
        const uint32_t x2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x112, 0xF, 0xF, false);   // row_shr:2
        const uint32_t xoff = x1 + x2;
        const uint32_t S = lane_get(xoff + a, 2);
        const uint32_t s1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nb, 0x101, 0xF, 0xF, false);  // row_shl:1
        const uint32_t s2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nb, 0x102, 0xF, 0xF, false);  // row_shl:2
        const uint32_t soff = S + s1 + s2;
        const uint32_t total = lane_get(soff + nb, 0);
        const uint32_t extra = field(pos + xoff, a);
        const uint32_t sbits = field(pos + soff, nb);
        st = live ? (e >> 16) + sbits : 0u;
        e = FSE[tbase + st];
        a_raw = NB[nbase_role + st];
        const uint32_t xb = XT[xt_role + (code & 63u)] & 0xFFFFFFu;            // (read by the offsets' lane too: no branch)
        const uint32_t pw = 1u << (code & 31u);
        const uint32_t val = (role == 0 ? pw : xb) + extra;
        pos += total;
        // the three values to every lane through the LDS crossbar (results the compiler takes for per-lane values)
        const uint32_t ofv = (uint32_t)__builtin_amdgcn_ds_bpermute(0, (int)val), mlen = (uint32_t)__builtin_amdgcn_ds_bpermute(4, (int)val),
                       llen = (uint32_t)__builtin_amdgcn_ds_bpermute(8, (int)val);
        // repeat offsets (3.1.1.5), branch free: idx 0 = rep0 as it is, 1 = rep1, 2 = rep2, 3 = rep0 - 1
        // (written with masks, not with conditions: no control flow on the chain)
        const uint32_t isrep = 0u - (uint32_t)(ofv <= 3);                       // all ones / zero
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
        astray = astray || offset > outp;
        outp += mlen;
        const bool mine = (uint32_t)lane == (i & 63);
        r_ll = mine ? llen : r_ll;
        r_ml = mine ? mlen : r_ml;
        r_of = mine ? offset : r_of;
        if ((i & 63) == 63) rec[(i & ~63u) + (uint32_t)lane] = make_uint4(r_ll, r_ml, r_of, 0u);
    }
    if ((uint32_t)lane < (nseq & 63)) rec[(nseq & ~63u) + (uint32_t)lane] = make_uint4(r_ll, r_ml, r_of, 0u);
    const uint32_t worst_o = lane_get(worst, 0), worst_m = lane_get(worst, 1), worst_l = lane_get(worst, 2);
    wave_lds_sync();
    if (worst_l > 35 || worst_m > 52 || worst_o > 31 || __any(astray)) return false;
    if (__any(sum_ll > regen || outp > fcs || outp - opos0 > BLOCK_MAX)) return false;  // both only grow
    if (pos != 8u * bsn) return false;  // every bit consumed, none beyond
    rep[0] = uni(rep0);
    rep[1] = uni(rep1);
    rep[2] = uni(rep2);
    return true;
}

__device__ __forceinline__ void stage_bytes(uint8_t* lds, const uint8_t* g, uint32_t n, int lane)
{
    for (uint32_t i = lane; i < n; i += WAVE) lds[i] = g[i];
    wave_lds_sync();
}

#ifndef VBZ_DEC_WAVES
#define VBZ_DEC_WAVES 1
#endif
// Span mode (batches of few, large reads): one wavefront decodes one SPAN of a frame -- a run of blocks that begins with a
// block carrying its own Huffman tree -- as announced by the index trailer zstd_encode.hip writes behind large frames.
// The index is verified, never trusted: a span must end exactly where the next one begins (in the frame and in the
// content), spans other than the first may not lean on an earlier Huffman table and may contain sequences only of the
// zero-run kind (offset = repeat offset 1 = 1 throughout, which every span confirms for itself; such a match never reads
// a byte outside its own block); if anything is off, the frame is decoded again by one wavefront the ordinary way, which
// is also what decides every error verdict.
struct DecSpan
{
    uint32_t read;
    uint32_t src_pos;   // first block of the span (offset in the read's source)
    uint32_t src_end;   // where the span must end (the next span's first block); unused for the last span
    uint32_t dst_pos;   // content offset of the span's first byte
    uint32_t flags;     // DSPAN_*
    uint32_t ord;       // ordinal of the span in its frame
    uint32_t tree_pos;  // != 0: the block (offset in the read's source) whose Huffman tree is in force where this span begins
    uint32_t dst_len;   // content bytes of the span (what the index says)
};
constexpr uint32_t DSPAN_WHOLE = 1, DSPAN_LAST = 2, DSPAN_FIRST = 4;
// Shared tables (zstd_encode.hip, round 5): the data bytes of a large read are spans of ONE block each, the first with the region's tree
// description, the others treeless.  The plan kernel looks at the first block of every span (does it bring a tree?) and gives the
// spans behind the LAST such span T that block as tree_pos: a span with a tree_pos reads that tree before its own first block.  Like
// the index itself this is verified, not trusted -- every span reports whether its first block brought a tree and whether a later
// block did (status word 3: DSPAN_ST_*), and a frame stands only if T brought its tree in its first block and nowhere else, and no
// span behind T brought any: then the table in force at every span behind T is T's, by the frame's own rules.
constexpr uint32_t DSPAN_ST_REP1 = 1, DSPAN_ST_TREE_FIRST = 2, DSPAN_ST_TREE_LATER = 4;
constexpr uint32_t IDX_MAGIC = 0x184D2A5Cu;      // zstd_encode.hip: the span index trailer
constexpr uint32_t DSPAN_MIN_CONTENT = 4u << 10;    // an honest index has at most fcs / this + 4 spans (the writer's shortest spans hold 4 KB of control bytes)
// A span whose block carries zero-run sequences stages its literals and (literal, match) length pairs behind the frame's
// content in the destination slot, in a stripe of its own: DSPAN_WS_FACTOR bytes per byte of its content (literals <= content, eight
// bytes per sequence and a sequence regenerates four bytes or more -- what does not fit is decoded by the ordinary decoder) + 32,
// laid out by content position: the stripes of different spans cannot overlap.  (Round 4 gave every span 64 KB by its ordinal, which
// ran out of the slot from the fifth span with sequences on.)
constexpr uint32_t DSPAN_WS_FACTOR = 3;
// TIMED: per-phase shader-clock counters (VBZ_HIP_PHASE_TIMING); a separate instantiation, the counters cost
// dozens of registers in the production kernel otherwise
// FUSED: the frame's content is the svb stream of int16 zig-zag samples and b.dst its slot in the library's scratch: once the
// frame is decoded the same wave decodes the stream (svb_wave.h) into the read's final destination (fuse.out) and
// b.result[] gets the FINAL verdict -- the stream is read back while it is still in the caches, and no svb_decode launch
// follows.
struct SvbFuse
{
    uint8_t* out;
    const uint64_t* out_off;
    const uint32_t* out_size;   // exact decoded byte count of every read
};
// SPANS: the span-mode instantiation (dspans != nullptr; the other one carries none of its code: the ordinary decoder's register
// allocation is not the spans' business)
template <bool TIMED, bool FUSED, bool SPANS = false>
__global__ __launch_bounds__(WAVE, VBZ_DEC_WAVES) void zstd_decode_kernel(ReadBatch b, uint32_t toosmall_code, unsigned long long* dbg, const SeqDTables* dtabs,
                                                                      const DecSpan* dspans_, const uint32_t* dspan_count, uint32_t* dspan_status,
                                                                      const uint32_t* only, SvbFuse fuse, RefChains chains)
{
    const DecSpan* const dspans = SPANS ? dspans_ : nullptr;
    unsigned long long tph[PHASE_SLOTS] = {};
    unsigned long long tlast = TIMED ? __builtin_readcyclecounter() : 0;
#define PHASE(k) do { if (TIMED) { unsigned long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; } } while (0)
    const int lane = threadIdx.x;
    DecSpan sp = {};
    if (dspans) {
        if (blockIdx.x >= *dspan_count) return;
        sp = dspans[blockIdx.x];
    }
    // partial: this wave decodes one span of the frame and reports to dspan_status instead of the read's result
    const bool partial = SPANS && dspans != nullptr && !(sp.flags & DSPAN_WHOLE);
    const uint32_t r = dspans ? sp.read : blockIdx.x;
    if (only && !only[r]) return;  // second launch of span mode: only the frames whose spans did not work out
    if (b.gate && b.gate[r] >= GATE_SKIP) {
        if (lane == 0 && b.gate[r] != GATE_SKIP) b.result[r] = b.gate[r];
        return;
    }
    const uint8_t* src = b.src + b.src_off[r];
    const uint32_t n = b.src_size[r];
    if (n >= E_FIRST) {
        if (lane == 0) b.result[r] = n;
        return;
    }
    uint8_t* dst = b.dst + b.dst_off[r];
    const uint32_t cap = b.dst_cap[r];
#define FLUSH()                                                                                                       \
    do {                                                                                                              \
        if (flush_tasks(src, dst, ntask, fse_live, lane)) FAIL();                                                     \
        if (d_active) {                                                                                               \
            PHASE(1);                                                                                                 \
            if (place_zero_runs(dst, reinterpret_cast<const uint2*>(dst + d_pairs), d_nseq, d_lit, d_ltype, d_regen,    \
                                d_opos, fcs, block_max, reinterpret_cast<uint8_t*>(&L.u.inbuf[0][0]),                 \
                                (uint32_t)sizeof(L.u.inbuf) - 8u, lane) == 0xFFFFFFFFu)                                \
                FAIL();                                                                                               \
            d_active = false;                                                                                         \
            __syncthreads();                                                                                          \
            PHASE(4);                                                                                                 \
        }                                                                                                             \
    } while (0)
#define FAIL()                                                          \
    do {                                                                \
        if (lane == 0) {                                                \
            if (partial) dspan_status[4 * blockIdx.x] = 2;              \
            else b.result[r] = E_ZSTD;                                  \
        }                                                               \
        return;                                                         \
    } while (0)

#ifdef VBZ_SPLIT_DEBUG
    if (lane == 0) split_dbg[0] = split_dbg[1] = 0;
#endif
    // ---- frame header (RFC 8878 3.1.1.1)
    stage_bytes(L.u.p.hbuf, src, n < 32 ? n : 32, lane);
    if (lane == 0) {
        uint32_t err = 0, pos = 0;
        uint64_t fcs = 0, window = 0;
        const uint8_t* h = L.u.p.hbuf;
        if (n < 6) err = 1;
        if (!err) {
            const uint32_t magic = h[0] | (h[1] << 8) | (h[2] << 16) | ((uint32_t)h[3] << 24);
            const uint32_t fhd = h[4];
            const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
            if (magic != 0xFD2FB528u || (fhd & 0x08)) err = 1;
            pos = 5;
            if (!single) {
                const uint32_t wd = h[pos++];
                const int wlog = 10 + (int)(wd >> 3);
                if (wlog > 31) err = 1;
                window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7);
            }
            // Dictionary_ID field (0, 1, 2 or 4 bytes): the value 0 means "no dictionary" and is accepted like libzstd does;
            // any other dictionary is one this decoder does not have (libzstd: dictionary_wrong), reported behind the
            // content-size check like ZSTD_getFrameContentSize + ZSTD_decompress would (vbz.cpp:236-273)
            bool foreign_dict = false;
            {
                const uint32_t dsz = did_flag == 3 ? 4u : (uint32_t)did_flag;
                if (pos + dsz > n) err = 1;
                else
                    for (uint32_t i = 0; i < dsz; ++i) foreign_dict |= h[pos + i] != 0;
                pos += dsz;
            }
            const int fsz = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
            if (fsz == 0) err = 1;  // unknown content size: the reference rejects it too (vbz.cpp:236-240)
            if (pos + fsz > n) err = 1;
            if (!err) {
                for (int i = 0; i < fsz; ++i) fcs |= (uint64_t)h[pos + i] << (8 * i);
                if (fsz == 2) fcs += 256;
                pos += fsz;
                if (single) window = fcs;
                if (!err && fcs > cap) err = 2;  // dstSize_tooSmall
                if (!err && foreign_dict) err = 1;
            }
            L.ctl[C_A] = pos;
            L.ctl[C_B] = (uint32_t)fcs;
            L.ctl[C_C] = (uint32_t)(window < BLOCK_MAX ? window : BLOCK_MAX);
            L.ctl[C_D] = (fhd >> 2) & 1;
        }
        L.ctl[C_ERR] = err;
    }
    __syncthreads();
    if (L.ctl[C_ERR] == 2) {
        if (partial) FAIL();
        if (lane == 0) b.result[r] = toosmall_code;
        return;
    }
    if (L.ctl[C_ERR]) FAIL();
    uint32_t pos = L.ctl[C_A];
    const uint32_t fcs = L.ctl[C_B];
    const uint32_t block_max = L.ctl[C_C];
    const uint32_t has_checksum = L.ctl[C_D];

    // the encoder's checkpoint trailer (a skippable frame that ends the buffer), if any: see zero_run_chain_segments
    const uint8_t* cp_tab = nullptr;
    uint32_t cp_count = 0, cp_spacing = 0;
    if (n >= 64) {
        uint32_t tb, ne = n;  // ne: where the checkpoint trailer would end (an index trailer may follow it)
        __builtin_memcpy(&tb, src + n - 4, 4);
        if (tb >= 24 && tb <= n - 16 && (tb & 7u) == 0) {  // (n >= 64; no sum that could wrap: these are arbitrary bytes)
            uint32_t m[2];
            __builtin_memcpy(m, src + n - tb, 8);
            if (m[0] == IDX_MAGIC && m[1] == tb - 8) {
                ne = n - tb;
                __builtin_memcpy(&tb, src + ne - 4, 4);
            }
        }
        if (tb >= 20 && tb <= 8 + 4 + 4 * 63 + 4 && tb + 16 <= ne) {
            uint32_t m[3];
            __builtin_memcpy(m, src + ne - tb, 12);
            const uint32_t cnt = m[2] >> 16;
            if (m[0] == 0x184D2A5Bu && m[1] == tb - 8 && tb == 16 + 4 * cnt && cnt >= 1) {
                cp_tab = src + ne - tb + 12;
                cp_count = cnt;
                cp_spacing = m[2] & 0xFFFFu;
            }
        }
    }
    // a frame whose sequence chains are already walked (zstd_decode_ref.hip): its records are taken, nothing else changes
    const RefPre* mypre = (chains.pre != nullptr && !dspans && chains.pre[r].ok) ? chains.pre + r : nullptr;
    const uint32_t first_block = partial ? sp.src_pos : pos;
    const uint32_t first_opos = partial ? sp.dst_pos : 0u;
    uint32_t opos = 0, ntask = 0;
    uint32_t rep0_end = 1;  // repeat offset 1 when the span is done (lane 0)
    uint32_t tree_info = 0; // span mode: DSPAN_ST_TREE_*
    for (int attempt = 0;; ++attempt) {
    if (partial && attempt != 0) FAIL();  // a frame that needs the careful second attempt is not decoded in spans
    // attempt 0 lets the stream decoders' rings reuse the LDS of FSE tables that are (normally) dead; if a later
    // block turns out to repeat such a table, the frame is decoded again with the tables kept (attempt 1)
    bool restart = false;
    const bool use_pre = mypre != nullptr && attempt == 0;  // (anything unexpected about them: the frame again, without)
    uint32_t pre_k = 0;
    pos = first_block;
    opos = first_opos;
    ntask = 0;
    bool huf_valid = false;
    bool fse_live = attempt != 0;  // FSE tables must survive: flush_tasks parks them in registers
    bool tables_built = false, tables_lost = false;
    // one block whose zero-run sequences are already decoded into pairs but not yet placed: its literals are
    // still queued as stream tasks (staged behind the frame), it is finished right after the next flush
    bool d_active = false;
    uint32_t d_opos = 0, d_regen = 0, d_nseq = 0, d_ltype = 0, d_pairs = 0;
    const uint8_t* d_lit = nullptr;
    int cur_slot = 0, cur_log = 0;      // current Huffman table: slot and table log
    tree_info = 0;
    // a span that begins under another block's tree (see DSPAN_ST_*) takes one trip through the block loop at that block first, up
    // to the point where its table stands, and starts over at its own first block
    bool tree_only = SPANS && partial && sp.tree_pos != 0;
    if (tree_only) pos = sp.tree_pos;
    bool have_ll = false, have_of = false, have_ml = false;
    int log_ll = 0, log_of = 0, log_ml = 0;
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;  // lane 0 only
    // the 24 bytes at the head of the next block, requested as soon as this block's size is known (lane i holds
    // byte i): by the time the block is done they have arrived
    uint32_t pf_byte = 0;
    bool pf_ok = false;
    // fast path below: the one-byte sequences section ("no sequences") of a literals-only block is requested when the block is
    // queued and looked at one block later (or when the frame ends): nobody waits for that load
    uint32_t lazy_seq = 0;
    bool cp_avail = cp_count != 0 && (!partial || (sp.flags & DSPAN_FIRST));  // the trailer describes the frame's first sequences section

    for (;;) {
        if (pos + 3 > n) FAIL();
        PHASE(0);
        // 24 staged bytes cover the block header, the literals header and (treeless blocks) the jump table
        if (pf_ok) {  // (one wave: LDS hand-overs need no vmcnt drain, the requests below stay in flight)
            wave_lds_sync();
            if (lane < 24) L.u.p.hbuf[lane] = (uint8_t)pf_byte;
            wave_lds_sync();
        } else {
            stage_bytes(L.u.p.hbuf, src + pos, (n - pos) < 24 ? (n - pos) : 24, lane);
        }
        // the 24 bytes in registers (one LDS round trip); hwin(o) = the eight bytes from offset o <= 16
        const uint64_t H0 = *reinterpret_cast<const uint64_t*>(L.u.p.hbuf), H1 = *reinterpret_cast<const uint64_t*>(L.u.p.hbuf + 8),
                       H2 = *reinterpret_cast<const uint64_t*>(L.u.p.hbuf + 16);
        auto hwin = [&](uint32_t o) -> uint64_t {
            const uint64_t a = o < 8 ? H0 : H1, c = o < 8 ? H1 : H2;
            const uint32_t sh = 8u * (o & 7u);
            return sh ? ((a >> sh) | (c << (64u - sh))) : a;
        };
        const uint32_t bh = (uint32_t)H0 & 0xFFFFFFu;
        pos += 3;
        const uint32_t last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3 || (SPANS && tree_only && btype != 2)) FAIL();
        pf_ok = false;
        if (!last) {
            const uint64_t nextpos = (uint64_t)pos + (btype == 1 ? 1u : bsize);
            if (nextpos + 3 <= n) {
                pf_byte = (lane < 24 && nextpos + (uint32_t)lane < n) ? src[nextpos + (uint32_t)lane] : 0u;
                pf_ok = true;
            }
        }
        // ---- fast path: a compressed block of four treeless Huffman streams and nothing else (what the device encoder
        // writes for the data bytes: 14 or 15 of a read's 16 or 17 blocks).  Same checks, same tasks as the general code
        // below, without its detours; the verdict on the block's last byte -- it must say "no sequences" -- is collected
        // later.  Anything else about the block sends it down the general path.
        bool quick = false;
        if (btype == 2 && huf_valid && ntask + 4 <= (uint32_t)WAVE && bsize >= 5 && bsize < BLOCK_MAX && (uint64_t)pos + bsize <= n) {
            const uint64_t v = hwin(3);
            const uint32_t h0 = (uint32_t)v & 0xFF, fmt = (h0 >> 2) & 3;
            if ((h0 & 3) == 3 && fmt != 0) {
                uint32_t lh, regen, csize;
                if (fmt == 1) { lh = 3; regen = (uint32_t)(v >> 4) & 0x3FF; csize = (uint32_t)(v >> 14) & 0x3FF; }
                else if (fmt == 2) { lh = 4; regen = (uint32_t)(v >> 4) & 0x3FFF; csize = (uint32_t)(v >> 18) & 0x3FFF; }
                else { lh = 5; regen = (uint32_t)(v >> 4) & 0x3FFFF; csize = (uint32_t)(v >> 22) & 0x3FFFF; }
                const uint64_t j = hwin(3 + lh);
                const uint32_t s1 = (uint32_t)j & 0xFFFFu, s2 = (uint32_t)(j >> 16) & 0xFFFFu, s3 = (uint32_t)(j >> 32) & 0xFFFFu;
                const uint32_t seg = (regen + 3) >> 2;
                if (lh + csize + 1 == bsize && regen != 0 && csize >= 10 && regen <= BLOCK_MAX && regen <= block_max &&
                    s1 + s2 + s3 <= csize - 6 && seg * 3 <= regen && (uint64_t)opos + regen <= fcs) {
                    if (lazy_seq) FAIL();                   // the block before this one did have sequences after all
                    lazy_seq = src[pos + bsize - 1];        // (every lane the same byte)
                    const uint32_t q0 = pos + lh + 6, qn = csize - 6;
                    // (a walked frame's block whose literals -- its content -- stand in the output already: ref_pieces_kernel, RefLits.tail)
                    bool standing = false;
                    if (mypre != nullptr && attempt == 0 && chains.lits != nullptr) {
                        for (uint32_t k = 0; k < chains.lits_units && !standing; ++k) {
                            const RefLits c = chains.lits[(size_t)k * b.n_reads + r];
                            standing = uni((c.blk != 0 && c.blk == pos - 3 && c.regen == regen && c.csize == csize && c.tail != 0u &&
                                            (uint64_t)opos + regen == c.tail) ? 1u : 0u) != 0;
                        }
                    }
                    if (standing) {
                    } else if (lane < 4) {
                        const uint32_t so = lane == 0 ? 0 : (lane == 1 ? s1 : (lane == 2 ? s1 + s2 : s1 + s2 + s3));
                        const uint32_t sz = lane == 0 ? s1 : (lane == 1 ? s2 : (lane == 2 ? s3 : qn - s1 - s2 - s3));
                        L.t_src[ntask + lane] = q0 + so;
                        L.t_size[ntask + lane] = sz;
                        L.t_out[ntask + lane] = opos + (uint32_t)lane * seg;
                        L.t_cnt[ntask + lane] = lane < 3 ? seg : regen - 3 * seg;
                        L.t_tab[ntask + lane] = (uint32_t)cur_slot * (uint32_t)HUF_SLOT | ((uint32_t)cur_log << 16);
                    }
                    if (!standing) ntask += 4;
                    wave_lds_sync();
                    opos += regen;
                    pos += bsize;
                    quick = true;
                }
            }
        }
        if (quick) {
        } else if (btype == 0) {  // Raw_Block
            if (bsize > block_max || pos + bsize > n || (uint64_t)opos + bsize > fcs) FAIL();
            for (uint32_t i = lane; i < bsize; i += WAVE) dst[opos + i] = src[pos + i];
            opos += bsize;
            pos += bsize;
        } else if (btype == 1) {  // RLE_Block
            if (bsize > block_max || pos + 1 > n || (uint64_t)opos + bsize > fcs) FAIL();
            const uint8_t v = src[pos];
            for (uint32_t i = lane; i < bsize; i += WAVE) dst[opos + i] = v;
            opos += bsize;
            pos += 1;
        } else {  // Compressed_Block
            if (bsize >= BLOCK_MAX || pos + bsize > n || bsize < 2) FAIL();
            const uint8_t* blk = src + pos;
            // ---- literals section header (3.1.1.3.1.1), parsed by every lane from the staged bytes
            uint32_t ltype, lh = 0, regen = 0, csize = 0, streams = 1, tree_used = 0, nw = 0, tlog = 0;
            uint32_t jt[3] = { 0, 0, 0 };  // jump table when it sits right behind the header (treeless blocks)
            {
                const uint64_t v = hwin(3);  // h[0..7] of the literals section
                const uint32_t h0 = (uint32_t)v & 0xFF;
                const uint32_t fmt = (h0 >> 2) & 3;
                ltype = h0 & 3;
                if (ltype < 2) {
                    if (fmt == 0 || fmt == 2) { lh = 1; regen = h0 >> 3; }
                    else if (fmt == 1) { lh = 2; regen = ((uint32_t)v & 0xFFFFu) >> 4; }
                    else { lh = 3; regen = ((uint32_t)v & 0xFFFFFFu) >> 4; }
                    if (lh > bsize) FAIL();
                    csize = ltype == 0 ? regen : 1;
                } else {
                    if (bsize < 5) FAIL();  // libzstd: srcSize >= 5 for compressed literals
                    if (fmt < 2) { lh = 3; regen = (uint32_t)(v >> 4) & 0x3FF; csize = (uint32_t)(v >> 14) & 0x3FF; streams = fmt == 0 ? 1 : 4; }
                    else if (fmt == 2) { lh = 4; regen = (uint32_t)(v >> 4) & 0x3FFF; csize = (uint32_t)(v >> 18) & 0x3FFF; streams = 4; }
                    else { lh = 5; regen = (uint32_t)(v >> 4) & 0x3FFFF; csize = (uint32_t)(v >> 22) & 0x3FFFF; streams = 4; }
                    if (regen == 0 || csize == 0) FAIL();
                }
                if (regen > BLOCK_MAX) FAIL();
                if ((uint64_t)lh + csize > bsize) FAIL();
                if (ltype == 3 && !huf_valid) FAIL();
                if (ltype == 3) {
                    const uint64_t j = hwin(3 + lh);
                    jt[0] = (uint32_t)j & 0xFFFFu;
                    jt[1] = (uint32_t)(j >> 16) & 0xFFFFu;
                    jt[2] = (uint32_t)(j >> 32) & 0xFFFFu;
                }
            }
            // ---- sequences section header (3.1.1.3.2.1): its first bytes are requested now, used further down
            const uint32_t lit_end = lh + csize;  // offset of the sequences section in the block
            if (lit_end >= bsize) FAIL();
            const uint8_t* sq = blk + lit_end;
            const uint32_t sqn = bsize - lit_end;
            uint64_t sq8;  // sq[0..7]; bytes past the block are never looked at (the arena is readable 16 bytes past its end)
            __builtin_memcpy(&sq8, sq, 8);
#define SQB(i) ((uint32_t)(sq8 >> (8 * (i))) & 0xFFu)
            wave_lds_sync();
            PHASE(6);
            // a walked frame's literals may stand already (ref_pieces_kernel, beside the walk): the same block and sizes here, the same
            // place further down.  Then nobody needs this block's table unless a later block is coded under it: the last block's is not built.
            bool lits_cand = false;
            RefLits rl = {};
            uint32_t lits_unit = 0;
            if (use_pre && chains.lits != nullptr && ltype >= 2 && streams == 4) {
                for (uint32_t k = 0; k < chains.lits_units && !lits_cand; ++k) {   // (a record per block that was taken, in block order)
                    const RefLits c = chains.lits[(size_t)k * b.n_reads + r];
                    if (uni((c.blk != 0 && c.blk == pos - 3 && c.regen == regen && c.csize == csize) ? 1u : 0u) != 0) {
                        rl = c;
                        lits_unit = k;
                        lits_cand = true;
                    }
                }
            }
            if (ltype == 2 && !(lits_cand && last)) {
                PHASE(0);
                // tree description: at most 129 bytes; the weights are decoded by wave-uniform code, the wave fills the table
                const uint32_t tn = csize < 160 ? csize : 160;
                int used_tree = huf_read_tree(blk + lh, tn, lane, &nw, &tlog);
                if (used_tree == -2) {  // unusual description: the byte-serial reader
                    stage_bytes(L.u.p.hbuf, blk + lh, tn, lane);
                    if (lane == 0) {
                        int inw = 0, ilog = 0;
                        const int used = huf_read_weights(L.u.p.hbuf, (int)tn, &inw, &ilog);
                        L.ctl[C_F] = (uint32_t)used;
                        L.ctl[C_G] = (uint32_t)inw;
                        L.ctl[C_H] = (uint32_t)ilog;
                    }
                    __syncthreads();
                    used_tree = (int)L.ctl[C_F];
                    nw = L.ctl[C_G];
                    tlog = L.ctl[C_H];
                }
                if (used_tree < 0) FAIL();
                if (SPANS && !tree_only) tree_info |= (pos - 3u == first_block) ? DSPAN_ST_TREE_FIRST : DSPAN_ST_TREE_LATER;
                tree_used = (uint32_t)used_tree;
                // new Huffman table: pick the slot not used by the current table; pending tasks that
                // still reference the slot we are about to overwrite must run first
                int slot = huf_valid ? 1 - cur_slot : 0;
                bool clash = (tlog == 12) || (huf_valid && cur_log == 12);
                for (uint32_t t = 0; t < ntask && !clash; ++t)
                    clash = ((L.t_tab[t] & 0xFFFF) == (uint32_t)slot * (uint32_t)HUF_SLOT);
                if (clash && ntask) {
                    FLUSH();
                }
                if (tlog == 12) slot = 0;
                huf_fill_table(&L.huf[0][0] + slot * HUF_SLOT, nw, tlog, lane);
                huf_valid = true;
                cur_slot = slot;
                cur_log = (int)tlog;
                PHASE(5);
            }
            if (SPANS && tree_only) {   // the table stands (or the block has no tree: not what the index promised)
                if (ltype != 2) FAIL();
                tree_only = false;
                pos = first_block;
                pf_ok = false;
                continue;
            }
            uint32_t nseq = SQB(0), sq_used = 1;
            const bool has_seq = nseq != 0;
            if (!has_seq && sqn != 1) FAIL();
            if ((uint64_t)opos + regen > fcs) FAIL();
            // ---- literals: where do they go?
            //   no sequences : straight to the output (Huffman streams become pending tasks)
            //   sequences    : Huffman literals are staged right-aligned at the end of the frame's
            //                  output, where the growing output can never overtake the unread part
            // A block whose sequences are all "offset 1" runs (OF table = RLE of code 0 while repeat offset 1 is 1:
            // how zstd_encode.hip codes zero runs) is not executed in order: lane 0 only walks its FSE states now,
            // its literal streams join the task queue (staged behind the frame in the slack of the destination
            // slot), and the runs are placed in parallel after the next flush.
            bool defer = false, fast_tabs = false;
            uint32_t ws_lit = 0, ws_pairs = 0, ns_fast = 0, used_fast = 0;
            if (has_seq && attempt == 0 && !use_pre) {
                if (d_active) FLUSH();
                uint32_t go = 0;
                if (lane == 0 && sqn >= 4) {
                    const uint32_t used0 = nseq < 128 ? 1u : (nseq < 255 ? 2u : 3u);
                    const uint32_t ns0 = nseq < 128 ? nseq : (nseq < 255 ? ((nseq - 128) << 8) + SQB(1) : SQB(1) + (SQB(2) << 8) + 0x7F00);
                    if (used0 + 2 < sqn) {
                        const uint32_t modes = SQB(used0);
                        const uint32_t llm = modes >> 6, ofm = (modes >> 4) & 3;
                        const uint32_t ofsym_at = used0 + 1 + (llm == 1 ? 1u : 0u);
                        // (64-bit: a crafted index on a frame of half a gigabyte could make the stripe offset wrap)
                        const uint64_t ws_stripe = (uint64_t)DSPAN_WS_FACTOR * sp.dst_len + 32u;
                        const uint64_t ws_lit64 = (((uint64_t)fcs + 15u) & ~15ull) +
                                                  (partial ? (((uint64_t)DSPAN_WS_FACTOR * sp.dst_pos + 32ull * sp.ord + 15u) & ~15ull) : 0ull);
                        const bool ws_ok = ws_lit64 + BLOCK_MAX + 16 < 0xFFFFFFF0ull;
                        ws_lit = ws_ok ? (uint32_t)ws_lit64 : 0u;
                        ws_pairs = ws_lit + (ltype >= 2 ? ((regen + 7u) & ~7u) : 0u);
                        go = (ws_ok && ofm == 1 && llm != 2 && ofsym_at < sqn && SQB(ofsym_at) == 0 && rep0 == 1 &&
                              (uint64_t)ws_pairs + 8ull * ns0 + 8 <= cap &&
                              (!partial || (uint64_t)ws_pairs + 8ull * ns0 + 8 + 16 <= (uint64_t)ws_lit + ws_stripe)) ? 1u : 0u;
                        // predefined LL and ML tables (what zstd_encode.hip writes): nothing to build
                        if (go && modes == 0x10u && dtabs != nullptr) go = 2u | (used0 << 2) | (ns0 << 4);
                    }
                }
                go = (uint32_t)__builtin_amdgcn_readlane((int)go, 0);
                defer = go != 0;
                fast_tabs = (go & 2u) != 0;
                ns_fast = go >> 4;
                used_fast = (go >> 2) & 3u;
                ws_lit = (uint32_t)__builtin_amdgcn_readlane((int)ws_lit, 0);
                ws_pairs = (uint32_t)__builtin_amdgcn_readlane((int)ws_pairs, 0);
            }
            // Other blocks with sequences (libzstd's frames): if the slot has room behind the frame for the literals and
            // one 16-byte record per sequence, the sequences are decoded first and executed in parallel (see below);
            // otherwise they run one by one with the literals staged right-aligned at the end of the output.
            // general sequences reach across spans (matches, repeat offsets): only the first span may have them
            if (has_seq && !defer && partial && !(sp.flags & DSPAN_FIRST)) FAIL();
            bool par = false;
            uint32_t ws_plit = 0, ws_seq = 0;
            if (has_seq && !defer) {
                const uint32_t ns_hdr = nseq < 128 ? nseq : (nseq < 255 ? ((nseq - 128) << 8) + SQB(1) : SQB(1) + (SQB(2) << 8) + 0x7F00);
                ws_plit = (fcs + 15u) & ~15u;
                ws_seq = ws_plit + (ltype >= 2 ? ((regen + 15u) & ~15u) : 0u);
                par = (uint64_t)ws_seq + (use_pre ? 0ull : 16ull * ns_hdr) + 16 <= cap;  // (walked chains: the records are elsewhere)
            }
            const uint32_t lit_dst = !has_seq ? opos : (defer ? ws_lit : (par ? ws_plit : fcs - regen));
            const uint8_t* lit_src = blk + lh;  // raw literals are read in place
            // (with sequences: the stripes are read where the literals would be staged; without: the literals are the block's content and
            // stand in the output if the block begins where the pieces' kernel took it to begin)
            const bool lits_ahead = lits_cand && uni((has_seq ? rl.at == lit_dst : (rl.tail != 0u && (uint64_t)opos + regen == rl.tail)) ? 1u : 0u) != 0;
            if (lits_cand && !lits_ahead) {   // (not where this decoder wants them: the frame again, without hand-overs)
                restart = true;
                break;
            }
            if (ltype >= 2 && !lits_ahead) {
                const uint8_t* q = blk + lh + tree_used;
                uint32_t qn = csize - tree_used;
                if (ntask + streams > WAVE) {
                    FLUSH();
                }
                const uint32_t tabref = (uint32_t)cur_slot * (uint32_t)HUF_SLOT | ((uint32_t)cur_log << 16);
                if (streams == 1) {
                    if (lane == 0) {
                        L.t_src[ntask] = (uint32_t)(q - src);
                        L.t_size[ntask] = qn;
                        L.t_out[ntask] = lit_dst;
                        L.t_cnt[ntask] = regen;
                        L.t_tab[ntask] = tabref;
                    }
                    ntask += 1;
                } else {
                    if (qn < 10) FAIL();
                    uint32_t s1 = jt[0], s2 = jt[1], s3 = jt[2];
                    if (ltype == 2) {  // the jump table follows the tree description
                        s1 = q[0] | ((uint32_t)q[1] << 8);
                        s2 = q[2] | ((uint32_t)q[3] << 8);
                        s3 = q[4] | ((uint32_t)q[5] << 8);
                    }
                    q += 6;
                    qn -= 6;
                    if (s1 + s2 + s3 > qn) FAIL();
                    const uint32_t s4 = qn - s1 - s2 - s3;
                    const uint32_t seg = (regen + 3) >> 2;
                    if (seg * 3 > regen) FAIL();
                    if (lane < 4) {
                        const uint32_t so = lane == 0 ? 0 : (lane == 1 ? s1 : (lane == 2 ? s1 + s2 : s1 + s2 + s3));
                        const uint32_t sz = lane == 0 ? s1 : (lane == 1 ? s2 : (lane == 2 ? s3 : s4));
                        L.t_src[ntask + lane] = (uint32_t)(q - src) + so;
                        L.t_size[ntask + lane] = sz;
                        L.t_out[ntask + lane] = lit_dst + (uint32_t)lane * seg;
                        L.t_cnt[ntask + lane] = lane < 3 ? seg : regen - 3 * seg;
                        L.t_tab[ntask + lane] = tabref;
                    }
                    ntask += 4;
                }
                wave_lds_sync();
            }
            PHASE(7);
            if (!has_seq) {
                if (ltype == 0) {
                    for (uint32_t i = lane; i < regen; i += WAVE) dst[opos + i] = lit_src[i];
                } else if (ltype == 1) {
                    const uint8_t v = lit_src[0];
                    for (uint32_t i = lane; i < regen; i += WAVE) dst[opos + i] = v;
                }
                if (regen > block_max) FAIL();
                opos += regen;
            } else if (defer) {
                PHASE(0);
                uint2 llt = make_uint2(0, 0), mlt = make_uint2(0, 0);
                uint32_t lgl = 6, lgm = 6;
                if (fast_tabs) {
                    nseq = ns_fast;
                    sq_used = used_fast + 2u;  // count, modes byte, RLE symbol of the OF table
                    if (sq_used >= sqn) FAIL();
                    llt = dtabs->ll[lane];
                    mlt = dtabs->ml[lane];
                    tables_lost = true;  // nothing was built in LDS: a later Repeat_Mode block restarts the frame
                } else {
                    // tables of this block (the union LDS is free: queued tasks are only descriptors)
                    stage_bytes(L.u.p.hbuf, sq, sqn < HBUF ? sqn : HBUF, lane);
                    if (lane == 0) {
                        uint32_t err = 0, used = 1, ns = nseq;
                        const bool lost = tables_lost;
                        const uint8_t* h = L.u.p.hbuf;
                        const int hn = (int)(sqn < HBUF ? sqn : HBUF);
                        if (ns >= 128) {
                            if (ns == 255) {
                                if (hn < 3) err = 1; else { ns = h[1] + ((uint32_t)h[2] << 8) + 0x7F00; used = 3; }
                            } else {
                                if (hn < 2) err = 1; else { ns = ((ns - 128) << 8) + h[1]; used = 2; }
                            }
                        }
                        if (!err && (int)used >= hn) err = 1;
                        if (!err) {
                            const uint32_t modes = h[used++];
                            if (modes & 3) err = 1;
                            // a repeated table that a ring flush has overwritten: decode the frame again, carefully
                            if (!err && lost && (((modes >> 6) & 3) == 3 || ((modes >> 4) & 3) == 3 || ((modes >> 2) & 3) == 3)) err = 3;
                            int u;
                            if (!err) {
                                u = seq_table(L.u.p.fse[0], &log_ll, &have_ll, (modes >> 6) & 3, h + used, hn - (int)used, LL_DEFAULT, 36, 6, 35, 9);
                                if (u < 0) err = 1; else used += (uint32_t)u;
                            }
                            if (!err) {
                                u = seq_table(L.u.p.fse[1], &log_of, &have_of, (modes >> 4) & 3, h + used, hn - (int)used, OF_DEFAULT, 29, 5, 31, 8);
                                if (u < 0) err = 1; else used += (uint32_t)u;
                            }
                            if (!err) {
                                u = seq_table(L.u.p.fse[2], &log_ml, &have_ml, (modes >> 2) & 3, h + used, hn - (int)used, ML_DEFAULT, 53, 6, 52, 9);
                                if (u < 0) err = 1; else used += (uint32_t)u;
                            }
                            if (!err && used >= sqn) err = 1;
                        }
                        L.ctl[C_ERR] = err;
                        L.ctl[C_I] = ns;
                        L.ctl[C_J] = used;
                    }
                    __syncthreads();
                    if (L.ctl[C_ERR] == 3) {
                        restart = true;
                        break;
                    }
                    if (L.ctl[C_ERR]) FAIL();
                    nseq = L.ctl[C_I];
                    sq_used = L.ctl[C_J];
                    tables_built = true;
                    // compact per-lane tables; eligibility: accuracy logs <= 6, OF = RLE of code 0, valid codes
                    uint32_t bad = 0;
                    if (lane == 0) bad = (log_ll > 6 || log_ml > 6 || log_of != 0 || (L.u.p.fse[1][0] & 0xFF) != 0) ? 1u : 0u;
                    if (__builtin_amdgcn_readlane((int)bad, 0)) {
                        restart = true;
                        break;
                    }
                    lgl = (uint32_t)__builtin_amdgcn_readlane(log_ll, 0);
                    lgm = (uint32_t)__builtin_amdgcn_readlane(log_ml, 0);
                    uint32_t e1 = 0, e2 = 0;
                    if (lane < (1 << lgl)) {
                        const uint32_t e = L.u.p.fse[0][lane], c = e & 0xFF;
                        if (c > 35) e1 = 1;
                        else llt = make_uint2(LL_BASE[c], (e >> 16) | ((uint32_t)LL_BITS[c] << 16) | (((e >> 8) & 0xFF) << 24));
                    }
                    if (lane < (1 << lgm)) {
                        const uint32_t e = L.u.p.fse[2][lane], c = e & 0xFF;
                        if (c > 52) e2 = 1;
                        else mlt = make_uint2(ML_BASE[c], (e >> 16) | ((uint32_t)ML_BITS[c] << 16) | (((e >> 8) & 0xFF) << 24));
                    }
                    if (__any(e1 | e2)) FAIL();
                }
                {
                    PHASE(2);
                    uint32_t total = 0;
                    uint32_t ok = 3;
                    if (fast_tabs && cp_avail) {
                        cp_avail = false;  // the trailer describes one sequences section
                        ok = zero_run_chain_segments(sq + sq_used, sqn - sq_used, reinterpret_cast<uint2*>(dst + ws_pairs), nseq, llt, mlt,
                                                     regen, cp_tab, cp_count, cp_spacing, lane, &total, &L.u.inbuf[0][0], (uint32_t)sizeof(L.u.inbuf));
                    }
                    if (ok == 3)
                        ok = zero_run_chain(sq + sq_used, sqn - sq_used, reinterpret_cast<uint2*>(dst + ws_pairs), nseq, llt, mlt, lgl, lgm,
                                            regen, lane, &total);
                    PHASE(3);
                    if (ok == 0) FAIL();
                    if (ok == 2) {  // not a pure zero-run block after all: decode the frame again, in order
                        restart = true;
                        break;
                    }
                    if ((uint64_t)opos + total > fcs || total > block_max) FAIL();
                    d_active = true;
                    d_opos = opos;
                    d_regen = regen;
                    d_nseq = nseq;
                    d_ltype = ltype;
                    d_pairs = ws_pairs;
                    d_lit = ltype >= 2 ? dst + ws_lit : lit_src;
                    opos += total;
                    tables_lost = true;  // the next ring flush reuses the LDS of these tables
                }
            } else {
                // everything decoded so far must be in memory before matches can read it
                PHASE(0);
                FLUSH();
                PHASE(8);
                // the block's chain may have been walked ahead of this launch (zstd_decode_ref.hip): then its records, its sequence
                // count and the repeat offsets behind it are taken from there, no tables are built and no bit stream is read here
                const RefBlock* pb = nullptr;
                if (use_pre) {
                    if (!par || pre_k >= mypre->nblk || mypre->blk[pre_k].pos != pos - 3) {
                        restart = true;
                        break;
                    }
                    pb = &mypre->blk[pre_k++];
                }
                // (this block's own Huffman literals were queued above, so they were decoded with the ring too)
                if (!fse_live && tables_built) tables_lost = true;  // the ring shares its LDS with the FSE tables
                if (pb) {
                    nseq = uni(pb->nseq);
                } else {
                stage_bytes(L.u.p.hbuf, sq, sqn < HBUF ? sqn : HBUF, lane);
                if (lane == 0) {
                    uint32_t err = 0, used = 1, ns = nseq;
                    const bool lost = tables_lost;
                    const uint8_t* h = L.u.p.hbuf;
                    const int hn = (int)(sqn < HBUF ? sqn : HBUF);
                    if (ns >= 128) {
                        if (ns == 255) {
                            if (hn < 3) err = 1; else { ns = h[1] + ((uint32_t)h[2] << 8) + 0x7F00; used = 3; }
                        } else {
                            if (hn < 2) err = 1; else { ns = ((ns - 128) << 8) + h[1]; used = 2; }
                        }
                    }
                    if (!err && (int)used >= hn) err = 1;
                    if (!err) {
                        const uint32_t modes = h[used++];
                        if (modes & 3) err = 1;
                        // a repeated table that a ring flush has overwritten: decode the frame again, carefully
                        if (!err && lost && (((modes >> 6) & 3) == 3 || ((modes >> 4) & 3) == 3 || ((modes >> 2) & 3) == 3)) err = 3;
                        int u;
                        if (!err) {
                            u = seq_table(L.u.p.fse[0], &log_ll, &have_ll, (modes >> 6) & 3, h + used, hn - (int)used, LL_DEFAULT, 36, 6, 35, 9);
                            if (u < 0) err = 1; else used += (uint32_t)u;
                        }
                        if (!err) {
                            u = seq_table(L.u.p.fse[1], &log_of, &have_of, (modes >> 4) & 3, h + used, hn - (int)used, OF_DEFAULT, 29, 5, 31, 8);
                            if (u < 0) err = 1; else used += (uint32_t)u;
                        }
                        if (!err) {
                            u = seq_table(L.u.p.fse[2], &log_ml, &have_ml, (modes >> 2) & 3, h + used, hn - (int)used, ML_DEFAULT, 53, 6, 52, 9);
                            if (u < 0) err = 1; else used += (uint32_t)u;
                        }
                        if (!err && used >= sqn) err = 1;
                    }
                    L.ctl[C_ERR] = err;
                    L.ctl[C_I] = ns;
                    L.ctl[C_J] = used;
                }
                __syncthreads();
                if (L.ctl[C_ERR] == 3) {
                    restart = true;
                    break;
                }
                if (L.ctl[C_ERR]) FAIL();
                nseq = L.ctl[C_I];
                sq_used = L.ctl[C_J];
                tables_built = true;
                }
                PHASE(9);
                const uint8_t* litp = ltype == 0 ? lit_src : dst + lit_dst;
                // Literals that were decoded ahead stand in 64 stripes (ref_pieces_kernel; RefLits): stripe q holds the literals
                // [P[q], P[q + 1]) at sbase + q * pcap and, behind them, the next stripe's first 256 (= LANE_COPY_MAX): whatever a LANE
                // reads -- a run of up to 256 literals, the source of a match inside it -- it reads in one piece from where the run
                // begins (`lf` in the trips below); only what the whole wavefront moves (longer runs) goes stripe by stripe (lits_wave).
                const bool striped = lits_ahead;
                const uint8_t* const sbase = dst + rl.tb;
                const uint32_t spcap = rl.pcap;
                // P: lane q holds P[q].  Where a literal stands is a ballot (how many stripes begin at or before it), no table in LDS:
                const uint32_t Pv = striped ? chains.lits_pos[((size_t)lits_unit * b.n_reads + r) * WAVE + lane] : 0u;
                auto lit_seg = [&](uint32_t idx, uint32_t& room) -> const uint8_t* {   // idx UNIFORM; room: literals of its stripe from it on
                    if (!striped) {
                        room = 0xFFFFFFFFu;
                        return litp + idx;
                    }
                    const uint32_t q = (uint32_t)__popcll(__ballot(Pv <= idx)) - 1u;   // (P[0] = 0)
                    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)Pv, (int)q);
                    const uint32_t p1n = q < 63u ? (uint32_t)__builtin_amdgcn_readlane((int)Pv, (int)(q + 1u)) : regen;
                    const uint32_t p1 = p1n < regen ? p1n : regen;   // (lanes without a stripe hold 0xFFFFFFFF)
                    room = p1 - idx;
                    return sbase + q * spcap + (idx - p0);
                };
                // all lanes (uniform arguments): n literals from idx on to o.  Out of stripes: four stripes' loads are in flight before their
                // stores -- the literals behind a block's last sequence are most of what a read's data bytes are, tens of kilobytes that
                // cross dozens of stripes; a stripe at a time (load, wait, store) this copy alone took longer than all the sequences.
                auto lits_wave = [&](gu8* o, uint32_t idx, uint32_t n) {
                    if (!striped) {
                        wave_copy(o, (gcu8*)(litp + idx), n, lane);
                        return;
                    }
                    constexpr int GW = VBZ_DEC_LITS_GW;
                    const uint32_t k0 = 16u * (uint32_t)lane;
                    while (n) {
                        gcu8* f[GW];
                        gu8* d[GW];
                        uint32_t c[GW];
#pragma unroll
                        for (int u = 0; u < GW; ++u) {
                            uint32_t room = 0;
                            f[u] = (gcu8*)lit_seg(idx < regen ? idx : 0u, room);
                            c[u] = n < room ? n : room;
                            d[u] = o;
                            o += c[u];
                            idx += c[u];
                            n -= c[u];
                        }
                        u32x4 v[GW][2];
                        uint32_t tb_[GW];
#pragma unroll
                        for (int u = 0; u < GW; ++u) {
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch)
                                if (k0 + 1024u * ch + 16u <= c[u]) v[u][ch] = *(gld16*)(f[u] + k0 + 1024u * ch);
                            const uint32_t rb = c[u] & 15u;
                            tb_[u] = (uint32_t)lane < rb ? f[u][c[u] - rb + (uint32_t)lane] : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < GW; ++u) {
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch)
                                if (k0 + 1024u * ch + 16u <= c[u]) *(gst16*)(d[u] + k0 + 1024u * ch) = v[u][ch];
                            const uint32_t rb = c[u] & 15u;
                            if ((uint32_t)lane < rb) d[u][c[u] - rb + (uint32_t)lane] = (uint8_t)tb_[u];
                            for (uint32_t k = k0 + 2048u; k + 16u <= c[u]; k += 1024u) *(gst16*)(d[u] + k) = *(gld16*)(f[u] + k);   // (a stripe of more than 2 KB)
                        }
                    }
                };
                const uint8_t rle_byte = ltype == 1 ? lit_src[0] : 0;
                const uint8_t* bs = sq + sq_used;
                const uint32_t bsn = sqn - sq_used;
                if (!pb) fse_live = true;  // libzstd-style frame: keep the FSE tables across the flushes of later blocks
                BitReader br;
                uint32_t sl = 0, so = 0, sm = 0;
                uint32_t err = 0;
                if (lane == 0 && !pb) {
                    if (!br.init(bs, bsn)) err = 1;
                    if (!err) {
                        sl = br.read(log_ll);
                        so = br.read(log_of);
                        sm = br.read(log_ml);
                        if (br.over) err = 1;
                    }
                }
                if (__builtin_amdgcn_readlane((int)err, 0)) FAIL();
                uint32_t lpos = 0;
                const uint32_t block_start = opos;
                if (par) {
                    // ---- (A) the three state machines are walked once, the sequences only recorded, fully checked
                    const uint4* seqbuf = pb ? reinterpret_cast<const uint4*>(chains.recs) + (((uint64_t)uni(pb->rec_hi) << 32) | uni(pb->rec_lo))
                                             : reinterpret_cast<const uint4*>(dst + ws_seq);
                    if (pb) {
                        rep0 = uni(pb->rep[0]);
                        rep1 = uni(pb->rep[1]);
                        rep2 = uni(pb->rep[2]);
                    } else {
                        uint4* seqbuf = reinterpret_cast<uint4*>(dst + ws_seq);
                        uint32_t reps[3] = { rep0, rep1, rep2 };
#ifdef VBZ_SEQ_CHAIN_SCALAR
                        const bool good = general_sequence_records(bs, bsn, seqbuf, nseq, (uint32_t)__builtin_amdgcn_readlane(log_ll, 0), (uint32_t)__builtin_amdgcn_readlane(log_of, 0),
                                                                   (uint32_t)__builtin_amdgcn_readlane(log_ml, 0), regen, opos, fcs, reps, lane);
#else
                        const bool good = general_sequence_records_lanes(bs, bsn, seqbuf, nseq, (uint32_t)__builtin_amdgcn_readlane(log_ll, 0), (uint32_t)__builtin_amdgcn_readlane(log_of, 0),
                                                                         (uint32_t)__builtin_amdgcn_readlane(log_ml, 0), regen, opos, fcs, reps, lane);
#endif
                        if (!good) FAIL();
                        rep0 = reps[0];
                        rep1 = reps[1];
                        rep2 = reps[2];
                    }
                    __syncthreads();  // the records (and everything decoded so far) are in memory
                    PHASE(10);
                    // ---- (B) all literals at once: positions from prefix sums, 64 sequences per trip
                    {
                        // With them go the matches that repeat the sequence's OWN literals (offset <= literal length) without
                        // overlapping themselves, or as a pattern of 1, 2, 4, 8 or 16 bytes -- the zero runs of the control bytes
                        // are all of that kind: their source is read from the literal buffer, so they depend on nothing that is
                        // written here and need no round of (C) (SEQ_OWN: the same predicate takes them out of (C)'s lists).
#define SEQ_OWN(sv) (ltype != 1 && (sv).y != 0 && (sv).z <= (sv).x && ((sv).z >= (sv).y || (sv).z == 1 || (sv).z == 2 || (sv).z == 4 || (sv).z == 8 || (sv).z == 16))
                        uint32_t lposw = 0, oposw = opos;
                        uint4 svn = (uint32_t)lane < nseq ? seqbuf[lane] : make_uint4(0u, 0u, 0u, 0u);
                        for (uint32_t base = 0; base < nseq; base += WAVE) {
                            const uint32_t i = base + (uint32_t)lane;
                            const uint4 sv = svn;
                            svn = i + WAVE < nseq ? seqbuf[i + WAVE] : make_uint4(0u, 0u, 0u, 0u);   // (the next trip's, under this trip's copies)
                            const uint32_t ll = sv.x, tot = sv.x + sv.y;
                            const uint32_t il = wave_incl_scan_u32(ll), it = wave_incl_scan_u32(tot);
                            const uint32_t lp = lposw + il - ll;
                            uint8_t* o = dst + oposw + it - tot;
                            const bool big = ll > LANE_COPY_MAX;  // long runs are moved by the whole wave
                            const bool own = i < nseq && SEQ_OWN(sv);
                            const bool ownpat = own && sv.z < sv.y;
                            const bool ownbig = own && sv.y > LANE_COPY_MAX;
                            u32x4 pv = {};
                            // (where this lane's run stands; the source of an own match of a run of up to 256 literals lies inside the run)
                            // where this lane's run stands: the trip's runs are consecutive literals, so they begin in the stripe of the
                            // trip's first literal or in one of the next few -- one pass over those stripes' first indices
                            const uint8_t* lf = litp + lp;
                            if (striped) {
                                const uint32_t lpend = lposw + (uint32_t)__builtin_amdgcn_readlane((int)il, 63);
                                uint32_t q0 = (uint32_t)__popcll(__ballot(Pv <= lposw)) - 1u;
                                uint32_t q = q0, p0 = (uint32_t)__builtin_amdgcn_readlane((int)Pv, (int)q0);
                                for (uint32_t bq = q0 + 1u; bq < (uint32_t)WAVE; ++bq) {
                                    const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)Pv, (int)bq);
                                    if (pb >= lpend) break;
                                    const bool in = lp >= pb;
                                    q = in ? bq : q;
                                    p0 = in ? pb : p0;
                                }
                                lf = sbase + q * spcap + (lp - p0);
                                // (a long run is moved by the wavefront, stripe by stripe; the source of its own match is looked up by itself)
                                for (uint64_t bo = __ballot(big && own); bo; bo &= bo - 1) {
                                    const int bl = __ffsll((long long)bo) - 1;
                                    uint32_t room;
                                    const uint8_t* f = lit_seg((uint32_t)__builtin_amdgcn_readlane((int)(lp + ll - sv.z), bl), room);
                                    if (lane == bl) lf = f - (ll - sv.z);
                                }
                            }
                            if (ownpat) pv = period_pattern((gcu8*)(lf + ll - sv.z), sv.z);
                            if (ltype == 1) {
                                for (uint32_t k = 0; k < ll; ++k) o[k] = rle_byte;
                            } else if (!big) {
                                lane_copy((gu8*)o, (gcu8*)lf, ll);
                            }
                            if (own && !ownbig) {
                                if (ownpat) lane_fill((gu8*)o + ll, pv, sv.y);
                                else lane_copy((gu8*)o + ll, (gcu8*)(lf + ll - sv.z), sv.y);
                            }
                            for (uint64_t bigs = ltype == 1 ? 0ull : __ballot(big); bigs; bigs &= bigs - 1) {
                                const int bl = __ffsll((long long)bigs) - 1;
                                const uint32_t bo = (uint32_t)__builtin_amdgcn_readlane((int)(oposw + it - tot), bl);
                                const uint32_t bf = (uint32_t)__builtin_amdgcn_readlane((int)lp, bl);
                                lits_wave((gu8*)dst + bo, bf, (uint32_t)__builtin_amdgcn_readlane((int)ll, bl));
                            }
                            for (uint64_t bigs = __ballot(ownbig); bigs; bigs &= bigs - 1) {
                                const int bl = __ffsll((long long)bigs) - 1;
                                const uint32_t bo = (uint32_t)__builtin_amdgcn_readlane((int)(oposw + it - sv.y), bl), bn = (uint32_t)__builtin_amdgcn_readlane((int)sv.y, bl);
                                if (__builtin_amdgcn_readlane((int)ownpat, bl)) {
                                    u32x4 bv;
                                    bv.x = (uint32_t)__builtin_amdgcn_readlane((int)pv.x, bl);
                                    bv.y = (uint32_t)__builtin_amdgcn_readlane((int)pv.y, bl);
                                    bv.z = (uint32_t)__builtin_amdgcn_readlane((int)pv.z, bl);
                                    bv.w = (uint32_t)__builtin_amdgcn_readlane((int)pv.w, bl);
                                    wave_fill((gu8*)dst + bo, bv, bn, lane);
                                } else {
                                    lits_wave((gu8*)dst + bo, (uint32_t)__builtin_amdgcn_readlane((int)(lp + ll - sv.z), bl), bn);
                                }
                            }
                            lposw += (uint32_t)__builtin_amdgcn_readlane((int)il, 63);
                            oposw += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
                        }
                        PHASE(2);   // (the trips; the literals behind the last sequence are slot 11)
                        // literals behind the last sequence
                        const uint32_t rest = regen - lposw;
                        if ((uint64_t)oposw + rest > fcs) FAIL();
                        if (ltype == 1) {
                            for (uint32_t k = lane; k < rest; k += WAVE) dst[oposw + k] = rle_byte;
                        } else if (!(striped && rl.tail != 0u && oposw + rest == rl.tail)) {   // (else they stand there already: RefLits.tail)
                            lits_wave((gu8*)dst + oposw, lposw, rest);
                        }
                        lpos = regen;
                        __syncthreads();
                        PHASE(11);
                        // ---- (C) matches, 64 sequences per trip, in rounds: a match may be copied once everything
                        // below the first not yet copied match of the trip is final and its source ends below that
                        // point; the first one is always ready (a lane copies byte by byte, so it may overlap itself)
                        uint32_t ow = opos;
                        uint4 svm = (uint32_t)lane < nseq ? seqbuf[lane] : make_uint4(0u, 0u, 0u, 0u);
                        for (uint32_t base = 0; base < nseq; base += WAVE) {
                            const uint32_t i = base + (uint32_t)lane;
                            const uint4 sv = svm;
                            svm = i + WAVE < nseq ? seqbuf[i + WAVE] : make_uint4(0u, 0u, 0u, 0u);
                            const uint32_t tot = sv.x + sv.y;
                            const uint32_t it = wave_incl_scan_u32(tot);
                            const uint32_t mdst = ow + it - sv.y;  // where this lane's match goes
                            const uint32_t msrc = mdst - sv.z;
                            uint64_t todo = __ballot(i < nseq && sv.y != 0 && !SEQ_OWN(sv));
                            while (todo) {
                                const int f = __ffsll((long long)todo) - 1;
                                const uint32_t frontier = (uint32_t)__builtin_amdgcn_readlane((int)mdst, f);
                                const bool mine = ((todo >> lane) & 1ull) && (lane == f || msrc + sv.y <= frontier || sv.z <= sv.x);  // ... or its source lies in its own literals
                                // a match longer than its offset repeats the bytes in front of it: offsets 1, 2, 4, 8, 16
                                // (runs of a byte, of a key pattern) become pattern stores, other short offsets go byte by byte
                                const bool apart = sv.z >= sv.y;
                                const bool pat = !apart && (sv.z == 1 || sv.z == 2 || sv.z == 4 || sv.z == 8 || sv.z == 16);
                                const bool rep = !apart && !pat && sv.z >= 16;   // a long match that overlaps itself at some other distance
                                const bool big = mine && sv.y > LANE_COPY_MAX && (apart || pat || rep);
                                u32x4 pv = {};
                                if (mine && pat) pv = period_pattern((gcu8*)dst + msrc, sv.z);
                                if (mine && !big) {
                                    if (apart || sv.z >= 80) {
                                        lane_copy((gu8*)dst + mdst, (gcu8*)dst + msrc, sv.y);
                                    } else if (pat) {
                                        lane_fill((gu8*)dst + mdst, pv, sv.y);
                                    } else {
                                        for (uint32_t k = 0; k < sv.y; ++k) dst[mdst + k] = dst[msrc + (k % sv.z)];
                                    }
                                }
                                for (uint64_t bigs = __ballot(big); bigs; bigs &= bigs - 1) {
                                    const int bl = __ffsll((long long)bigs) - 1;
                                    const uint32_t bo = (uint32_t)__builtin_amdgcn_readlane((int)mdst, bl), bn = (uint32_t)__builtin_amdgcn_readlane((int)sv.y, bl);
                                    if (__builtin_amdgcn_readlane((int)pat, bl)) {
                                        u32x4 bv;
                                        bv.x = (uint32_t)__builtin_amdgcn_readlane((int)pv.x, bl);
                                        bv.y = (uint32_t)__builtin_amdgcn_readlane((int)pv.y, bl);
                                        bv.z = (uint32_t)__builtin_amdgcn_readlane((int)pv.z, bl);
                                        bv.w = (uint32_t)__builtin_amdgcn_readlane((int)pv.w, bl);
                                        wave_fill((gu8*)dst + bo, bv, bn, lane);
                                    } else if (__builtin_amdgcn_readlane((int)rep, bl)) {
                                        wave_copy_repeat((gu8*)dst + bo, (uint32_t)__builtin_amdgcn_readlane((int)sv.z, bl), bn, lane);
                                    } else {
                                        wave_copy((gu8*)dst + bo, (gcu8*)dst + (uint32_t)__builtin_amdgcn_readlane((int)msrc, bl), bn, lane);
                                    }
                                }
                                todo &= ~__ballot(mine);
                                __syncthreads();  // these bytes are sources of later matches
                            }
                            ow += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
                        }
                        opos = ow + rest;
#undef SEQ_OWN
                    }
                    if (opos - block_start > BLOCK_MAX || opos - block_start > block_max) FAIL();
                    __syncthreads();
                    PHASE(4);
                } else {
                uint32_t safe = opos;  // output below this position is known to be in memory
                for (uint32_t i = 0; i < nseq; ++i) {
                    uint32_t llen = 0, mlen = 0, offset = 0;
                    if (lane == 0) {
                        const uint32_t el = L.u.p.fse[0][sl], eo = L.u.p.fse[1][so], em = L.u.p.fse[2][sm];
                        const uint32_t lc = el & 0xFF, oc = eo & 0xFF, mc = em & 0xFF;
                        if (lc > 35 || mc > 52 || oc > 31) err = 1;
                        if (!err) {
                            const uint32_t ofv = (oc ? (1u << oc) : 1u) + (oc ? br.read((int)oc) : 0u);
                            mlen = ML_BASE[mc] + (ML_BITS[mc] ? br.read(ML_BITS[mc]) : 0u);
                            llen = LL_BASE[lc] + (LL_BITS[lc] ? br.read(LL_BITS[lc]) : 0u);
                            if (ofv > 3) {
                                offset = ofv - 3;
                                rep2 = rep1; rep1 = rep0; rep0 = offset;
                            } else {
                                const uint32_t idx = ofv - 1 + (llen == 0 ? 1u : 0u);
                                if (idx == 0) {
                                    offset = rep0;
                                } else {
                                    offset = idx == 3 ? rep0 - 1 : (idx == 1 ? rep1 : rep2);
                                    if (offset == 0) offset = 1;  // libzstd forces an invalid 0 to 1
                                    if (idx > 1) rep2 = rep1;
                                    rep1 = rep0;
                                    rep0 = offset;
                                }
                            }
                            if (i + 1 < nseq) {
                                sl = (el >> 16) + br.read((int)((el >> 8) & 0xFF));
                                sm = (em >> 16) + br.read((int)((em >> 8) & 0xFF));
                                so = (eo >> 16) + br.read((int)((eo >> 8) & 0xFF));
                            }
                            if (br.over) err = 1;
                        }
                    }
                    llen = (uint32_t)__builtin_amdgcn_readlane((int)llen, 0);
                    mlen = (uint32_t)__builtin_amdgcn_readlane((int)mlen, 0);
                    offset = (uint32_t)__builtin_amdgcn_readlane((int)offset, 0);
                    if (__builtin_amdgcn_readlane((int)err, 0)) FAIL();
                    if (lpos + llen > regen) FAIL();
                    if ((uint64_t)opos + llen + mlen > fcs) FAIL();
                    if (ltype == 1) {
                        for (uint32_t k = lane; k < llen; k += WAVE) dst[opos + k] = rle_byte;
                    } else {
                        for (uint32_t k = lane; k < llen; k += WAVE) dst[opos + k] = litp[lpos + k];
                    }
                    opos += llen;
                    lpos += llen;
                    if (offset > opos) FAIL();
                    const uint32_t from = opos - offset;
                    const uint32_t span = offset < mlen ? offset : mlen;
                    if (from + span > safe) {
                        __syncthreads();  // drain this wave's stores before reading them back
                        safe = opos;
                    }
                    if (offset >= mlen) {
                        for (uint32_t k = lane; k < mlen; k += WAVE) dst[opos + k] = dst[from + k];
                    } else {
                        for (uint32_t k = lane; k < mlen; k += WAVE) dst[opos + k] = dst[from + (k % offset)];
                    }
                    opos += mlen;
                    if (opos - block_start > BLOCK_MAX) FAIL();
                }
                if (lane == 0 && !br.finished()) err = 1;
                if (__builtin_amdgcn_readlane((int)err, 0)) FAIL();
                // remaining literals
                const uint32_t rest = regen - lpos;
                if ((uint64_t)opos + rest > fcs) FAIL();
                if (ltype == 1) {
                    for (uint32_t k = lane; k < rest; k += WAVE) dst[opos + k] = rle_byte;
                } else {
                    // source and destination may overlap when the staged literals sit at the very end
                    // of the frame: then they are already in place (from == to) or strictly above
                    const uint8_t* from = litp + lpos;
                    uint8_t* to = dst + opos;
                    if (from != to) {
                        for (uint32_t k0 = 0; k0 < rest; k0 += WAVE) {
                            const uint32_t k = k0 + lane;
                            uint8_t v = 0;
                            if (k < rest) v = from[k];
                            __syncthreads();
                            if (k < rest) to[k] = v;
                        }
                    }
                }
                opos += rest;
                if (opos - block_start > BLOCK_MAX || opos - block_start > block_max) FAIL();
                __syncthreads();
                }
            }
            pos += bsize;
        }
        if (last) {
            if (partial && !(sp.flags & DSPAN_LAST)) FAIL();  // the frame ends here: nothing of it may follow in later spans
            break;
        }
        if (partial && !(sp.flags & DSPAN_LAST)) {
            if (pos == sp.src_end) break;
            if (pos > sp.src_end) FAIL();
        }
    }
    if (restart) continue;
    if (lazy_seq) FAIL();
    PHASE(0);
    if (ntask || d_active) {
        FLUSH();
    }
    PHASE(1);
    rep0_end = rep0;
    break;
    }
    if (partial && !(sp.flags & DSPAN_LAST)) {
        // the span ended where the next one starts: report where the content stands
        if (lane == 0) {
            dspan_status[4 * blockIdx.x + 1] = pos;
            dspan_status[4 * blockIdx.x + 2] = opos;
            dspan_status[4 * blockIdx.x + 3] = (rep0_end == 1 ? DSPAN_ST_REP1 : 0u) | tree_info;
            dspan_status[4 * blockIdx.x] = 1;
        }
        return;
    }
    if (has_checksum) {
        if (pos + 4 > n) FAIL();
        pos += 4;  // xxh64 of the content: not verified
    }
    // skippable frames behind the frame are legal and ignored (RFC 8878 3.1.2); a second data frame is not what
    // vbz writes and is rejected
    while (n - pos >= 8) {
        uint32_t m[2];
        __builtin_memcpy(m, src + pos, 8);
        if ((m[0] & 0xFFFFFFF0u) != 0x184D2A50u || (uint64_t)pos + 8 + m[1] > n) break;
        pos += 8 + m[1];
    }
    if (pos != n) FAIL();
    if (opos != fcs) FAIL();
    if (partial) {
        if (lane == 0) {
            dspan_status[4 * blockIdx.x + 1] = pos;
            dspan_status[4 * blockIdx.x + 2] = opos;
            dspan_status[4 * blockIdx.x + 3] = (rep0_end == 1 ? DSPAN_ST_REP1 : 0u) | tree_info;
            dspan_status[4 * blockIdx.x] = 1;
        }
        return;
    }
    if (FUSED) {
        static_assert(sizeof(DecLds) >= svbwave::LDS_TOTAL, "the wave's svb decoder works in the frame decoder's LDS");
        __syncthreads();  // every byte of the stream is in memory (raw / RLE blocks store without a drain)
        const uint32_t res = svbwave::svb_decode_wave_i16zz(dst, fcs, fuse.out + fuse.out_off[r], fuse.out_size[r], reinterpret_cast<uint8_t*>(&L), lane);
        if (lane == 0) b.result[r] = res;
    } else if (lane == 0) {
        b.result[r] = fcs;
    }
#ifdef VBZ_SPLIT_DEBUG
    tph[2] = split_dbg[0];
    tph[3] = split_dbg[1];
#endif
    if (TIMED && lane == 0 && !dspans)
        for (int k = 0; k < PHASE_SLOTS; ++k) dbg[(size_t)r * PHASE_SLOTS + k] = tph[k];
#undef SQB
#undef PHASE
#undef FLUSH
#undef FAIL
}

// ---- span mode: plan and finish ----------------------------------------------------------------------------------------
// Frame header and index trailer of every read -> its spans (or one WHOLE span when there is no usable index).
__global__ __launch_bounds__(1024) void zstd_dspan_plan_kernel(ReadBatch b, uint32_t max_spans, DecSpan* spans, uint32_t* dspan_first,
                                                               uint32_t* dspan_count, uint32_t* dspan_status)
{
    __shared__ uint32_t wcnt[16];
    __shared__ uint32_t carry_c, limit_s;
    __shared__ uint32_t q_ns[1024], q_tb[1024], q_ok[1024], q_first[1024], q_hl[1024], q_fcs[1024];
    __shared__ uint32_t q_T[1024];   // 1 + the last span of the read whose first block brings a Huffman tree (0: none)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) {
        carry_c = 0;
        limit_s = max_spans;   // the first span that has no descriptor (a read whose spans do not fit the arrays)
    }
    __syncthreads();
    for (uint32_t base = 0; base < b.n_reads; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t here = (b.n_reads - base) < 1024u ? (b.n_reads - base) : 1024u;
        // (1) one thread per read: frame header (single segment, no dictionary, no checksum: what zstd_encode.hip writes)
        // and the envelope of the index trailer
        q_ok[tid] = 0;
        q_T[tid] = 0;
        if (i < b.n_reads) {
            const uint32_t n = b.src_size[i];
            const bool gated = (b.gate && b.gate[i] >= GATE_SKIP) || n >= E_FIRST;
            if (!gated && n >= 64) {
                const uint8_t* src = b.src + b.src_off[i];
                uint32_t magic, hl = 0;
                __builtin_memcpy(&magic, src, 4);
                const uint32_t fhd = src[4];
                const int fcs_flag = fhd >> 6;
                uint64_t fcs = 0;
                bool ok = magic == 0xFD2FB528u && (fhd & 0x3F) == 0x20;
                if (ok) {
                    const uint32_t fsz = fcs_flag == 0 ? 1u : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
                    for (uint32_t k = 0; k < fsz; ++k) fcs |= (uint64_t)src[5 + k] << (8 * k);
                    if (fsz == 2) fcs += 256;
                    hl = 5 + fsz;
                    ok = fcs <= b.dst_cap[i] && fcs < 0xFFFFFFF0ull;
                }
                uint32_t tb = 0, ns = 0, treeless = 0;
                if (ok) {
                    __builtin_memcpy(&tb, src + n - 4, 4);
                    ok = tb >= 32 && (tb & 7u) == 0 && (uint64_t)tb + hl + 8 <= n;
                }
                if (ok) {
                    uint32_t m[3];
                    __builtin_memcpy(m, src + n - tb, 12);
                    ns = m[2] & 0x7FFFFFFFu;
                    treeless = m[2] >> 31;   // the writer's hint that spans lean on earlier trees (a wrong hint costs time, nothing else)
                    ok = m[0] == IDX_MAGIC && m[1] == tb - 8 && ns >= 2 && tb == 16 + 8 * ns && ns <= fcs / DSPAN_MIN_CONTENT + 4;
                }
                q_ok[tid] = ok ? 1u : 0u;
                q_T[tid] = treeless ? 0x80000000u : 0u;
                q_ns[tid] = ns;
                q_tb[tid] = tb;
                q_hl[tid] = hl;
                q_fcs[tid] = (uint32_t)fcs;
            }
        }
        __syncthreads();
        // (2) all threads: the entries of each index are strictly increasing in the frame and in the content, lie inside
        // the frame, and the first one is the first block
        for (uint32_t q = 0; q < here; ++q) {
            if (!q_ok[q]) continue;
            const uint32_t n = b.src_size[base + q], tb = q_tb[q], ns = q_ns[q];
            const uint8_t* e = b.src + b.src_off[base + q] + n - tb + 12;
            bool bad = false;
            for (uint32_t j = tid; j < ns; j += 1024) {
                uint32_t v[2], pv[2] = { 0, 0 };
                __builtin_memcpy(v, e + 8 * j, 8);
                if (j) __builtin_memcpy(pv, e + 8 * (j - 1), 8);
                bool ok = j == 0 ? (v[0] == q_hl[q] && v[1] == 0) : (v[0] > pv[0] && v[1] > pv[1]);
                ok = ok && (uint64_t)v[0] + 3 <= n - tb && v[1] < q_fcs[q];
                bad |= !ok;
                if (ok && (q_T[q] & 0x80000000u)) {   // does the span's first block bring a tree?  (compressed block, Compressed_Literals_Block)
                    uint32_t h4;
                    __builtin_memcpy(&h4, b.src + b.src_off[base + q] + v[0], 4);   // (the fourth byte: inside the frame or its trailer)
                    if (((h4 >> 1) & 3u) == 2u && ((h4 >> 24) & 3u) == 2u) atomicMax(&q_T[q], 0x80000000u | (j + 1));
                }
            }
            if (bad) q_ok[q] = 0;  // (racing stores of the same value)
        }
        __syncthreads();
        // (3) span counts -> first span of every read
        const uint32_t cnt = i < b.n_reads ? (q_ok[tid] ? q_ns[tid] : 1u) : 0u;
        const uint32_t ci = wave_incl_scan_u32(cnt);
        if (lane == 63) wcnt[w] = ci;
        __syncthreads();
        uint32_t pc = carry_c;
        for (int k = 0; k < w; ++k) pc += wcnt[k];
        if (i < b.n_reads) {
            const uint32_t si = pc + ci - cnt;
            dspan_first[i] = si;
            q_first[tid] = si;
            if (si + cnt > max_spans) {  // cannot happen with the host's bound: the read is left to the second launch, and the
                q_ok[tid] = 2;           // span launch stops in front of its (unwritten) descriptors
                atomicMin(&limit_s, si);
            }
        }
        __syncthreads();
        // (4) all threads: the descriptors
        for (uint32_t q = 0; q < here; ++q) {
            const uint32_t si = q_first[q];
            if (q_ok[q] == 2) continue;
            if (!q_ok[q]) {
                if (tid == 0 && si < max_spans) {
                    DecSpan d = {};
                    d.read = base + q;
                    d.flags = DSPAN_WHOLE | DSPAN_FIRST | DSPAN_LAST;
                    spans[si] = d;
                    dspan_status[4 * si] = 0;
                }
                continue;
            }
            const uint32_t n = b.src_size[base + q], tb = q_tb[q], ns = q_ns[q];
            const uint8_t* e = b.src + b.src_off[base + q] + n - tb + 12;
            for (uint32_t j = tid; j < ns; j += 1024) {
                uint32_t v[2], nx[2] = { 0, 0 };
                __builtin_memcpy(v, e + 8 * j, 8);
                if (j + 1 < ns) __builtin_memcpy(nx, e + 8 * (j + 1), 8);
                DecSpan d;
                d.read = base + q;
                d.src_pos = v[0];
                d.src_end = nx[0];
                d.dst_pos = v[1];
                d.flags = (j == 0 ? DSPAN_FIRST : 0u) | (j + 1 == ns ? DSPAN_LAST : 0u);
                d.ord = j;
                d.tree_pos = 0;
                d.dst_len = (j + 1 < ns ? nx[1] : q_fcs[q]) - v[1];
                const uint32_t T1 = q_T[q] & 0x7FFFFFFFu;
                if (T1 != 0 && j + 1 > T1) __builtin_memcpy(&d.tree_pos, e + 8 * (T1 - 1), 4);   // (never 0: behind the frame header)
                spans[si + j] = d;
                dspan_status[4 * (si + j)] = 0;
            }
        }
        __syncthreads();
        if (tid == 1023) carry_c = pc + ci;
        __syncthreads();
    }
    if (tid == 0) {
        dspan_first[b.n_reads] = carry_c;
        *dspan_count = carry_c < limit_s ? carry_c : limit_s;
    }
}

// one workgroup per read: did the spans work out?  redo[r] = 1 sends the frame to the ordinary one-wavefront decoder.
__global__ __launch_bounds__(256) void zstd_dspan_finish_kernel(ReadBatch b, const DecSpan* spans, const uint32_t* dspan_first, uint32_t max_spans,
                                                                const uint32_t* dspan_status, uint32_t* redo)
{
    __shared__ uint32_t bad_s;
    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t s0 = dspan_first[r], s1 = dspan_first[r + 1];
    if (tid == 0) bad_s = 0;
    __syncthreads();
    if (s1 <= s0 || s1 > max_spans) {
        if (tid == 0) redo[r] = 1;
        return;
    }
    if (spans[s0].flags & DSPAN_WHOLE) {  // the ordinary decoder has already given its verdict
        if (tid == 0) redo[r] = 0;
        return;
    }
    bool bad = false;
    const uint32_t tree_pos = spans[s1 - 1].tree_pos;   // (the last span has one iff any span has)
    for (uint32_t k = s0 + tid; k < s1; k += 256) {
        bool ok = dspan_status[4 * k] == 1;
        const uint32_t st3 = dspan_status[4 * k + 3];
        // the span ended where the next one starts, and left repeat offset 1 at 1 (what a later zero-run span assumed)
        if (ok && k + 1 < s1)
            ok = dspan_status[4 * k + 1] == spans[k + 1].src_pos && dspan_status[4 * k + 2] == spans[k + 1].dst_pos && (st3 & DSPAN_ST_REP1) != 0;
        // the tree the spans behind T were given is the one in force there: T brought it in its first block and nothing after it brought another
        if (ok && tree_pos) {
            const uint32_t trees = st3 & (DSPAN_ST_TREE_FIRST | DSPAN_ST_TREE_LATER);
            if (spans[k].src_pos == tree_pos) ok = trees == DSPAN_ST_TREE_FIRST;
            else if (spans[k].src_pos > tree_pos) ok = trees == 0;
        }
        bad |= !ok;
    }
    if (bad) bad_s = 1;
    __syncthreads();
    if (tid == 0) {
        if (!bad_s) b.result[r] = dspan_status[4 * (s1 - 1) + 2];  // == the frame content size (checked by the last span)
        redo[r] = bad_s;
    }
}

}  // namespace

hipError_t launch_zstd_decode(const ReadBatch& b, uint32_t toosmall_code, unsigned long long* dbg, const void* seq_dtables,
                              hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    const SvbFuse none = { nullptr, nullptr, nullptr };
#ifdef VBZ_EXPERIMENTS   // the timed instantiation (phase cycle counters) is part of the experiments build only
    if (dbg) {
        hipLaunchKernelGGL((zstd_decode_kernel<true, false>), dim3(b.n_reads), dim3(WAVE), 0, s, b, toosmall_code, dbg,
                           reinterpret_cast<const SeqDTables*>(seq_dtables), nullptr, nullptr, nullptr, nullptr, none, RefChains());
        return hipGetLastError();
    }
#endif
    (void)dbg;
    hipLaunchKernelGGL((zstd_decode_kernel<false, false>), dim3(b.n_reads), dim3(WAVE), 0, s, b, toosmall_code, nullptr,
                       reinterpret_cast<const SeqDTables*>(seq_dtables), nullptr, nullptr, nullptr, nullptr, none, RefChains());
    return hipGetLastError();
}

hipError_t launch_zstd_decode_only(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, const uint32_t* only, RefChains chains,
                                   unsigned long long* dbg, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    const SvbFuse none = { nullptr, nullptr, nullptr };
#ifdef VBZ_EXPERIMENTS
    if (dbg) {
        hipLaunchKernelGGL((zstd_decode_kernel<true, false>), dim3(b.n_reads), dim3(WAVE), 0, s, b, toosmall_code, dbg,
                           reinterpret_cast<const SeqDTables*>(seq_dtables), nullptr, nullptr, nullptr, only, none, chains);
        return hipGetLastError();
    }
#endif
    (void)dbg;
    hipLaunchKernelGGL((zstd_decode_kernel<false, false>), dim3(b.n_reads), dim3(WAVE), 0, s, b, toosmall_code, nullptr,
                       reinterpret_cast<const SeqDTables*>(seq_dtables), nullptr, nullptr, nullptr, only, none, chains);
    return hipGetLastError();
}

#ifdef VBZ_EXPERIMENTS   // the svb decoder on the frame's wavefront: measured slower (profiles/r03_fused_svb_decode.md), kept for tools/ only
hipError_t launch_zstd_decode_svb_i16zz(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, uint8_t* out, const uint64_t* out_off,
                                        const uint32_t* out_size, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    const SvbFuse fuse = { out, out_off, out_size };
    hipLaunchKernelGGL((zstd_decode_kernel<false, true>), dim3(b.n_reads), dim3(WAVE), 0, s, b, toosmall_code, nullptr,
                       reinterpret_cast<const SeqDTables*>(seq_dtables), nullptr, nullptr, nullptr, nullptr, fuse, RefChains());
    return hipGetLastError();
}
#endif

// ---- span mode (few, large reads) ------------------------------------------------------------------------------------------
size_t zstd_dspan_desc_bytes() { return sizeof(DecSpan); }

uint32_t zstd_dspan_max_spans(uint64_t content_bytes, uint32_t n_reads)
{
    const uint64_t v = content_bytes / DSPAN_MIN_CONTENT + 5ull * n_reads + 1;
    return v > 0x7FFFFFF0ull ? 0u : (uint32_t)v;
}

hipError_t launch_zstd_decode_spans(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, void* dspan_desc, uint32_t* dspan_first,
                                    uint32_t* dspan_count, uint32_t max_spans, uint32_t* dspan_status, uint32_t* redo, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    DecSpan* spans = reinterpret_cast<DecSpan*>(dspan_desc);
    const SeqDTables* dt = reinterpret_cast<const SeqDTables*>(seq_dtables);
    hipLaunchKernelGGL(zstd_dspan_plan_kernel, dim3(1), dim3(1024), 0, s, b, max_spans, spans, dspan_first, dspan_count, dspan_status);
    const SvbFuse none = { nullptr, nullptr, nullptr };
    hipLaunchKernelGGL((zstd_decode_kernel<false, false, true>), dim3(max_spans), dim3(WAVE), 0, s, b, toosmall_code, nullptr, dt, spans, dspan_count, dspan_status,
                       nullptr, none, RefChains());
    hipLaunchKernelGGL(zstd_dspan_finish_kernel, dim3(b.n_reads), dim3(256), 0, s, b, spans, dspan_first, max_spans, dspan_status, redo);
    hipLaunchKernelGGL((zstd_decode_kernel<false, false>), dim3(b.n_reads), dim3(WAVE), 0, s, b, toosmall_code, nullptr, dt, nullptr, nullptr, nullptr, redo, none, RefChains());
    return hipGetLastError();
}

// host: decoding tables of the predefined LL / ML distributions (RFC 8878 3.1.1.3.2.2), built the way
// fse_build builds them on the device, in the compact form of SeqDTables (uploaded once per context)
size_t seq_dtables_bytes() { return sizeof(SeqDTables); }

void seq_dtables_build(void* host_buffer)
{
    static const int16_t ll_norm[36] = { 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1 };
    static const int16_t ml_norm[53] = { 1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                         1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1 };
    static const uint32_t ll_base[36] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64,
                                          128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536 };
    static const uint8_t ll_bits[36] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };
    static const uint32_t ml_base[53] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
                                          29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                          4099, 8195, 16387, 32771, 65539 };
    static const uint8_t ml_bits[53] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };
    SeqDTables* t = static_cast<SeqDTables*>(host_buffer);
    auto build = [](uint2* out, const int16_t* norm, int nsym, const uint32_t* base, const uint8_t* bits) {
        const int log = 6, size = 1 << log, mask = size - 1;
        int sym[64], next[64], high = size - 1;
        for (int s = 0; s < nsym; ++s) {
            if (norm[s] == -1) {
                sym[high--] = s;
                next[s] = 1;
            } else {
                next[s] = norm[s];
            }
        }
        const int step = (size >> 1) + (size >> 3) + 3;
        int pos = 0;
        for (int s = 0; s < nsym; ++s)
            for (int i = 0; i < norm[s]; ++i) {
                sym[pos] = s;
                pos = (pos + step) & mask;
                while (pos > high) pos = (pos + step) & mask;
            }
        for (int u = 0; u < size; ++u) {
            const int s = sym[u], ns = next[s]++;
            int hb = 0;
            while ((2 << hb) <= ns) ++hb;
            const int nb = log - hb;
            const uint32_t nbase = (uint32_t)((ns << nb) - size);
            out[u] = make_uint2(base[s], nbase | ((uint32_t)bits[s] << 16) | ((uint32_t)nb << 24));
        }
    };
    build(t->ll, ll_norm, 36, ll_base, ll_bits);
    build(t->ml, ml_norm, 53, ml_base, ml_bits);
}

}  // namespace vbzhip
