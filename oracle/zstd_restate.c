/*
 * zstd_restate.c -- CPU ORACLE (test infrastructure): restatement of the zstd frame DECODER.
 *
 * The reference calls ZSTD_getFrameContentSize/ZSTD_decompress (vbz/vbz.cpp:236-273) from the
 * external dependency facebook/zstd (conan pin zstd/1.4.8, reference CMakeLists.txt:92-93), whose
 * source is absent from the reference tree.  The decoder is fully specified by RFC 8878
 * ("Zstandard Compression and the application/zstd Media Type"); this file restates that
 * published algorithm in plain C.  It is validated against the pinned libzstd binary itself
 * (tests/test_oracle_zstd.py: thousands of frames produced by libzstd at levels 1..19 must decode
 * to the same bytes) and is the serial mirror of the HIP decoder in
 * vbz_compression_amd/csrc/zstd_decode.hip (same structure, same error conditions).
 *
 * Supported: single frames, single-segment or windowed, raw/RLE/compressed blocks, literals
 * raw/RLE/Huffman(1 or 4 streams)/treeless, FSE-compressed or direct Huffman weights, sequences in
 * predefined/RLE/FSE/repeat modes, repeat offsets, content checksum (skipped, not verified),
 * concatenated frames and skippable frames.  Dictionaries are not supported (dictID != 0 fails).
 */
#include "vbz_oracle.h"

#include <string.h>

#define ZR_ERR ((size_t)-1)
#define ZR_BLOCK_MAX (128u << 10)

typedef struct {
    uint8_t symbol;
    uint8_t nbits;
    uint16_t base; /* newState base */
} fse_entry;

typedef struct {
    fse_entry e[512];
    int log;
} fse_table;

typedef struct {
    uint16_t e[1 << 12]; /* symbol | nbits << 8 */
    int log;
    int valid;
} huf_table;

/* backward bit reader (RFC 8878 4.1 "Huffman-coded streams", 4.2 "FSE bitstreams") */
typedef struct {
    const uint8_t* p;
    int64_t pos; /* number of unread bits: bits [0,pos) of the little-endian integer p[0..] */
} bitr;

static int bitr_init(bitr* b, const uint8_t* p, size_t n)
{
    if (n == 0 || p[n - 1] == 0) return -1;
    int hb = 7;
    while (!((p[n - 1] >> hb) & 1)) --hb;
    b->p = p;
    b->pos = (int64_t)(n - 1) * 8 + hb;
    return 0;
}

/* peek the next nb bits (as the most significant of the unread bits); bits below 0 read as 0 */
static uint32_t bitr_peek(const bitr* b, int nb)
{
    uint32_t v = 0;
    for (int i = 0; i < nb; ++i) {
        int64_t bit = b->pos - 1 - i;
        uint32_t x = 0;
        if (bit >= 0) x = (b->p[bit >> 3] >> (bit & 7)) & 1u;
        v = (v << 1) | x;
    }
    return v;
}

static uint32_t bitr_read(bitr* b, int nb)
{
    uint32_t v = bitr_peek(b, nb);
    b->pos -= nb;
    return v;
}

static int highbit(uint32_t v)
{
    int r = 0;
    while (v >>= 1) ++r;
    return r;
}

/* --- FSE table description, RFC 8878 4.1.1 ------------------------------------------------- */
/* returns bytes consumed or -1; norm[] gets the normalized counts (-1 = "less than 1") */
static int fse_read_ncount(const uint8_t* p, size_t n, int16_t* norm, int max_symbol, int max_log, int* out_log, int* out_nsym)
{
    if (n < 1) return -1;
    uint64_t bitpos = 0;
    /* forward little-endian bit reader; reading past the end yields zeros (checked afterwards) */
    int log = (int)(p[0] & 0xF) + 5;
    bitpos = 4;
    if (log > max_log) return -1;
    int remaining = (1 << log) + 1;
    int threshold = 1 << log;
    int nbits = log + 1;
    int sym = 0;
    int prev0 = 0;
    while (remaining > 1 && sym <= max_symbol) {
        if (prev0) {
            /* repeat flags: 2 bits each, 3 means "3 more zeros and continue" */
            for (;;) {
                uint32_t r = 0;
                for (int i = 0; i < 2; ++i) {
                    uint64_t bp = bitpos + (unsigned)i;
                    uint32_t bit = (bp >> 3) < n ? (p[bp >> 3] >> (bp & 7)) & 1u : 0u;
                    r |= bit << i;
                }
                bitpos += 2;
                for (uint32_t k = 0; k < r; ++k) {
                    if (sym > max_symbol) return -1;
                    norm[sym++] = 0;
                }
                if (r != 3) break;
            }
            prev0 = 0;
            if (sym > max_symbol) break; /* nothing left to describe: checked below */
            continue;
        }
        int max = (2 * threshold - 1) - remaining;
        uint32_t v = 0;
        for (int i = 0; i < nbits; ++i) {
            uint64_t bp = bitpos + (unsigned)i;
            uint32_t bit = (bp >> 3) < n ? (p[bp >> 3] >> (bp & 7)) & 1u : 0u;
            v |= bit << i;
        }
        int count;
        if ((int)(v & (uint32_t)(threshold - 1)) < max) {
            count = (int)(v & (uint32_t)(threshold - 1));
            bitpos += (unsigned)(nbits - 1);
        } else {
            count = (int)(v & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            bitpos += (unsigned)nbits;
        }
        count--; /* value 0 means probability "-1" */
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = (count == 0);
        while (remaining < threshold) {
            nbits--;
            threshold >>= 1;
        }
    }
    if (remaining != 1) return -1;
    if (sym > max_symbol + 1) return -1;
    size_t used = (size_t)((bitpos + 7) >> 3);
    if (used > n) return -1;
    *out_log = log;
    *out_nsym = sym;
    return (int)used;
}

/* --- FSE decoding table, RFC 8878 4.1.1 "from normalized distribution to decoding tables" -- */
static int fse_build(fse_table* t, const int16_t* norm, int nsym, int log)
{
    int size = 1 << log;
    uint16_t next[256];
    int high = size - 1;
    for (int s = 0; s < nsym; ++s) {
        if (norm[s] == -1) {
            t->e[high--].symbol = (uint8_t)s;
            next[s] = 1;
        } else {
            next[s] = (uint16_t)norm[s];
        }
    }
    int step = (size >> 1) + (size >> 3) + 3;
    int mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; ++s) {
        for (int i = 0; i < norm[s]; ++i) {
            t->e[pos].symbol = (uint8_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    }
    if (pos != 0) return -1;
    for (int u = 0; u < size; ++u) {
        int s = t->e[u].symbol;
        uint32_t ns = next[s]++;
        int nb = log - highbit(ns);
        t->e[u].nbits = (uint8_t)nb;
        t->e[u].base = (uint16_t)((ns << nb) - (uint32_t)size);
    }
    t->log = log;
    return 0;
}

static void fse_build_rle(fse_table* t, uint8_t symbol)
{
    t->e[0].symbol = symbol;
    t->e[0].nbits = 0;
    t->e[0].base = 0;
    t->log = 0;
}

/* --- Huffman tree description, RFC 8878 4.2.1 --------------------------------------------- */
static int huf_read_table(huf_table* h, const uint8_t* p, size_t n)
{
    uint8_t w[256];
    int nw = 0;
    size_t used;
    if (n < 1) return -1;
    int hb = p[0];
    if (hb >= 128) { /* direct 4-bit weights */
        nw = hb - 127;
        used = 1 + (size_t)(nw + 1) / 2;
        if (used > n) return -1;
        for (int i = 0; i < nw; ++i) w[i] = (i & 1) ? (p[1 + i / 2] & 0xF) : (p[1 + i / 2] >> 4);
    } else { /* FSE-compressed weights, two interleaved states, max accuracy log 6 */
        used = 1 + (size_t)hb;
        if (hb == 0 || used > n) return -1;
        int16_t norm[256];
        int log, nsym;
        /* the weights' alphabet ends at 11 (HUF_TABLELOG_MAX - 1): libzstd >= 1.4.7 sizes its workspace for that and refuses a
         * description that lists a symbol beyond it, even one no weight ever takes (found by tools/soak_corrupt.py in round 5:
         * such a description can still be a consistent tree) */
        int hdr = fse_read_ncount(p + 1, (size_t)hb, norm, 11, 6, &log, &nsym);
        if (hdr < 0) return -1;
        fse_table t;
        if (fse_build(&t, norm, nsym, log) != 0) return -1;
        bitr b;
        if (bitr_init(&b, p + 1 + hdr, (size_t)hb - (size_t)hdr) != 0) return -1;
        uint32_t s1 = bitr_read(&b, log);
        uint32_t s2 = bitr_read(&b, log);
        if (b.pos < 0) return -1;
        for (;;) {
            if (nw > 253) return -1;
            w[nw++] = t.e[s1].symbol;
            s1 = t.e[s1].base + bitr_read(&b, t.e[s1].nbits);
            if (b.pos < 0) { w[nw++] = t.e[s2].symbol; break; }
            if (nw > 253) return -1;
            w[nw++] = t.e[s2].symbol;
            s2 = t.e[s2].base + bitr_read(&b, t.e[s2].nbits);
            if (b.pos < 0) { w[nw++] = t.e[s1].symbol; break; }
        }
    }
    /* last weight is implicit */
    uint32_t total = 0;
    for (int i = 0; i < nw; ++i) {
        if (w[i] >= 12) return -1;
        total += w[i] ? (1u << (w[i] - 1)) : 0;
    }
    if (total == 0) return -1;
    int log = highbit(total) + 1;
    if (log > 12) return -1;
    uint32_t rest = (1u << log) - total;
    if (rest & (rest - 1)) return -1; /* must be a power of two */
    w[nw++] = (uint8_t)(highbit(rest) + 1);
    {   /* at least two symbols of weight 1 and an even number of them (complete tree) */
        int r1 = 0;
        for (int i = 0; i < nw; ++i) r1 += (w[i] == 1);
        if (r1 < 2 || (r1 & 1)) return -1;
    }
    /* fill: increasing weight, then increasing symbol value */
    uint32_t idx = 0;
    for (int wt = 1; wt <= log; ++wt) {
        for (int s = 0; s < nw; ++s) {
            if (w[s] != wt) continue;
            uint32_t len = 1u << (wt - 1);
            uint16_t ent = (uint16_t)(s | ((log + 1 - wt) << 8));
            for (uint32_t k = 0; k < len; ++k) h->e[idx + k] = ent;
            idx += len;
        }
    }
    h->log = log;
    h->valid = 1;
    return (int)used;
}

/* test helper: code length of every symbol described by a Huffman tree description (0 = absent) */
int vbo_debug_huf_lengths(const uint8_t* p, size_t n, uint8_t* nbits_out /*256*/, int* used_out)
{
    static _Thread_local huf_table h;
    int used = huf_read_table(&h, p, n);
    if (used < 0) return -1;
    memset(nbits_out, 0, 256);
    for (int i = 0; i < (1 << h.log); ++i) nbits_out[h.e[i] & 0xFF] = (uint8_t)(h.e[i] >> 8);
    if (used_out) *used_out = used;
    return h.log;
}

static int huf_decode_stream(const huf_table* h, const uint8_t* p, size_t n, uint8_t* out, size_t count)
{
    bitr b;
    if (bitr_init(&b, p, n) != 0) return -1;
    for (size_t i = 0; i < count; ++i) {
        uint16_t e = h->e[bitr_peek(&b, h->log)];
        out[i] = (uint8_t)e;
        b.pos -= e >> 8;
    }
    return b.pos == 0 ? 0 : -1; /* must be consumed exactly */
}

/* --- sequences ---------------------------------------------------------------------------- */
static const int16_t LL_DEFAULT[36] = { 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                        2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1 };
static const int16_t ML_DEFAULT[53] = { 1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                        1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1 };
static const int16_t OF_DEFAULT[29] = { 1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1 };
static const uint32_t LL_BASE[36] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                      20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536 };
static const uint8_t LL_BITS[36] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                     1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };
static const uint32_t ML_BASE[53] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                      21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                      43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539 };
static const uint8_t ML_BITS[53] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                     0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };

typedef struct {
    huf_table huf;
    fse_table ll, of, ml;
    int have_ll, have_of, have_ml;
    uint32_t rep[3];
} frame_ctx;

static int seq_table(fse_table* t, int* have, int mode, const uint8_t** pp, const uint8_t* end, const int16_t* def, int def_n,
                     int def_log, int max_sym, int max_log)
{
    const uint8_t* p = *pp;
    switch (mode) {
    case 0:
        if (fse_build(t, def, def_n, def_log) != 0) return -1;
        *have = 1;
        return 0;
    case 1:
        if (p >= end) return -1;
        if (*p > max_sym) return -1;
        fse_build_rle(t, *p);
        *pp = p + 1;
        *have = 1;
        return 0;
    case 2: {
        int16_t norm[64];
        int log, nsym;
        int used = fse_read_ncount(p, (size_t)(end - p), norm, max_sym, max_log, &log, &nsym);
        if (used < 0) return -1;
        if (fse_build(t, norm, nsym, log) != 0) return -1;
        *pp = p + used;
        *have = 1;
        return 0;
    }
    default: return *have ? 0 : -1; /* repeat */
    }
}

/* decode one compressed block into out[base..]; `base` bytes of history precede it in out.
 * Bytes at positions >= cap are not stored (but the block is still fully validated). */
static size_t decode_block(frame_ctx* fc, const uint8_t* src, size_t n, uint8_t* out, size_t base, size_t cap, uint8_t* lit)
{
    if (n < 1) return ZR_ERR;
    /* ---- literals section header, RFC 8878 3.1.1.3.1.1 */
    int type = src[0] & 3, fmt = (src[0] >> 2) & 3;
    size_t hsz, regen, csize = 0;
    int streams = 1;
    if (type < 2) {
        if (fmt == 0 || fmt == 2) { hsz = 1; regen = src[0] >> 3; }
        else if (fmt == 1) { hsz = 2; if (n < 2) return ZR_ERR; regen = (src[0] >> 4) | ((size_t)src[1] << 4); }
        else { hsz = 3; if (n < 3) return ZR_ERR; regen = (src[0] >> 4) | ((size_t)src[1] << 4) | ((size_t)src[2] << 12); }
    } else {
        if (fmt < 2) {
            hsz = 3;
            if (n < 3) return ZR_ERR;
            uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16);
            regen = (v >> 4) & 0x3FF;
            csize = v >> 14;
            streams = fmt == 0 ? 1 : 4;
        } else if (fmt == 2) {
            hsz = 4;
            if (n < 4) return ZR_ERR;
            uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
            regen = (v >> 4) & 0x3FFF;
            csize = v >> 18;
            streams = 4;
        } else {
            hsz = 5;
            if (n < 5) return ZR_ERR;
            uint64_t v = src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24) |
                         ((uint64_t)src[4] << 32);
            regen = (size_t)((v >> 4) & 0x3FFFF);
            csize = (size_t)(v >> 22);
            streams = 4;
        }
    }
    if (regen > ZR_BLOCK_MAX) return ZR_ERR;
    const uint8_t* p = src + hsz;
    const uint8_t* end = src + n;
    if (type == 0) {
        if ((size_t)(end - p) < regen) return ZR_ERR;
        memcpy(lit, p, regen);
        p += regen;
    } else if (type == 1) {
        if (p >= end) return ZR_ERR;
        memset(lit, *p, regen);
        p += 1;
    } else {
        if ((size_t)(end - p) < csize) return ZR_ERR;
        if (regen == 0 || csize == 0) return ZR_ERR;
        const uint8_t* q = p;
        const uint8_t* qend = p + csize;
        if (type == 2) {
            int used = huf_read_table(&fc->huf, q, csize);
            if (used < 0) return ZR_ERR;
            q += used;
        } else if (!fc->huf.valid) {
            return ZR_ERR;
        }
        if (streams == 1) {
            if (huf_decode_stream(&fc->huf, q, (size_t)(qend - q), lit, regen) != 0) return ZR_ERR;
        } else {
            if (qend - q < 10) return ZR_ERR;
            size_t s1 = q[0] | ((size_t)q[1] << 8), s2 = q[2] | ((size_t)q[3] << 8), s3 = q[4] | ((size_t)q[5] << 8);
            q += 6;
            size_t tot = (size_t)(qend - q);
            if (s1 + s2 + s3 > tot) return ZR_ERR;
            size_t s4 = tot - s1 - s2 - s3;
            size_t seg = (regen + 3) / 4;
            if (seg * 3 > regen) return ZR_ERR;
            if (huf_decode_stream(&fc->huf, q, s1, lit, seg) != 0) return ZR_ERR;
            if (huf_decode_stream(&fc->huf, q + s1, s2, lit + seg, seg) != 0) return ZR_ERR;
            if (huf_decode_stream(&fc->huf, q + s1 + s2, s3, lit + 2 * seg, seg) != 0) return ZR_ERR;
            if (huf_decode_stream(&fc->huf, q + s1 + s2 + s3, s4, lit + 3 * seg, regen - 3 * seg) != 0) return ZR_ERR;
        }
        p = qend;
    }
    /* ---- sequences section header, RFC 8878 3.1.1.3.2.1 */
    if (p >= end) return ZR_ERR;
    size_t nseq = *p++;
    if (nseq >= 128) {
        if (nseq == 255) {
            if (end - p < 2) return ZR_ERR;
            nseq = (size_t)p[0] + ((size_t)p[1] << 8) + 0x7F00;
            p += 2;
        } else {
            if (end - p < 1) return ZR_ERR;
            nseq = ((nseq - 128) << 8) + *p++;
        }
    }
    size_t opos = base; /* logical output position */
    size_t lpos = 0;
#define PUT(byte_expr)                                    \
    do {                                                  \
        uint8_t bb__ = (byte_expr);                       \
        if (opos < cap) out[opos] = bb__;                 \
        ++opos;                                           \
    } while (0)
    if (nseq == 0) {
        if (p != end) return ZR_ERR;
        for (size_t i = 0; i < regen; ++i) PUT(lit[i]);
        if (opos - base > ZR_BLOCK_MAX) return ZR_ERR;
        return opos - base;
    }
    if (p >= end) return ZR_ERR;
    int modes = *p++;
    if (modes & 3) return ZR_ERR;
    if (seq_table(&fc->ll, &fc->have_ll, (modes >> 6) & 3, &p, end, LL_DEFAULT, 36, 6, 35, 9) != 0) return ZR_ERR;
    if (seq_table(&fc->of, &fc->have_of, (modes >> 4) & 3, &p, end, OF_DEFAULT, 29, 5, 31, 8) != 0) return ZR_ERR;
    if (seq_table(&fc->ml, &fc->have_ml, (modes >> 2) & 3, &p, end, ML_DEFAULT, 53, 6, 52, 9) != 0) return ZR_ERR;
    bitr b;
    if (bitr_init(&b, p, (size_t)(end - p)) != 0) return ZR_ERR;
    uint32_t sl = bitr_read(&b, fc->ll.log);
    uint32_t so = bitr_read(&b, fc->of.log);
    uint32_t sm = bitr_read(&b, fc->ml.log);
    if (b.pos < 0) return ZR_ERR;
    for (size_t i = 0; i < nseq; ++i) {
        int lc = fc->ll.e[sl].symbol, oc = fc->of.e[so].symbol, mc = fc->ml.e[sm].symbol;
        if (lc > 35 || mc > 52 || oc > 31) return ZR_ERR;
        uint32_t ofv = (1u << oc) + bitr_read(&b, oc);
        uint32_t mlen = ML_BASE[mc] + bitr_read(&b, ML_BITS[mc]);
        uint32_t llen = LL_BASE[lc] + bitr_read(&b, LL_BITS[lc]);
        uint32_t offset;
        if (ofv > 3) {
            offset = ofv - 3;
            fc->rep[2] = fc->rep[1];
            fc->rep[1] = fc->rep[0];
            fc->rep[0] = offset;
        } else {
            uint32_t idx = ofv - 1 + (llen == 0);
            if (idx == 0) {
                offset = fc->rep[0];
            } else {
                offset = idx == 3 ? fc->rep[0] - 1 : fc->rep[idx];
                if (offset == 0) offset = 1; /* libzstd forces an invalid 0 to 1 instead of failing */
                if (idx > 1) fc->rep[2] = fc->rep[1];
                fc->rep[1] = fc->rep[0];
                fc->rep[0] = offset;
            }
        }
        if (i + 1 < nseq) {
            sl = fc->ll.e[sl].base + bitr_read(&b, fc->ll.e[sl].nbits);
            sm = fc->ml.e[sm].base + bitr_read(&b, fc->ml.e[sm].nbits);
            so = fc->of.e[so].base + bitr_read(&b, fc->of.e[so].nbits);
        }
        if (b.pos < 0) return ZR_ERR;
        if (lpos + llen > regen) return ZR_ERR;
        for (uint32_t k = 0; k < llen; ++k) PUT(lit[lpos + k]);
        lpos += llen;
        if (offset > opos) return ZR_ERR;
        for (uint32_t k = 0; k < mlen; ++k) {
            size_t from = opos - offset;
            PUT(from < cap ? out[from] : 0);
        }
        if (opos - base > ZR_BLOCK_MAX) return ZR_ERR;
    }
    if (b.pos != 0) return ZR_ERR;
    for (; lpos < regen;) PUT(lit[lpos++]);
#undef PUT
    if (opos - base > ZR_BLOCK_MAX) return ZR_ERR;
    return opos - base;
}

static size_t decode_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* consumed, uint8_t* lit, frame_ctx* fc)
{
    if (n < 5) return ZR_ERR;
    uint32_t magic = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { /* skippable frame */
        if (n < 8) return ZR_ERR;
        size_t sz = src[4] | ((size_t)src[5] << 8) | ((size_t)src[6] << 16) | ((size_t)src[7] << 24);
        if (n - 8 < sz) return ZR_ERR;
        *consumed = 8 + sz;
        return 0;
    }
    if (magic != 0xFD2FB528u) return ZR_ERR;
    uint8_t fhd = src[4];
    int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did_flag = fhd & 3;
    if (fhd & 0x08) return ZR_ERR; /* reserved bit */
    size_t pos = 5;
    uint64_t window = 0;
    if (!single) {
        if (pos >= n) return ZR_ERR;
        uint8_t wd = src[pos++];
        int wlog = 10 + (wd >> 3);
        if (wlog > 31) return ZR_ERR;
        window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7);
    }
    static const int did_sz[4] = { 0, 1, 2, 4 };
    if (pos + (size_t)did_sz[did_flag] > n) return ZR_ERR;
    uint32_t did = 0;
    for (int i = 0; i < did_sz[did_flag]; ++i) did |= (uint32_t)src[pos + (size_t)i] << (8 * i);
    pos += (size_t)did_sz[did_flag];
    if (did != 0) return ZR_ERR;
    int fcs_sz = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    if (pos + (size_t)fcs_sz > n) return ZR_ERR;
    uint64_t fcs = 0;
    for (int i = 0; i < fcs_sz; ++i) fcs |= (uint64_t)src[pos + (size_t)i] << (8 * i);
    if (fcs_sz == 2) fcs += 256;
    pos += (size_t)fcs_sz;
    if (single) window = fcs;
    uint64_t block_max = window < ZR_BLOCK_MAX ? window : ZR_BLOCK_MAX;
    memset(fc, 0, sizeof(*fc));
    fc->rep[0] = 1;
    fc->rep[1] = 4;
    fc->rep[2] = 8;
    size_t out = 0;
    for (;;) {
        if (pos + 3 > n) return ZR_ERR;
        uint32_t bh = src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        pos += 3;
        int last = bh & 1, btype = (bh >> 1) & 3;
        size_t bsize = bh >> 3;
        if (btype == 3) return ZR_ERR;
        if (btype == 0) {
            if (bsize > block_max) return ZR_ERR;
            if (pos + bsize > n) return ZR_ERR;
            for (size_t i = 0; i < bsize; ++i)
                if (out + i < cap) dst[out + i] = src[pos + i];
            out += bsize;
            pos += bsize;
        } else if (btype == 1) {
            if (bsize > block_max) return ZR_ERR;
            if (pos + 1 > n) return ZR_ERR;
            for (size_t i = 0; i < bsize; ++i)
                if (out + i < cap) dst[out + i] = src[pos];
            out += bsize;
            pos += 1;
        } else {
            if (bsize >= ZR_BLOCK_MAX || pos + bsize > n) return ZR_ERR;
            size_t r = decode_block(fc, src + pos, bsize, dst, out, cap, lit);
            if (r == ZR_ERR) return ZR_ERR;
            if (r > block_max) return ZR_ERR;
            out += r;
            pos += bsize;
        }
        if (last) break;
    }
    if (checksum) {
        if (pos + 4 > n) return ZR_ERR;
        pos += 4; /* xxh64 low 32 bits: not verified by this restatement */
    }
    if (fcs_sz && out != fcs) return ZR_ERR;
    *consumed = pos;
    return out;
}

size_t vbo_zstd_restate_decompress(void* dst, size_t cap, const void* src_, size_t n)
{
    static _Thread_local uint8_t lit[ZR_BLOCK_MAX + 32];
    static _Thread_local frame_ctx fc;
    const uint8_t* src = (const uint8_t*)src_;
    size_t total = 0;
    /* ZSTD_decompress decodes every concatenated frame; an empty input decodes to nothing */
    while (n > 0) {
        size_t used = 0;
        size_t r = decode_frame(src, n, (uint8_t*)dst + total, cap > total ? cap - total : 0, &used, lit, &fc);
        if (r == ZR_ERR) return ZR_ERR;
        if (r > (cap > total ? cap - total : 0)) return ZR_ERR; /* dstSize_tooSmall */
        total += r;
        src += used;
        n -= used;
    }
    return total;
}
