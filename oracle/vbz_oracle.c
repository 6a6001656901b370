/*
 * vbz_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of the reference's VBZ path.  Every function cites the reference
 * file:line it follows (paths relative to the reference checkout).  See vbz_oracle.h for how
 * this restatement is pinned.
 */
#define _GNU_SOURCE
#include "vbz_oracle.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * zstd: the pinned external dependency (facebook/zstd 1.4.8), used through its public one-shot
 * API exactly as the reference calls it (vbz/vbz.cpp:109,194,236,263).
 * ------------------------------------------------------------------------------------------ */
typedef size_t (*zstd_compress_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef size_t (*zstd_bound_fn)(size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
typedef unsigned long long (*zstd_fcs_fn)(const void*, size_t);
typedef const char* (*zstd_version_fn)(void);

static struct {
    int tried;
    void* handle;
    zstd_compress_fn compress;
    zstd_decompress_fn decompress;
    zstd_bound_fn bound;
    zstd_iserror_fn is_error;
    zstd_fcs_fn content_size;
    zstd_version_fn version;
} g_zstd;

static int zstd_load(void)
{
    if (!g_zstd.tried) {
        const char* names[] = { getenv("VBO_LIBZSTD"), "libzstd.so.1", "libzstd.so", NULL, NULL };
        g_zstd.tried = 1;
        for (int i = 0; i < 3 && !g_zstd.handle; ++i) {
            if (names[i] && names[i][0]) g_zstd.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        }
        if (g_zstd.handle) {
            g_zstd.compress = (zstd_compress_fn)dlsym(g_zstd.handle, "ZSTD_compress");
            g_zstd.decompress = (zstd_decompress_fn)dlsym(g_zstd.handle, "ZSTD_decompress");
            g_zstd.bound = (zstd_bound_fn)dlsym(g_zstd.handle, "ZSTD_compressBound");
            g_zstd.is_error = (zstd_iserror_fn)dlsym(g_zstd.handle, "ZSTD_isError");
            g_zstd.content_size = (zstd_fcs_fn)dlsym(g_zstd.handle, "ZSTD_getFrameContentSize");
            g_zstd.version = (zstd_version_fn)dlsym(g_zstd.handle, "ZSTD_versionString");
            if (!g_zstd.compress || !g_zstd.decompress || !g_zstd.bound || !g_zstd.is_error ||
                !g_zstd.content_size || !g_zstd.version) {
                dlclose(g_zstd.handle);
                g_zstd.handle = NULL;
            }
        }
    }
    return g_zstd.handle != NULL;
}

const char* vbo_zstd_version(void) { return zstd_load() ? g_zstd.version() : NULL; }

size_t vbo_zstd_compress(void* dst, size_t cap, const void* src, size_t n, int level)
{
    if (!zstd_load()) return (size_t)-1;
    size_t r = g_zstd.compress(dst, cap, src, n, level);
    return g_zstd.is_error(r) ? (size_t)-1 : r;
}

size_t vbo_zstd_decompress(void* dst, size_t cap, const void* src, size_t n)
{
    if (!zstd_load()) return (size_t)-1;
    size_t r = g_zstd.decompress(dst, cap, src, n);
    return g_zstd.is_error(r) ? (size_t)-1 : r;
}

size_t vbo_zstd_bound(size_t n)
{
    if (zstd_load()) return g_zstd.bound(n);
    /* published formula of ZSTD_compressBound (zstd.h, ZSTD_COMPRESSBOUND) */
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

unsigned long long vbo_zstd_content_size(const void* src, size_t n)
{
    if (!zstd_load()) return (unsigned long long)-2;
    return g_zstd.content_size(src, n);
}

/* ------------------------------------------------------------------------------------------
 * L1 -- streamvbyte (public format of lemire/streamvbyte, scalar restatement)
 *   keys[ceil(N/4)] ++ data; key byte j holds codes of ints 4j..4j+3, 2 bits each LSB first;
 *   code c => c+1 little-endian data bytes.
 * Reference call sites: vbz/v0/vbz_streamvbyte_impl.h:25,49,59 ; size: vbz/v0/vbz_streamvbyte.cpp:17
 * ------------------------------------------------------------------------------------------ */
static uint32_t svb_max_bytes(uint32_t count) { return (count + 3) / 4 + 4 * count; }

static size_t svb_encode_u32(const uint32_t* in, uint32_t count, uint8_t* out)
{
    uint8_t* keys = out;
    uint8_t* data = out + (count + 3) / 4;
    uint32_t key = 0;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t v = in[i];
        uint32_t code = (v > 0xFFu) + (v > 0xFFFFu) + (v > 0xFFFFFFu);
        key |= code << (2 * (i & 3));
        for (uint32_t b = 0; b <= code; ++b) *data++ = (uint8_t)(v >> (8 * b));
        if ((i & 3) == 3) { *keys++ = (uint8_t)key; key = 0; }
    }
    if (count & 3) *keys = (uint8_t)key;
    return (size_t)(data - out);
}

/* streamvbyte_validate_stream: sum of (code+1) over `count` codes must equal the data bytes. */
static bool svb_validate(const uint8_t* in, size_t in_bytes, uint32_t count)
{
    if (in_bytes == 0 || count == 0) return in_bytes == count;
    size_t key_len = ((size_t)count + 3) / 4;
    if (key_len > in_bytes) return false;
    uint64_t need = 0;
    for (uint32_t i = 0; i < count; ++i) need += 1u + ((in[i >> 2] >> (2 * (i & 3))) & 3u);
    return need == in_bytes - key_len;
}

static size_t svb_decode_u32(const uint8_t* in, uint32_t* out, uint32_t count)
{
    const uint8_t* data = in + (count + 3) / 4;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t code = (in[i >> 2] >> (2 * (i & 3))) & 3u;
        uint32_t v = 0;
        for (uint32_t b = 0; b <= code; ++b) v |= (uint32_t)(*data++) << (8 * b);
        out[i] = v;
    }
    return (size_t)(data - in);
}

/* streamvbyte_zigzag.c: 32-bit zig-zag delta (reference call sites vbz/v0/vbz_streamvbyte_impl.h:34,76) */
static uint32_t zigzag32(int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); }
static int32_t unzigzag32(uint32_t v) { return (int32_t)((v >> 1) ^ (0u - (v & 1u))); }

/* ------------------------------------------------------------------------------------------
 * v1 nibble ("half") codec for 1-byte integers (reference vbz/v1/vbz_streamvbyte_impl.h:20-216)
 *   code 0: value 0, no data; 1: one nibble; 2: two nibbles; 3: four nibbles (low 16 bits).
 *   nibbles are appended low-nibble-first into the data bytes.
 * ------------------------------------------------------------------------------------------ */
static size_t half_encode_u32(const uint32_t* in, uint32_t count, uint8_t* out)
{
    uint8_t* keys = out;
    uint8_t* data = out + (count + 3) / 4;
    size_t nib = 0; /* nibbles written */
    uint32_t key = 0;
    if (count == 0) return 0;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t v = in[i];
        uint32_t code = v == 0 ? 0 : (v < 16 ? 1 : (v < 256 ? 2 : 3));
        uint32_t n = (1u << code) >> 1; /* 0,1,2,4 nibbles */
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t q = (v >> (4 * k)) & 0xF;
            if ((nib & 1) == 0) data[nib >> 1] = (uint8_t)q;
            else data[nib >> 1] |= (uint8_t)(q << 4);
            ++nib;
        }
        key |= code << (2 * (i & 3));
        if ((i & 3) == 3) { *keys++ = (uint8_t)key; key = 0; }
    }
    if (count & 3) *keys = (uint8_t)key;
    return (size_t)((data + (nib + 1) / 2) - out);
}

static bool half_validate(const uint8_t* in, size_t in_bytes, uint32_t count)
{
    if (in_bytes == 0 || count == 0) return in_bytes == count;
    size_t key_len = ((size_t)count + 3) / 4;
    if (key_len > in_bytes) return false;
    uint64_t nib = 0;
    for (uint32_t i = 0; i < count; ++i) nib += (1u << ((in[i >> 2] >> (2 * (i & 3))) & 3u)) >> 1;
    return (nib + 1) / 2 == in_bytes - key_len;
}

static size_t half_decode_u32(const uint8_t* in, uint32_t* out, uint32_t count)
{
    const uint8_t* data = in + (count + 3) / 4;
    size_t nib = 0;
    if (count == 0) return 0;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t code = (in[i >> 2] >> (2 * (i & 3))) & 3u;
        uint32_t n = (1u << code) >> 1;
        uint32_t v = 0;
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t q = (data[nib >> 1] >> (4 * (nib & 1))) & 0xF;
            v |= q << (4 * k);
            ++nib;
        }
        out[i] = v;
    }
    return (size_t)((data + (nib + 1) / 2) - in);
}

/* ------------------------------------------------------------------------------------------
 * StreamVByteWorkerV0<int16_t,true>  (the x86 SSSE3 specialisation, the int16 hot path)
 * reference vbz/v0/vbz_streamvbyte_impl_sse3.h:403-659
 * ------------------------------------------------------------------------------------------ */

/* compress: sse3.h:406-466 (+ compress_int_registers :582-609, scalar_to_zig_zag :360-372).
 * delta wraps in int16 (x[-1] = 0); zz = (d<<1)^(d>>15) as u16; code = zz > 255 (never 2 or 3). */
static vbo_size_t i16zz_compress(const uint8_t* src, vbo_size_t src_size, uint8_t* dst)
{
    uint32_t n = src_size / 2;
    if (n == 0) return 0; /* :410-413 */
    uint32_t key_len = (n >> 2) + (((n & 3) + 3) >> 2); /* :415 */
    uint8_t* keys = dst;
    uint8_t* data = dst + key_len;
    int16_t prev = 0;
    uint32_t key = 0;
    for (uint32_t i = 0; i < n; ++i) {
        int16_t x;
        memcpy(&x, src + 2 * (size_t)i, 2);
        int16_t d = (int16_t)(uint16_t)((uint16_t)x - (uint16_t)prev);       /* _mm_sub_epi16 :432 */
        uint16_t zz = (uint16_t)(((uint16_t)d << 1) ^ (uint16_t)(d >> 15)); /* :436-438 */
        prev = x;
        uint32_t code = zz > 0xFF;
        key |= code << (2 * (i & 3));
        *data++ = (uint8_t)zz;
        if (code) *data++ = (uint8_t)(zz >> 8);
        if ((i & 3) == 3) { *keys++ = (uint8_t)key; key = 0; }
    }
    if (n & 3) *keys = (uint8_t)key;
    return (vbo_size_t)(data - dst);
}

/* decompress: sse3.h:468-580.
 * SIMD body (:494-540) runs for 8-value groups while >= 32 data bytes remain; it keeps only the low
 * 16 bits of each decoded value and un-zigzags in 16 bits.  The scalar tail (:542-572) decodes the
 * rest with bounds checks, un-zigzags in 32 bits and truncates.  Leftover data => STREAM_ERROR. */
static vbo_size_t i16zz_decompress(const uint8_t* src, vbo_size_t src_size, uint8_t* dst, vbo_size_t dst_size)
{
    uint32_t count = dst_size / 2;
    if (count == 0) return 0; /* :472-476 */
    uint32_t key_bytes = (count + 3) / 4;
    if (src_size < key_bytes) return VBO_INPUT_SIZE_ERROR; /* :478-482 */
    const uint8_t* keys = src;
    const uint8_t* data = src + key_bytes;
    size_t remaining = src_size - key_bytes;
    uint32_t groups = count / 8;
    uint32_t out_i = 0;
    uint16_t prev = 0;
    uint32_t g = 0;
    for (; g < groups; ++g) {
        if (remaining < 32) break; /* :498-501 */
        for (uint32_t k = 0; k < 8; ++k) {
            uint32_t i = g * 8 + k;
            uint32_t code = (keys[i >> 2] >> (2 * (i & 3))) & 3u;
            uint32_t v = 0;
            for (uint32_t b = 0; b <= code; ++b) v |= (uint32_t)data[b] << (8 * b);
            data += code + 1;
            remaining -= code + 1;
            uint16_t lo = (uint16_t)v;                                        /* to_16_bit shuffles :510-514 */
            uint16_t dz = (uint16_t)((lo >> 1) ^ (uint16_t)(0u - (lo & 1u))); /* :516-521 */
            prev = (uint16_t)(prev + dz);                                     /* :524-538 */
            memcpy(dst + 2 * (size_t)out_i, &prev, 2);
            ++out_i;
        }
    }
    for (uint32_t i = out_i; i < count; ++i) { /* scalar tail :542-572 */
        uint32_t code = (keys[i >> 2] >> (2 * (i & 3))) & 3u;
        if (remaining < code + 1) return VBO_STREAM_ERROR; /* decompress_int :638-642 */
        uint32_t v = 0;
        for (uint32_t b = 0; b <= code; ++b) v |= (uint32_t)data[b] << (8 * b);
        data += code + 1;
        remaining -= code + 1;
        uint32_t dz = (v >> 1) ^ (0u - (v & 1u)); /* zig_zag_to_scalar :374-385, 32-bit then truncate */
        prev = (uint16_t)(dz + prev);
        memcpy(dst + 2 * (size_t)i, &prev, 2);
    }
    if (remaining != 0) return VBO_STREAM_ERROR; /* :574-577 */
    return count * 2;
}

/* ------------------------------------------------------------------------------------------
 * Generic StreamVByteWorkerV0<T,ZigZag> / StreamVByteWorkerV1<int8,ZigZag>
 * reference vbz/v0/vbz_streamvbyte_impl.h:14-104, vbz/v1/vbz_streamvbyte_impl.h:219-301
 * T is always the SIGNED type of the given width (dispatch at vbz/v0/vbz_streamvbyte.cpp:37-62).
 * ------------------------------------------------------------------------------------------ */
static int32_t load_signed(const uint8_t* p, int size)
{
    if (size == 1) return (int8_t)p[0];
    if (size == 2) { int16_t v; memcpy(&v, p, 2); return v; }
    int32_t v; memcpy(&v, p, 4); return v;
}

static void store_trunc(uint8_t* p, int size, uint32_t v)
{
    if (size == 1) p[0] = (uint8_t)v;
    else if (size == 2) { uint16_t t = (uint16_t)v; memcpy(p, &t, 2); }
    else memcpy(p, &v, 4);
}

static vbo_size_t generic_compress(const uint8_t* src, vbo_size_t src_size, uint8_t* dst, int size, bool zigzag, bool half)
{
    uint32_t n = src_size / (uint32_t)size;
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    if (!tmp) return VBO_OUT_OF_MEMORY_ERROR;
    int32_t prev = 0;
    for (uint32_t i = 0; i < n; ++i) {
        int32_t x = load_signed(src + (size_t)i * size, size);
        if (zigzag) { tmp[i] = zigzag32((int32_t)((uint32_t)x - (uint32_t)prev)); prev = x; }
        else tmp[i] = (uint32_t)x;
    }
    size_t r = half ? half_encode_u32(tmp, n, dst) : svb_encode_u32(tmp, n, dst);
    free(tmp);
    return (vbo_size_t)r;
}

static vbo_size_t generic_decompress(const uint8_t* src, vbo_size_t src_size, uint8_t* dst, vbo_size_t dst_size, int size,
                                     bool zigzag, bool half)
{
    uint32_t n = dst_size / (uint32_t)size;
    if (!(half ? half_validate(src, src_size, n) : svb_validate(src, src_size, n))) return VBO_STREAM_ERROR;
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    if (!tmp) return VBO_OUT_OF_MEMORY_ERROR;
    size_t used = half ? half_decode_u32(src, tmp, n) : svb_decode_u32(src, tmp, n);
    if (used != src_size) { free(tmp); return VBO_STREAM_ERROR; }
    uint32_t prev = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = tmp[i];
        if (zigzag) { v = (uint32_t)unzigzag32(v) + prev; prev = v; }
        store_trunc(dst + (size_t)i * size, size, v);
    }
    free(tmp);
    return n * (uint32_t)size;
}

vbo_size_t vbo_max_streamvbyte_size(size_t integer_size, vbo_size_t source_size)
{
    /* vbz/v0/vbz_streamvbyte.cpp:7-18 == vbz/v1/vbz_streamvbyte.cpp:9-20 */
    if (source_size % integer_size != 0) return VBO_INPUT_SIZE_ERROR;
    return (vbo_size_t)svb_max_bytes((uint32_t)(source_size / integer_size));
}

/* the CPU baseline leg of bench.py switches the SSSE3 form of the int16 zig-zag stage on (vbz_oracle_simd.c) */
static int g_simd_svb = 0;
void vbo_use_simd_svb(int on) { g_simd_svb = on && vbo_simd_available(); }

vbo_size_t vbo_streamvbyte_compress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap, int integer_size,
                                    bool zigzag, unsigned version)
{
    (void)dst_cap;
    /* vbz/v0/vbz_streamvbyte.cpp:20-65, vbz/v1/vbz_streamvbyte.cpp:22-65 */
    if (integer_size != 1 && integer_size != 2 && integer_size != 4) return VBO_INTEGER_SIZE_ERROR;
    if (src_size % (uint32_t)integer_size != 0) return VBO_INPUT_SIZE_ERROR;
    if (integer_size == 2 && zigzag) {
        if (g_simd_svb) return vbo_i16zz_compress_simd((const uint8_t*)src, src_size, (uint8_t*)dst);
        return i16zz_compress((const uint8_t*)src, src_size, (uint8_t*)dst);
    }
    return generic_compress((const uint8_t*)src, src_size, (uint8_t*)dst, integer_size, zigzag,
                            version == 1 && integer_size == 1);
}

vbo_size_t vbo_streamvbyte_decompress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_size, int integer_size,
                                      bool zigzag, unsigned version)
{
    /* vbz/v0/vbz_streamvbyte.cpp:67-108, vbz/v1/vbz_streamvbyte.cpp:67-113 */
    if (integer_size != 1 && integer_size != 2 && integer_size != 4) return VBO_INTEGER_SIZE_ERROR;
    if (dst_size % (uint32_t)integer_size != 0) return VBO_DESTINATION_SIZE_ERROR;
    if (integer_size == 2 && zigzag) {
        if (g_simd_svb) {
            const vbo_size_t r = vbo_i16zz_decompress_simd((const uint8_t*)src, src_size, (uint8_t*)dst, dst_size);
            if (r != VBO_SIMD_DECLINED) return r;
        }
        return i16zz_decompress((const uint8_t*)src, src_size, (uint8_t*)dst, dst_size);
    }
    return generic_decompress((const uint8_t*)src, src_size, (uint8_t*)dst, dst_size, integer_size, zigzag,
                              version == 1 && integer_size == 1);
}

/* ------------------------------------------------------------------------------------------
 * L2 -- C API, reference vbz/vbz.cpp (quirks kept: see comments)
 * ------------------------------------------------------------------------------------------ */
static bool valid_int_size(const VboOptions* o) /* vbz.cpp:44-50 */
{
    return o->integer_size == 0 || o->integer_size == 1 || o->integer_size == 2 || o->integer_size == 4;
}

bool vbo_is_error(vbo_size_t v) { return v >= VBO_FIRST_ERROR; } /* vbz.cpp:61-64 */

const char* vbo_error_string(vbo_size_t v) /* vbz.cpp:66-77 */
{
    switch (v) {
    case VBO_ZSTD_ERROR: return "VBZ_ZSTD_ERROR";
    case VBO_INPUT_SIZE_ERROR: return "VBZ_INPUT_SIZE_ERROR";
    case VBO_INTEGER_SIZE_ERROR: return "VBZ_INTEGER_SIZE_ERROR";
    case VBO_DESTINATION_SIZE_ERROR: return "VBZ_DESTINATION_SIZE_ERROR";
    case VBO_STREAM_ERROR: return "VBZ_STREAMVBYTE_STREAM_ERROR";
    case VBO_VERSION_ERROR: return "VBZ_VERSION_ERROR";
    case VBO_OUT_OF_MEMORY_ERROR: return "VBZ_OUT_OF_MEMORY_ERROR";
    default: return "VBZ_UNKNOWN_ERROR";
    }
}

vbo_size_t vbo_max_compressed_size(vbo_size_t source_size, const VboOptions* o) /* vbz.cpp:79-114 */
{
    if (!valid_int_size(o)) return VBO_INTEGER_SIZE_ERROR;
    vbo_size_t max_size = source_size;
    if (o->integer_size != 0) {
        if (o->vbz_version > 1) return VBO_VERSION_ERROR;
        max_size = vbo_max_streamvbyte_size(o->integer_size, max_size);
        if (vbo_is_error(max_size)) return max_size;
    }
    if (o->zstd_compression_level != 0) max_size = (vbo_size_t)vbo_zstd_bound(max_size);
    return max_size + 4; /* always include the sized header, :112-113 */
}

vbo_size_t vbo_compress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap, const VboOptions* o)
{
    /* vbz.cpp:116-208 */
    if (!valid_int_size(o)) return VBO_INTEGER_SIZE_ERROR;
    if (o->zstd_compression_level == 0 && o->integer_size == 0) {
        if (src_size > dst_cap) return VBO_DESTINATION_SIZE_ERROR; /* copy_buffer :32-42 */
        memcpy(dst, src, src_size);
        return src_size;
    }
    const void* cur = src;
    vbo_size_t cur_size = src_size;
    void* scratch = NULL;
    if (o->integer_size != 0) {
        if (o->vbz_version > 1) return VBO_VERSION_ERROR;
        vbo_size_t max_svb = vbo_max_streamvbyte_size(o->integer_size, src_size);
        if (vbo_is_error(max_svb)) return max_svb;
        void* svb_dst = dst;
        vbo_size_t svb_cap = dst_cap;
        if (o->zstd_compression_level != 0) {
            scratch = malloc(max_svb ? max_svb : 1);
            if (!scratch) return VBO_OUT_OF_MEMORY_ERROR;
            svb_dst = scratch;
            svb_cap = max_svb;
        } else if (max_svb > dst_cap) {
            return VBO_DESTINATION_SIZE_ERROR;
        }
        /* return value is not checked by the reference (:176-185) */
        cur_size = vbo_streamvbyte_compress(src, src_size, svb_dst, svb_cap, (int)o->integer_size,
                                            o->perform_delta_zig_zag, o->vbz_version);
        cur = svb_dst;
    }
    if (o->zstd_compression_level == 0) return cur_size;
    size_t r = vbo_zstd_compress(dst, dst_cap, cur, cur_size, (int)o->zstd_compression_level);
    free(scratch);
    if (r == (size_t)-1) return VBO_ZSTD_ERROR;
    return (vbo_size_t)r;
}

vbo_size_t vbo_decompress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_size, const VboOptions* o)
{
    /* vbz.cpp:210-300 */
    if (!valid_int_size(o)) return VBO_INTEGER_SIZE_ERROR;
    if (o->zstd_compression_level == 0 && o->integer_size == 0) {
        if (src_size > dst_size) return VBO_DESTINATION_SIZE_ERROR;
        memcpy(dst, src, src_size);
        return src_size;
    }
    const void* cur = src;
    vbo_size_t cur_size = src_size;
    void* scratch = NULL;
    if (o->zstd_compression_level != 0) {
        unsigned long long content = vbo_zstd_content_size(src, src_size);
        if (content >= (unsigned long long)-2) return VBO_ZSTD_ERROR; /* ZSTD_isError on the u64, :236-240 */
        void* zdst = dst;
        size_t zcap = dst_size;
        if (o->integer_size != 0) {
            scratch = malloc(content ? (size_t)content : 1);
            if (!scratch) return VBO_OUT_OF_MEMORY_ERROR;
            zdst = scratch;
            zcap = (vbo_size_t)content; /* truncated to vbz_size_t like :257 */
        } else if (content > dst_size) {
            return VBO_DESTINATION_SIZE_ERROR;
        }
        size_t r = vbo_zstd_decompress(zdst, zcap, src, src_size);
        if (r == (size_t)-1) { free(scratch); return VBO_ZSTD_ERROR; }
        cur = zdst;
        cur_size = (vbo_size_t)r;
    }
    if (o->integer_size == 0) return cur_size;
    if (o->vbz_version > 1) { free(scratch); return VBO_VERSION_ERROR; }
    vbo_size_t res = vbo_streamvbyte_decompress(cur, cur_size, dst, dst_size, (int)o->integer_size,
                                                o->perform_delta_zig_zag, o->vbz_version);
    free(scratch);
    return res;
}

vbo_size_t vbo_compress_sized(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap, const VboOptions* o)
{
    /* vbz.cpp:302-330. NOTE reference quirk kept: an error from vbz_compress has 4 added to it. */
    if (!valid_int_size(o)) return VBO_INTEGER_SIZE_ERROR;
    if (dst_cap < 4) return VBO_DESTINATION_SIZE_ERROR; /* gsl subspan would terminate; report instead */
    memcpy(dst, &src_size, 4);
    vbo_size_t r = vbo_compress(src, src_size, (uint8_t*)dst + 4, dst_cap - 4, o);
    return r + 4;
}

vbo_size_t vbo_decompress_sized(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap, const VboOptions* o)
{
    /* vbz.cpp:332-366 */
    if (!valid_int_size(o)) return VBO_INTEGER_SIZE_ERROR;
    if (src_size < 4) return VBO_INPUT_SIZE_ERROR;
    uint32_t original;
    memcpy(&original, src, 4);
    if (dst_cap < original) return VBO_DESTINATION_SIZE_ERROR;
    return vbo_decompress((const uint8_t*)src + 4, src_size - 4, dst, original, o);
}

vbo_size_t vbo_decompressed_size(const void* src, vbo_size_t src_size, const VboOptions* o)
{
    /* vbz.cpp:368-386 */
    if (!valid_int_size(o)) return VBO_INTEGER_SIZE_ERROR;
    if (src_size < 4) return VBO_INPUT_SIZE_ERROR;
    uint32_t original;
    memcpy(&original, src, 4);
    return original;
}

/* ------------------------------------------------------------------------------------------
 * L3 -- HDF5 filter 32020, reference vbz_plugin/vbz_plugin.cpp:97-229 (POSIX malloc/free branch)
 * ------------------------------------------------------------------------------------------ */
size_t vbo_filter(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes, size_t* buf_size, void** buf)
{
    (void)nbytes;
    if (cd_nelmts < 3) return 0; /* :109-112 */
    VboOptions o;
    o.vbz_version = cd_values[0];
    o.integer_size = cd_values[1];
    o.perform_delta_zig_zag = cd_values[2] != 0;
    o.zstd_compression_level = cd_nelmts > 3 ? cd_values[3] : 1; /* :118-122 */
    void* out = NULL;
    vbo_size_t out_cap = 0, used = 0;
    if (flags & 0x0100u) { /* H5Z_FLAG_REVERSE :136 */
        if (*buf_size > 0xFFFFFFFFull) return 0;
        vbo_size_t expect = vbo_decompressed_size(*buf, (vbo_size_t)*buf_size, &o);
        if (vbo_is_error(expect)) return 0;
        out = malloc(expect ? expect : 1);
        if (!out) return 0;
        used = vbo_decompress_sized(*buf, (vbo_size_t)*buf_size, out, expect, &o);
        if (vbo_is_error(used) || used != expect) { free(out); return 0; }
    } else {
        if (*buf_size > 0xFFFFFFFFull) return 0;
        if (o.integer_size == 0 || *buf_size % o.integer_size != 0) return 0; /* :194-199 (int_size 0 would be a div-by-zero there) */
        out_cap = vbo_max_compressed_size((vbo_size_t)*buf_size, &o);
        if (vbo_is_error(out_cap)) return 0;
        out = malloc(out_cap);
        if (!out) return 0;
        used = vbo_compress_sized(*buf, (vbo_size_t)*buf_size, out, out_cap, &o);
        if (vbo_is_error(used)) { free(out); return 0; }
    }
    free(*buf);
    *buf = out;
    *buf_size = out_cap; /* reference leaves 0 here on the reverse branch (:106,227) */
    return used;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic signal of SURVEY.md section 8(d).  All arithmetic is uint64 wrap-around.
 * ------------------------------------------------------------------------------------------ */
uint64_t vbo_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static uint64_t synth_key(uint64_t seed, uint64_t r) { return vbo_mix64(seed * 0x100000001B3ull + r); }

uint32_t vbo_synth_read_length(uint64_t seed, uint64_t r)
{
    return 90000u + (uint32_t)(vbo_mix64(synth_key(seed, r) ^ 0xC2B2AE3D27D4EB4Full) % 20001u);
}

void vbo_synth_signal(uint64_t seed, uint64_t r, int16_t* out, size_t n)
{
    uint64_t key = synth_key(seed, r);
    for (size_t i = 0; i < n; ++i) {
        uint64_t hs = vbo_mix64(key ^ ((uint64_t)(i / 32) * 0xD6E8FEB86659FD93ull));
        int32_t level = 200 + (int32_t)(hs % 321u);
        uint64_t hn = vbo_mix64(key ^ ((uint64_t)i * 0xA24BAED4963EE407ull) ^ 0x5555555555555555ull);
        int32_t noise = -60;
        for (int k = 0; k < 8; ++k) noise += (int32_t)((hn >> (4 * k)) & 15u);
        int32_t x = level + noise;
        if (x < -4096) x = -4096;
        if (x > 4095) x = 4095;
        out[i] = (int16_t)x;
    }
}

void vbo_synth_u32(uint64_t seed, uint64_t r, uint32_t* out, size_t n)
{
    uint64_t key = synth_key(seed, r);
    for (size_t i = 0; i < n; ++i) {
        uint64_t h = vbo_mix64(key ^ ((uint64_t)i * 0xA24BAED4963EE407ull));
        uint32_t sel = (uint32_t)(h & 127u);
        uint32_t s = sel < 90 ? 24 : (sel < 115 ? 16 : (sel < 125 ? 8 : 0));
        out[i] = (uint32_t)(h >> 32) >> s;
    }
}
