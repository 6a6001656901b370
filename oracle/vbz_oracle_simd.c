/*
 * vbz_oracle_simd.c -- CPU ORACLE (test infrastructure): an SSSE3 form of the int16 zig-zag streamvbyte stage for the CPU
 * BASELINE leg of bench.py, so that the baseline is the reference's class of CPU path (its hot path is an SSSE3 worker:
 * reference vbz/v0/vbz_streamvbyte_impl_sse3.h:403-659) and not a scalar loop.
 *
 * Own code, not the reference's: eight samples per step, shuffle tables indexed by the step's eight "two bytes?" bits
 * (built at first use), the delta chain as an in-register prefix sum.  Its OUTPUT is pinned by the scalar restatement
 * (vbz_oracle.c: i16zz_compress / i16zz_decompress, which the reference's known answers pin): tests/test_oracle.py
 * compares the two byte for byte.  Streams with three- or four-byte codes (never written by a vbz encoder) and the tail
 * of a stream are handed to the scalar functions.  Only vbo_bench_roundtrip() switches this path on (vbo_use_simd_svb).
 */
#include <stdatomic.h>
#include <string.h>

#include "vbz_oracle.h"

#if defined(__x86_64__) || defined(__i386__)
#include <tmmintrin.h>

static uint8_t enc_shuf[256][16];   /* bits b7..b0 (value k takes two bytes) -> pshufb mask that packs the bytes of 8 x u16 */
static uint8_t dec_shuf[256][16];   /* the inverse: packed bytes -> 8 x u16 (high byte zero for one-byte values) */
static uint16_t key_of[256];        /* b -> the two control bytes (codes 0 / 1 at two bits each) */
static uint8_t len_of[256];         /* b -> data bytes of the group: 8 + popcount(b) */
static atomic_int tables_ready;

static void build_tables(void)
{
    for (int b = 0; b < 256; ++b) {
        int o = 0;
        uint16_t key = 0;
        for (int k = 0; k < 8; ++k) {
            const int two = (b >> k) & 1;
            enc_shuf[b][o] = (uint8_t)(2 * k);
            dec_shuf[b][2 * k] = (uint8_t)o;
            ++o;
            if (two) {
                enc_shuf[b][o] = (uint8_t)(2 * k + 1);
                dec_shuf[b][2 * k + 1] = (uint8_t)o;
                ++o;
            } else {
                dec_shuf[b][2 * k + 1] = 0x80;
            }
            key |= (uint16_t)(two << (2 * k));
        }
        for (int i = o; i < 16; ++i) enc_shuf[b][i] = 0x80;
        key_of[b] = key;
        len_of[b] = (uint8_t)o;
    }
    atomic_store(&tables_ready, 1);
}

int vbo_simd_available(void) { return __builtin_cpu_supports("ssse3") ? 1 : 0; }

/* dst must hold the worst case of the int16 stream: (n + 3) / 4 + 2 n bytes, + 16 bytes of slack for the vector stores */
vbo_size_t vbo_i16zz_compress_simd(const uint8_t* src, vbo_size_t src_size, uint8_t* dst)
{
    if (!atomic_load(&tables_ready)) build_tables();
    const uint32_t n = src_size / 2;
    if (n == 0) return 0;
    const uint32_t key_len = (n + 3) / 4;
    uint8_t* keys = dst;
    uint8_t* data = dst + key_len;
    const __m128i lim = _mm_set1_epi16(0x00FF);
    __m128i prev = _mm_setzero_si128();   /* the previous group's samples */
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m128i x = _mm_loadu_si128((const __m128i*)(src + 2 * (size_t)i));
        const __m128i before = _mm_alignr_epi8(x, prev, 14);                      /* x[i-1] for every lane */
        const __m128i d = _mm_sub_epi16(x, before);                                /* wraps in 16 bits */
        const __m128i zz = _mm_xor_si128(_mm_slli_epi16(d, 1), _mm_srai_epi16(d, 15));
        /* unsigned zz > 255 <=> high byte != 0 */
        const __m128i wide = _mm_xor_si128(_mm_cmpeq_epi16(_mm_andnot_si128(lim, zz), _mm_setzero_si128()), _mm_set1_epi16(-1));
        const unsigned b = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(wide, _mm_setzero_si128())) & 0xFFu;
        _mm_storeu_si128((__m128i*)data, _mm_shuffle_epi8(zz, _mm_loadu_si128((const __m128i*)enc_shuf[b])));
        data += len_of[b];
        const uint16_t key = key_of[b];
        memcpy(keys, &key, 2);
        keys += 2;
        prev = x;
    }
    /* the last n & 7 samples: scalar (sse3.h:449-463 does the same) */
    int16_t p = 0;
    if (i) memcpy(&p, src + 2 * (size_t)(i - 1), 2);
    uint32_t key = 0;
    for (uint32_t k = i; k < n; ++k) {
        int16_t x;
        memcpy(&x, src + 2 * (size_t)k, 2);
        const int16_t d = (int16_t)(uint16_t)((uint16_t)x - (uint16_t)p);
        const uint16_t zz = (uint16_t)(((uint16_t)d << 1) ^ (uint16_t)(d >> 15));
        p = x;
        const uint32_t code = zz > 0xFF;
        key |= code << (2 * (k & 3));
        *data++ = (uint8_t)zz;
        if (code) *data++ = (uint8_t)(zz >> 8);
        if ((k & 3) == 3) { *keys++ = (uint8_t)key; key = 0; }
    }
    if (n & 3) *keys = (uint8_t)key;
    return (vbo_size_t)(data - dst);
}

/* Returns the decoded byte count, an error like the scalar function, or VBO_SIMD_DECLINED for a stream this path does
 * not take (a code above 1): the caller then uses the scalar function. */
vbo_size_t vbo_i16zz_decompress_simd(const uint8_t* src, vbo_size_t src_size, uint8_t* dst, vbo_size_t dst_size)
{
    if (!atomic_load(&tables_ready)) build_tables();
    const uint32_t count = dst_size / 2;
    if (count == 0) return 0;
    const uint32_t key_bytes = (count + 3) / 4;
    if (src_size < key_bytes) return VBO_INPUT_SIZE_ERROR;
    const uint8_t* keys = src;
    const uint8_t* data = src + key_bytes;
    size_t remaining = src_size - key_bytes;
    __m128i carry = _mm_setzero_si128();   /* the last sample, in every lane */
    uint32_t i = 0;
    /* like the reference's body: whole groups while 32 data bytes are left (a group reads 16) */
    for (; i + 8 <= count && remaining >= 32; i += 8) {
        uint16_t key;
        memcpy(&key, keys + (i >> 2), 2);
        if (key & 0xAAAAu) return VBO_SIMD_DECLINED;
        unsigned b = key & 0x5555u;                 /* the eight code bits, squeezed together */
        b = (b | (b >> 1)) & 0x3333u;
        b = (b | (b >> 2)) & 0x0F0Fu;
        b = (b | (b >> 4)) & 0x00FFu;
        const __m128i u = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)data), _mm_loadu_si128((const __m128i*)dec_shuf[b]));
        data += len_of[b];
        remaining -= len_of[b];
        __m128i d = _mm_xor_si128(_mm_srli_epi16(u, 1), _mm_sub_epi16(_mm_setzero_si128(), _mm_and_si128(u, _mm_set1_epi16(1))));
        d = _mm_add_epi16(d, _mm_slli_si128(d, 2));   /* prefix sum over the eight lanes */
        d = _mm_add_epi16(d, _mm_slli_si128(d, 4));
        d = _mm_add_epi16(d, _mm_slli_si128(d, 8));
        d = _mm_add_epi16(d, carry);
        _mm_storeu_si128((__m128i*)(dst + 2 * (size_t)i), d);
        carry = _mm_shuffle_epi8(d, _mm_set1_epi16(0x0F0E));
    }
    uint16_t prev = (uint16_t)_mm_extract_epi16(carry, 0);
    for (; i < count; ++i) {   /* the scalar tail, with its bounds checks (sse3.h:542-572) */
        const uint32_t code = (keys[i >> 2] >> (2 * (i & 3))) & 3u;
        if (code > 1) return VBO_SIMD_DECLINED;
        if (remaining < code + 1) return VBO_STREAM_ERROR;
        uint32_t v = data[0];
        if (code) v |= (uint32_t)data[1] << 8;
        data += code + 1;
        remaining -= code + 1;
        const uint32_t dz = (v >> 1) ^ (0u - (v & 1u));
        prev = (uint16_t)(dz + prev);
        memcpy(dst + 2 * (size_t)i, &prev, 2);
    }
    if (remaining != 0) return VBO_STREAM_ERROR;
    return count * 2;
}

#else

int vbo_simd_available(void) { return 0; }
vbo_size_t vbo_i16zz_compress_simd(const uint8_t* src, vbo_size_t src_size, uint8_t* dst) { (void)src; (void)src_size; (void)dst; return VBO_SIMD_DECLINED; }
vbo_size_t vbo_i16zz_decompress_simd(const uint8_t* src, vbo_size_t src_size, uint8_t* dst, vbo_size_t dst_size)
{
    (void)src; (void)src_size; (void)dst; (void)dst_size;
    return VBO_SIMD_DECLINED;
}

#endif
