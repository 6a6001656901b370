/*
 * vbz_oracle.h -- CPU ORACLE for the VBZ int16 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (nanoporetech/vbz_compression)
 * used as the checker for the HIP product path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product library (libvbz_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - the reference itself is UNBUILDABLE in this image without stand-ins (its
 *     third_party/streamvbyte submodule is empty and vbz/vbz_export.h is cmake-generated),
 *     so there is no oracle/_ref;
 *   - the restatement is pinned against every known-answer vector the reference's own
 *     tests hold for this path (tests/golden/kat.json, each entry cites file:line) and
 *     against the decode pins in the three shipped fast5 files (tests/golden/fast5_*.bin);
 *   - the zstd stage of the oracle is the pinned third-party dependency itself
 *     (facebook/zstd, conan pin zstd/1.4.8, reference CMakeLists.txt:92-93), loaded at run
 *     time with dlopen("libzstd.so.1"); vbo_zstd_version() reports what was loaded.
 *   - streamvbyte (lemire/streamvbyte, un-vendored submodule, unpinned) is restated from
 *     its published format; uint32 values needing 3-4 bytes have no KAT in the reference:
 *     "parity unpinned" for those beyond round trips.
 */
#ifndef VBZ_ORACLE_H
#define VBZ_ORACLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t vbo_size_t;

/* reference vbz/vbz.h:15-22 */
#define VBO_ZSTD_ERROR ((vbo_size_t)-1)
#define VBO_INPUT_SIZE_ERROR ((vbo_size_t)-2)
#define VBO_INTEGER_SIZE_ERROR ((vbo_size_t)-3)
#define VBO_DESTINATION_SIZE_ERROR ((vbo_size_t)-4)
#define VBO_STREAM_ERROR ((vbo_size_t)-5)
#define VBO_VERSION_ERROR ((vbo_size_t)-6)
#define VBO_OUT_OF_MEMORY_ERROR ((vbo_size_t)-7)
#define VBO_FIRST_ERROR VBO_OUT_OF_MEMORY_ERROR

/* reference vbz/vbz.h:29-53 (sizeof == 16: bool@0, u32@4,8,12) */
typedef struct VboOptions {
    bool perform_delta_zig_zag;
    unsigned int integer_size;
    unsigned int zstd_compression_level;
    unsigned int vbz_version;
} VboOptions;

/* ---- L1: integer codec (reference vbz/v0/vbz_streamvbyte.cpp, vbz/v1/vbz_streamvbyte.cpp) */
/* vbz_oracle_simd.c: an SSSE3 form of the int16 zig-zag stage, byte-identical to the scalar one, for the CPU baseline leg
 * only.  vbo_use_simd_svb(1) makes vbo_streamvbyte_compress / _decompress (hence vbo_compress / vbo_decompress) take it
 * for integer_size 2 with zig-zag; the default is the scalar restatement. */
#define VBO_SIMD_DECLINED ((vbo_size_t)-100)
int vbo_simd_available(void);
void vbo_use_simd_svb(int on);
vbo_size_t vbo_i16zz_compress_simd(const uint8_t* src, vbo_size_t src_size, uint8_t* dst);
vbo_size_t vbo_i16zz_decompress_simd(const uint8_t* src, vbo_size_t src_size, uint8_t* dst, vbo_size_t dst_size);
vbo_size_t vbo_max_streamvbyte_size(size_t integer_size, vbo_size_t source_size);
vbo_size_t vbo_streamvbyte_compress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap,
                                    int integer_size, bool zigzag, unsigned version);
vbo_size_t vbo_streamvbyte_decompress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_size,
                                      int integer_size, bool zigzag, unsigned version);

/* ---- L2: C API (reference vbz/vbz.cpp) */
bool vbo_is_error(vbo_size_t v);
const char* vbo_error_string(vbo_size_t v);
vbo_size_t vbo_max_compressed_size(vbo_size_t source_size, const VboOptions* o);
vbo_size_t vbo_compress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap, const VboOptions* o);
vbo_size_t vbo_decompress(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_size, const VboOptions* o);
vbo_size_t vbo_compress_sized(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap, const VboOptions* o);
vbo_size_t vbo_decompress_sized(const void* src, vbo_size_t src_size, void* dst, vbo_size_t dst_cap, const VboOptions* o);
vbo_size_t vbo_decompressed_size(const void* src, vbo_size_t src_size, const VboOptions* o);

/* ---- L3: HDF5 filter 32020 calling convention (reference vbz_plugin/vbz_plugin.cpp:97-229) */
size_t vbo_filter(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes,
                  size_t* buf_size, void** buf);

/* ---- zstd dependency (dlopen'd) */
const char* vbo_zstd_version(void);          /* NULL if libzstd.so.1 could not be loaded */
size_t vbo_zstd_compress(void* dst, size_t cap, const void* src, size_t n, int level);     /* (size_t)-1 on error */
size_t vbo_zstd_decompress(void* dst, size_t cap, const void* src, size_t n);              /* (size_t)-1 on error */
size_t vbo_zstd_bound(size_t n);
unsigned long long vbo_zstd_content_size(const void* src, size_t n);                       /* >= (u64)-2 on error/unknown */

/* ---- synthetic signal generator of SURVEY.md section 8(d) (integer-only, counter based) */
uint64_t vbo_mix64(uint64_t x);
uint32_t vbo_synth_read_length(uint64_t seed, uint64_t read_index);     /* config 2/5 length rule */
void vbo_synth_signal(uint64_t seed, uint64_t read_index, int16_t* out, size_t n);
void vbo_synth_u32(uint64_t seed, uint64_t read_index, uint32_t* out, size_t n);  /* config 4 values */

/* ---- restated zstd frame decoder (oracle/zstd_restate.c), the CPU mirror of the HIP decoder.
 * Returns decoded size, or (size_t)-1 on any format error. Writes at most cap bytes. */
size_t vbo_zstd_restate_decompress(void* dst, size_t cap, const void* src, size_t n);

/* ---- replay of the reference's fuzz target (oracle/vbz_oracle_fuzz.c; reference vbz/fuzzing/vbz_fuzz.cpp:138-161) */
uint32_t vbo_fuzz_max_destination(uint32_t size, const VboOptions* o);
int vbo_fuzz_decompress_sweep(const void* data, uint32_t size, const VboOptions* o, uint32_t max_destination, uint32_t* results);

#ifdef __cplusplus
}
#endif
#endif
