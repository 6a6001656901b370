/*
 * vbz.h -- drop-in C ABI of the VBZ codec, implemented on MI355X (gfx950) by libvbz_hip.so.
 *
 * This header mirrors the reference's public interface symbol for symbol so an existing caller
 * (libhdf5 filter pipeline, pyvbz's CFFI binding, the reference's own tests) can be re-linked
 * against libvbz_hip.so unchanged:
 *
 *   reference vbz/vbz.h:11-53    vbz_size_t, the seven error codes, struct CompressionOptions
 *   reference vbz/vbz.h:56-141   the eight extern "C" entry points declared below
 *
 * Same names, same argument meaning, same struct layout (sizeof(CompressionOptions) == 16:
 * bool @0, unsigned @4, @8, @12), same in-band uint32 error sentinels.  Host pointers in, host
 * pointers out: the library stages through its own device buffers, runs the HIP kernels and copies
 * the result back.  For throughput use the batched, device-resident extension in vbz_gpu.h.
 *
 * Deliberate divergences from the reference (all documented in DESIGN.md):
 *   - there is NO CPU fallback: when no gfx950 device / kernel image is usable (or a device operation fails) the
 *     entry points print the cause on stderr and return the reference's own VBZ_OUT_OF_MEMORY_ERROR, so that a
 *     binary compiled against the reference header and only re-linked (`ret >= VBZ_FIRST_ERROR`, -7) sees an error.
 *     VBZ_DEVICE_ERROR (-8) exists for the per-read results of the batched API (vbz_gpu.h); vbz_is_error() knows it.
 *   - vbz_compress_sized returns the error code of vbz_compress unchanged instead of adding 4 to it
 *     (reference vbz/vbz.cpp:321-329 turns VBZ_INPUT_SIZE_ERROR into 2).
 *   - the zstd stage is this library's own encoder: frames are standard zstd (RFC 8878) and decode
 *     with any libzstd / the reference's vbz_decompress, but are not byte-identical to libzstd's.
 *     A compressed buffer may end in zstd skippable frames with hints for this library's parallel decoder, which
 *     libzstd skips (RFC 8878 3.1.2): checkpoints of the sequences section (magic 0x184D2A5B, <= 272 bytes) and, behind
 *     reads of half a megabyte or more (and behind every read of a small batch), an index of the frame's spans (magic
 *     0x184D2A5C, 8 bytes per span of 4 - 32 KB of content, 256 KB in very large frames: not bounded by a constant).  vbz_gpu_set_trailers / VBZ_HIP_TRAILERS=0
 *     writes plain single frames.
 *   - the decoder reads RFC 8878 frames only.  zstd's legacy frame formats (v0.2 ... v0.7, magic 0xFD2FB522 ... 27), which
 *     a libzstd built with ZSTD_LEGACY_SUPPORT also decodes, are VBZ_ZSTD_ERROR here (no vbz writer ever produced them;
 *     32 files of the reference's fuzz corpus start with the v0.7 magic).  A frame may carry a Dictionary_ID field of 0
 *     ("no dictionary"); any other dictionary is VBZ_ZSTD_ERROR, as with libzstd when it has not got that dictionary.
 */
#ifndef VBZ_H_MI355X
#define VBZ_H_MI355X

#include <stdbool.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#ifndef VBZ_EXPORT
#define VBZ_EXPORT __attribute__((visibility("default")))
#endif

#define VBZ_DEFAULT_VERSION 0

typedef uint32_t vbz_size_t;

/* reference vbz/vbz.h:15-22 */
#define VBZ_ZSTD_ERROR ((vbz_size_t)-1)
#define VBZ_INPUT_SIZE_ERROR ((vbz_size_t)-2)
#define VBZ_INTEGER_SIZE_ERROR ((vbz_size_t)-3)
#define VBZ_DESTINATION_SIZE_ERROR ((vbz_size_t)-4)
#define VBZ_STREAMVBYTE_STREAM_ERROR ((vbz_size_t)-5)
#define VBZ_VERSION_ERROR ((vbz_size_t)-6)
#define VBZ_OUT_OF_MEMORY_ERROR ((vbz_size_t)-7)
#define VBZ_FIRST_ERROR VBZ_OUT_OF_MEMORY_ERROR
/* extension, vbz_gpu.h results only (never returned by the functions below): the HIP device is unusable */
#define VBZ_DEVICE_ERROR ((vbz_size_t)-8)

/* Deprecated aliases, reference vbz/vbz.h:24-27 */
#define VBZ_STREAMVBYTE_INPUT_SIZE_ERROR VBZ_INPUT_SIZE_ERROR
#define VBZ_STREAMVBYTE_INTEGER_SIZE_ERROR VBZ_INTEGER_SIZE_ERROR
#define VBZ_STREAMVBYTE_DESTINATION_SIZE_ERROR VBZ_DESTINATION_SIZE_ERROR

/* reference vbz/vbz.h:29-53 */
struct CompressionOptions
{
    /* delta + zig-zag before the variable-byte stage */
    bool perform_delta_zig_zag;
    /* 0 (no streamvbyte stage), 1, 2 or 4 */
    unsigned int integer_size;
    /* 0 = no zstd stage; any other value selects this library's zstd-format entropy stage.  The reference hands the level to
     * libzstd (vbz/vbz.cpp:194-207); this encoder does the same work at every level: Huffman-coded literals, run sequences
     * for the control bytes (what libzstd gets out of nanopore signal), and -- for a read that repeats a template, like the
     * reads of the reference's own perf generator -- ONE long repeat distance in the data bytes, coded as matches at that
     * distance (what libzstd's match finder makes of such reads at any level), whatever the read's length (a read of more
     * than 524 288 samples keeps its control bytes Huffman coded without the runs: 44 x where libzstd gets 148 x on a
     * 1 M-sample cycled read).  Levels above 1 write the bytes of level 1: this is not libzstd's 1 ... 22 scale, higher levels do
     * not search harder -- on nanopore signal libzstd's own level 5 (the level of the reference's HDF5 test,
     * vbz_plugin/test/vbz_hdf_plugin_test.cpp:34) is 0.1 - 0.5 % smaller than its level 1, and this encoder stays within 1 % of
     * libzstd's level 1 at any level (tests/test_gpu_parity.py::test_zstd_levels_above_one_write_the_level_one_frames).
     * Decoding does not depend on the level. */
    unsigned int zstd_compression_level;
    /* 0 or 1 (identical for integer_size 2 and 4, reference vbz/v1/vbz_streamvbyte.cpp:46-61) */
    unsigned int vbz_version;
};
#if !defined(__cplusplus)
typedef struct CompressionOptions CompressionOptions;
#endif

/* replaces reference vbz/vbz.cpp:61-64 */
VBZ_EXPORT bool vbz_is_error(vbz_size_t result_value);
/* replaces reference vbz/vbz.cpp:66-77 */
VBZ_EXPORT char const* vbz_error_string(vbz_size_t error_value);
/* replaces reference vbz/vbz.cpp:79-114 : svb bound (n+3)/4+4n -> ZSTD_compressBound -> +4 */
VBZ_EXPORT vbz_size_t vbz_max_compressed_size(vbz_size_t source_size, struct CompressionOptions const* options);
/* replaces reference vbz/vbz.cpp:116-208 */
VBZ_EXPORT vbz_size_t vbz_compress(void const* source, vbz_size_t source_size, void* destination,
                                   vbz_size_t destination_capacity, struct CompressionOptions const* options);
/* replaces reference vbz/vbz.cpp:210-300 ; destination_size must be the exact original byte count */
VBZ_EXPORT vbz_size_t vbz_decompress(void const* source, vbz_size_t source_size, void* destination,
                                     vbz_size_t destination_size, struct CompressionOptions const* options);
/* replaces reference vbz/vbz.cpp:302-330 : [u32 LE original_size][vbz_compress payload] */
VBZ_EXPORT vbz_size_t vbz_compress_sized(void const* source, vbz_size_t source_size, void* destination,
                                         vbz_size_t destination_capacity, struct CompressionOptions const* options);
/* replaces reference vbz/vbz.cpp:332-366 */
VBZ_EXPORT vbz_size_t vbz_decompress_sized(void const* source, vbz_size_t source_size, void* destination,
                                           vbz_size_t destination_capacity, struct CompressionOptions const* options);
/* replaces reference vbz/vbz.cpp:368-386 */
VBZ_EXPORT vbz_size_t vbz_decompressed_size(void const* source, vbz_size_t source_size,
                                            struct CompressionOptions const* options);

#if defined(__cplusplus)
}
#endif
#endif
