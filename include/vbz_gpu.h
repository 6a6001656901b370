/*
 * vbz_gpu.h -- batched, device-resident extension of the VBZ C ABI (MI355X / gfx950).
 *
 * The reference API (vbz.h) handles one host buffer per call, which cannot feed a GPU
 * (SURVEY.md section 8b "needed extension").  This header adds the entry points a caller that
 * already holds many reads in HBM binds instead: one call encodes or decodes a whole batch of
 * independent reads, each with exactly the semantics of the corresponding single-buffer call
 *
 *   vbz_gpu_compress_batch   == n x vbz_compress[_sized]    (reference vbz/vbz.cpp:116-208,302-330)
 *   vbz_gpu_decompress_batch == n x vbz_decompress[_sized]  (reference vbz/vbz.cpp:210-300,332-366)
 *
 * Plain C: pointers and sizes only, no torch / HIP types in the signatures (the stream is passed
 * as an opaque void* that is a hipStream_t).  All pointers in vbz_gpu_batch are DEVICE pointers.
 * Calls are asynchronous on the context's stream; results (per-read sizes or vbz error codes) land
 * in batch->result in stream order.
 *
 * How a batch is laid onto the device is the library's business and never changes the DECODED data.  The compressed bytes
 * (and so result[i] of a compress call) may depend on it: reads on the large-read path are coded as spans (unless they
 * repeat at one distance, below), a batch too small to fill the device is coded as spans too, and the matcher is used only
 * where dst_cap[i] leaves room for its workspace above the worst-case frame.  The library may write anywhere inside a read's
 * destination slot [dst_off[i], dst_off[i] + dst_cap[i]) (workspace, staging), not only the result[i] bytes it reports.
 *   - the shape rule: a batch whose average read is half a megabyte or more (a 10 M-element buffer), or which is too
 *     small to fill the device with one wavefront per read (up to 96 MB of reads of 64 KB and more: one HDF5 chunk per
 *     call), runs on the large-read path, many workgroups per read;
 *   - per-read routing: in any other batch the reads of 512 KB and more (at most 16 of them, 64 MB in all;
 *     the first in batch order; none if the batch holds more than 1024 such reads) are coded on that path beside the rest of the batch, on a second stream of the
 *     context that is forked from and joined to the context's stream inside the call -- to the caller the
 *     call is still one unit of work in stream order;
 *   - a read whose bytes repeat at one distance (a cycled template) gets that distance coded as zstd matches
 *     at every zstd_compression_level (the reference passes its level to libzstd, whose matcher is on at all
 *     of them), whatever its length: a long read that has such a distance is coded by one wavefront with the matcher
 *     instead of as spans (15-60 x smaller, at one wavefront's speed);
 *   - decompress: frames of this library's shape are decoded by a batched decoder, any other conforming zstd frame by the
 *     general one in the same call; in calls of 2 560 reads and more the sequence chains of frames libzstd wrote (the
 *     reference's files) are walked ahead of the general decoder, one lane per frame, on a second stream of the context
 *     that is forked from and joined to the context's stream inside the call.  Bytes and verdicts never depend on it
 *     (vbz_gpu_decode_paths tells which frames went which way).
 * Descriptor tables are untrusted like the data: before any other kernel runs, one thread per read checks
 * src_off + src_size <= src_bytes and dst_off + dst_cap <= dst_bytes (64-bit arithmetic); a read that fails gets
 * VBZ_INPUT_SIZE_ERROR or VBZ_DESTINATION_SIZE_ERROR and none of its addresses is ever formed
 * (vbz_gpu_compress_batch / vbz_gpu_decompress_batch; the stage-level entry points below trust their tables).
 */
#ifndef VBZ_GPU_H_MI355X
#define VBZ_GPU_H_MI355X

#include <stddef.h>
#include <stdint.h>

#include "vbz.h"

#if defined(__cplusplus)
extern "C" {
#endif

typedef struct vbz_gpu_ctx vbz_gpu_ctx;

typedef struct vbz_gpu_batch
{
    uint32_t n_reads;
    uint32_t reserved;
    /* inputs: read i occupies src[src_off[i] .. src_off[i]+src_size[i]).  Offsets must not overlap,
     * should be 16-byte aligned (a slower path handles other alignments), and the arena must be
     * readable for 16 bytes past the last input (the same slack streamvbyte asks for).
     * Decompression additionally reads whole ALIGNED 128-byte lines: device memory must be readable
     * from the 128-byte line that holds the first input byte to the end of the line that holds the
     * last one (+ 16).  Any arena that is an allocation of its own (hipMalloc: 256-byte aligned,
     * page granular) satisfies this; an arena carved out of a larger buffer does as long as the
     * enclosing buffer covers those lines. */
    const void* src;
    const uint64_t* src_off;
    const uint32_t* src_size;
    uint64_t src_bytes; /* extent of the src arena covering every read (sizes the scratch) */
    /* outputs: read i may use dst[dst_off[i] .. dst_off[i]+dst_cap[i]).
     *   compress:   dst_cap[i] >= vbz_max_compressed_size(src_size[i]) (as for vbz_compress)
     *   decompress: dst_cap[i] == the exact original byte count      (as for vbz_decompress);
     *               for the sized variant it is the capacity and the size comes from the header */
    void* dst;
    const uint64_t* dst_off;
    const uint32_t* dst_cap;
    uint64_t dst_bytes; /* extent of the dst arena covering every slot */
    /* per read: number of bytes produced, or a vbz error code (vbz_is_error) */
    uint32_t* result;
} vbz_gpu_batch;

/* Create a context on HIP device `device`.  `stream` is a hipStream_t to launch on (NULL: the
 * context creates its own non-blocking stream).  Returns NULL (message on stderr) if no gfx950
 * device or kernel image is usable. */
VBZ_EXPORT vbz_gpu_ctx* vbz_gpu_create(int device, void* stream);
VBZ_EXPORT void vbz_gpu_destroy(vbz_gpu_ctx* ctx);
VBZ_EXPORT void* vbz_gpu_stream(vbz_gpu_ctx* ctx);
VBZ_EXPORT const char* vbz_gpu_last_error(vbz_gpu_ctx* ctx);
/* Decoder hints behind the zstd frame.  By default a compressed buffer may end in zstd SKIPPABLE frames (RFC 8878 3.1.2;
 * libzstd, hence the reference's vbz_decompress, ignores them): checkpoints of the sequences section (magic 0x184D2A5B,
 * <= 272 bytes) and, for reads of half a megabyte or more (and for every read of a small batch), an index of the frame's spans (magic
 * 0x184D2A5C, 8 bytes per span of 4 - 32 KB of content, 256 KB in very large frames; bit 31 of its span count says that the data bytes share one Huffman table: spans that
 * begin with a treeless block).  This library's decoder uses them to decode one frame on many lanes / wavefronts and verifies
 * them; without them it decodes the same frames, more slowly.  enable = 0 writes plain single zstd frames (for consumers
 * that insist on consumed == source size after ONE frame); the compression itself (run sequences included) is unchanged.
 * The single-buffer API of vbz.h follows the environment variable VBZ_HIP_TRAILERS (0 / 1, default 1). */
VBZ_EXPORT void vbz_gpu_set_trailers(vbz_gpu_ctx* ctx, int enable);
/* Canonical encoding.  The reference's output for a buffer is a function of (input, options, libzstd version) alone
 * (vbz/vbz.cpp:116-208).  Every frame this library writes is standard zstd that the reference decodes, but by default WHICH kernels code
 * a read -- hence its bytes -- follows from the shape of the call it arrives in (few large reads, a handful of reads, thousands): the
 * same read may come out differently from the HDF5 filter, the bulk re-packer and a large batch.  enable = 1: a read's bytes depend on
 * the read, the options, the trailer setting and the library version only -- reads of 512 KiB of raw data and more are coded as spans
 * with a table each, all others by one wavefront, whatever else is in the call.  Destination slots must have the reference's capacity
 * (vbz_max_compressed_size), as the reference requires anyway.  Costs one stream synchronisation per compress call (the number of large
 * reads comes back to the host), nothing at thousands of ordinary reads per call, and the small-call latency of the default (a call
 * with one 100 k-sample read: ~0.4 ms instead of ~0.14).  Decoding is unaffected.  The single-buffer API of vbz.h, the HDF5 plugin and
 * the re-packer follow the environment variable VBZ_HIP_CANONICAL (0 / 1, default 0). */
VBZ_EXPORT void vbz_gpu_set_canonical(vbz_gpu_ctx* ctx, int enable);
/* wait for everything queued on the context's stream; returns 0 or a negative HIP error */
VBZ_EXPORT int vbz_gpu_synchronize(vbz_gpu_ctx* ctx);

/* Both return 0 when the batch was queued, negative on a launch/allocation failure (see
 * vbz_gpu_last_error; -2: options this library does not know, or declared arena extents beyond 2^46 bytes, refused before
 * anything is sized by them).  Per-read failures are reported in batch->result, not here. */
VBZ_EXPORT int vbz_gpu_compress_batch(vbz_gpu_ctx* ctx, const vbz_gpu_batch* batch,
                                      const struct CompressionOptions* options, int sized);
VBZ_EXPORT int vbz_gpu_decompress_batch(vbz_gpu_ctx* ctx, const vbz_gpu_batch* batch,
                                        const struct CompressionOptions* options, int sized);

/* Stage-level entry points (the two halves of the path, used by tests and stage benchmarks).
 *   svb:  reference vbz_delta_zig_zag_streamvbyte_{compress,decompress}_v{0,1}
 *         (vbz/v0/vbz_streamvbyte.cpp:20-108, vbz/v1/vbz_streamvbyte.cpp:22-113)
 *   zstd: the reference's ZSTD_compress / ZSTD_decompress call sites (vbz/vbz.cpp:194-207,236-273)
 * For zstd decompress, dst_cap[i] is the capacity and result[i] the frame content size. */
VBZ_EXPORT int vbz_gpu_svb_compress_batch(vbz_gpu_ctx* ctx, const vbz_gpu_batch* batch, int integer_size,
                                          int zigzag, int version);
VBZ_EXPORT int vbz_gpu_svb_decompress_batch(vbz_gpu_ctx* ctx, const vbz_gpu_batch* batch, int integer_size,
                                            int zigzag, int version);
/* key_bytes (device, nullable): length of the control-byte section of each svb stream, which the
 * encoder codes with its own Huffman table; NULL codes the whole stream as one region. */
VBZ_EXPORT int vbz_gpu_zstd_compress_batch(vbz_gpu_ctx* ctx, const vbz_gpu_batch* batch, const uint32_t* key_bytes);
VBZ_EXPORT int vbz_gpu_zstd_decompress_batch(vbz_gpu_ctx* ctx, const vbz_gpu_batch* batch);

/* Synthetic workload of SURVEY.md section 8(d), generated on the device (no host data needed).
 *   lengths:  out_len[i] = samples of read first_read+i (90 000 + mix(..) % 20 001), i < n_reads
 *   signal:   int16 samples of read first_read+i written at dst + off[i] (bytes), len[i] samples
 *   u32:      config-4 values, len[i] elements */
VBZ_EXPORT int vbz_gpu_synth_lengths(vbz_gpu_ctx* ctx, uint64_t seed, uint64_t first_read, uint32_t n_reads,
                                     uint32_t* out_len);
VBZ_EXPORT int vbz_gpu_synth_signal(vbz_gpu_ctx* ctx, uint64_t seed, uint64_t first_read, uint32_t n_reads,
                                    void* dst, const uint64_t* off, const uint32_t* len);
VBZ_EXPORT int vbz_gpu_synth_u32(vbz_gpu_ctx* ctx, uint64_t seed, uint64_t first_read, uint32_t n_reads,
                                 void* dst, const uint64_t* off, const uint32_t* len);

/* Per-kernel timing with HIP events recorded on the context's stream around every launch.
 * enable=1 starts collecting; vbz_gpu_profile_read synchronizes, copies up to `cap` entries
 * (kernel name, launches, total milliseconds) and returns the number of distinct kernels. */
VBZ_EXPORT void vbz_gpu_profile_enable(vbz_gpu_ctx* ctx, int enable);
VBZ_EXPORT int vbz_gpu_profile_read(vbz_gpu_ctx* ctx, const char** names, uint32_t* launches, double* total_ms, int cap);
VBZ_EXPORT void vbz_gpu_profile_reset(vbz_gpu_ctx* ctx);

/* How the frames of the context's last decompress launch group were decoded (a diagnostic; synchronizes): *batched = frames decoded by
 * the batched decoder for frames this library wrote, *walked = frames whose sequence chains were walked one lane per frame ahead of the
 * general decoder (frames the reference wrote with libzstd).  Returns the number of frames of that group, 0 when it did not run on these
 * paths (the large-read path, VBZ_HIP_FAST_DECODE=0), < 0 on a device error.  The bytes and verdicts never depend on the path. */
VBZ_EXPORT int vbz_gpu_decode_paths(vbz_gpu_ctx* ctx, uint32_t* batched, uint32_t* walked);
/* ... and for how many of the walked frames the literals of the first block were decoded beside the walk (64 pieces of the four Huffman
 * streams, one lane each, ahead of the general decoder; VBZ_HIP_REF_LITERALS=0: never).  A diagnostic like the above; < 0 on a device error. */
VBZ_EXPORT int vbz_gpu_decode_literals_ahead(vbz_gpu_ctx* ctx);
/* The same for the large-read path (few, large reads; synchronizes): *by_spans = frames of the last decompress launch group that were
 * decoded span by span as their index says (or, without an index, by one wavefront at once) -- the others needed the second, gated
 * launch of the one-wavefront decoder.  Returns the frames of that group, 0 when it did not run on the large-read path. */
VBZ_EXPORT int vbz_gpu_decode_span_paths(vbz_gpu_ctx* ctx, uint32_t* by_spans);

/* Version string of the library: "vbz_hip <semver> gfx950". */
VBZ_EXPORT const char* vbz_gpu_version(void);

#if defined(__cplusplus)
}
#endif
#endif
