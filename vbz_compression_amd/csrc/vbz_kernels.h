// vbz_kernels.h -- internal interface between the C ABI (vbz_api.hip) and the HIP kernels.
// Not installed; the public surface is include/vbz.h and include/vbz_gpu.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vbzhip {

// vbz error codes (include/vbz.h), usable in device code
constexpr uint32_t E_ZSTD = 0xFFFFFFFFu;
constexpr uint32_t E_INPUT_SIZE = 0xFFFFFFFEu;
constexpr uint32_t E_INTEGER_SIZE = 0xFFFFFFFDu;
constexpr uint32_t E_DESTINATION_SIZE = 0xFFFFFFFCu;
constexpr uint32_t E_STREAM = 0xFFFFFFFBu;
constexpr uint32_t E_VERSION = 0xFFFFFFFAu;
constexpr uint32_t E_OOM = 0xFFFFFFF9u;
constexpr uint32_t E_DEVICE = 0xFFFFFFF8u;
constexpr uint32_t E_FIRST = E_DEVICE;
constexpr uint32_t GATE_SKIP = E_FIRST - 1u;   // gate value "not in this launch group"; gate >= GATE_SKIP: no work here
constexpr int PHASE_SLOTS = 12;  // per-read cycle counters of the timed kernel builds (debug aid)

// One batch of independent reads ("reads" in the reference's vocabulary: one HDF5 chunk each).
// All pointers are device pointers.  `result[i]` receives the bytes produced or an error code.
// If `gate` is non-null, reads whose gate[i] is an error code are skipped and the error is kept; reads whose gate[i] is
// GATE_SKIP belong to another launch group of the same call (per-read routing: vbz_api.hip): nothing of theirs is touched.
struct ReadBatch
{
    uint32_t n_reads;
    const uint8_t* src;
    const uint64_t* src_off;
    const uint32_t* src_size;
    uint8_t* dst;
    const uint64_t* dst_off;
    const uint32_t* dst_cap;
    uint32_t* result;
    const uint32_t* gate;
};

// ---- the per-read plans of the staged encoder (zstd_encode.hip), and what the svb encoder leaves in them ---------------------------
// The entropy stage turns every run of >= RMIN equal control bytes into a zstd sequence and Huffman-codes what is left, each region from
// its own byte histogram.  The one-wavefront path does that in stages that hand a read on through its EncPlan: the svb encoder counts
// the data bytes' sample on the way (int16 zig-zag reads: svb_kernels.hip CNT), zstd_plan_kernel tokenises the control bytes, codes their
// sequences section and builds both regions' tables (two wavefronts per read), zstd_pack_kernel packs.
#ifndef VBZ_RMIN
#define VBZ_RMIN 12
#endif
constexpr uint32_t RMIN = VBZ_RMIN;  // shortest run of equal bytes that becomes a match (break-even is ~13 bytes); >= 8, <= 24
struct EncRegionPlan
{
    uint32_t S, nblk, nrec, seqmode, Sh, treeSize, huffLog, pad;
    uint32_t ctable[256];   // code | length << 16.  (reg[1], before zstd_plan_kernel: the histogram the svb encoder left: EncPlan::hist_mode)
    uint32_t tree[34];      // the tree description (136 bytes)
};
struct EncPlan
{
    EncRegionPlan reg[2];
    uint32_t seqBytes, seqOff;      // the sequences section of region 0, coded by zstd_plan_kernel at the top of the destination slot
    uint32_t cpCount, cpSpacing;    // its decoder checkpoints (CP_MAGIC trailer)
    uint32_t cp[64];
    // the control bytes are tokenised IN PLACE (their literals compacted to the front of the region, the run records at the tail of the
    // scratch slot): whoever codes the read afterwards takes the result from here
    uint32_t tok_done, tok_nrec, tok_lit;
    // histogram of the data bytes the svb encoder left: 0 none; 1: reg[1].ctable = the data bytes region_histogram's sample counts (one
    // kilobyte in four + the unaligned ends); 2: and histB = the data bytes the sample leaves out (reads so short that the region may be
    // counted exactly)
    uint32_t hist_mode;
    uint32_t pad[8];
    uint32_t histB[256];
};
constexpr uint32_t PLAN_OPEN = 1, PLAN_REG0 = 2, PLAN_REG1 = 4, PLAN_READY = 7;
constexpr uint32_t ENC_TRAILERS = 1, ENC_PRE_FILLED = 2;   // bits of the entropy stage's `trailers` argument
inline EncPlan* zstd_encode_plans(void* plan_meta) { return reinterpret_cast<EncPlan*>(plan_meta); }   // the plans inside plan_meta

// ---- streamvbyte stage (svb_kernels.hip) -------------------------------------------------------
// integer_size in {1,2,4}; zigzag.
// hdr: 0, or 4 to prepend / skip the sized header (u32 LE original size) in front of the svb stream.
// strict_cap: apply the reference's worst-case capacity rule (keys + 4 bytes per value) to dst_cap.
// half: the v1 nibble codec for 1-byte integers (vbz/v1/vbz_streamvbyte_impl.h)
// period_hint (nullable, n_reads words): the encoder also looks for ONE long repeat distance in the data bytes it writes
// (svb_kernels.hip: PeriodProbe) and leaves it there (0: none) for the entropy stage's long-repeat coder.
// plans (nullable: the plan_meta of launch_zstd_encode; int16 zig-zag reads into library scratch only): every read's data bytes are
// counted on the way and the histogram left in its plan (EncPlan::hist_mode, written for EVERY read of the launch).
hipError_t launch_svb_encode(const ReadBatch& b, int integer_size, bool zigzag, uint32_t hdr, bool strict_cap, bool half, uint32_t* period_hint,
                             void* plans, hipStream_t s);
bool svb_encode_fills_plans(int integer_size, bool zigzag, bool half);   // does launch_svb_encode(plans) write every read's hist_mode?
hipError_t launch_svb_decode(const ReadBatch& b, int integer_size, bool zigzag, bool half, hipStream_t s);
// The same stage with one read spread over many workgroups ("segments" of svb_seg_unit_bytes raw bytes), for batches of few,
// large reads (one 10 M-element buffer, one 400 k-sample read): seg_first[n_reads + 1] from launch_seg_plan; max_segs
// bounds the total segment count (the grid); seg_* are scratch arrays of max_segs entries.  Not for the nibble codec.
uint32_t svb_seg_unit_bytes(int integer_size);
hipError_t launch_svb_encode_seg(const ReadBatch& b, int integer_size, bool zigzag, uint32_t hdr, bool strict_cap, const uint32_t* seg_first,
                                 uint32_t max_segs, uint32_t* seg_bytes, uint64_t* seg_off, hipStream_t s);
hipError_t launch_svb_decode_seg(const ReadBatch& b, int integer_size, bool zigzag, const uint32_t* seg_first, uint32_t max_segs,
                                 uint32_t* seg_val, uint64_t* seg_pos, uint32_t* seg_run, hipStream_t s);

// ---- zstd-format entropy stage (zstd_encode.hip / zstd_decode.hip) -----------------------------
// encode: frame content = src read; key_elem = integer size whose key section (ceil(n/4) bytes, n
// derived from `orig_size[i] / key_elem`) is split into its own blocks; 0 = no split.
// hdr: 0 or 4 (sized header carrying orig_size[i] in front of the frame).
// key_bytes (nullable) gives the key-section length per read directly and overrides key_elem.
// dbg (nullable): 8 x u64 per read, shader-clock cycles spent per phase (debug aid, VBZ_HIP_PHASE_TIMING=1)
// src_cap + seq_tables (both nullable): the source streams live in library-owned scratch slots of that
// capacity, which lets the encoder rewrite the control-byte region as literals + zero-run sequences.
// trailers: append the decoder-checkpoint skippable frame when a sequences section was written (see zstd_encode.hip)
// deep_d (nullable, n_reads words): the repeat distance launch_svb_encode's probe proposes for every read (0: none).  Reads
// whose proposal holds are coded by a second launch with the long-repeat matcher (what libzstd's match finder gets out of
// template-cycling signal, at every level); deep_d[r] is rewritten with the verdict.  The matcher's workspace is the top of
// the destination slot (reads whose slot is too small for frame and workspace are coded without it).
// plan_meta (nullable, zstd_encode_plan_bytes(n_reads) bytes of device scratch): the per-read plans.  staged: the ordinary read is
// coded by the staged launches (tokeniser + sequences section, tables, packing -- each at the occupancy its own footprint allows) and
// only what they leave over by the one-launch kernel; same frames either way.  pre_filled: launch_svb_encode(plans) has left every
// read's hist_mode (and histogram) in the plans.
hipError_t launch_zstd_encode(const ReadBatch& b, const uint32_t* orig_size, uint32_t key_elem, const uint32_t* key_bytes,
                              uint32_t hdr, unsigned long long* dbg, const uint32_t* src_cap, const void* seq_tables, bool trailers,
                              uint32_t* deep_d, void* plan_meta, bool staged, bool pre_filled, unsigned long long* pack_dbg, hipStream_t s);
size_t zstd_encode_plan_bytes(uint32_t n_reads);
size_t seq_tables_bytes();
void seq_tables_build(void* host_buffer);
// A second stream and two events for launches that run beside the main chain (the chain walk beside the launches for own frames, the
// shared-table spans beside the control-byte spans); stream == nullptr: none
struct SideStream
{
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
typedef SideStream FastSide;
// The same stage for batches of few, large reads: one wavefront per SPAN of a read's stream (see zstd_encode.hip).
// stream_bytes bounds the total of the source streams.  span_desc: max_spans x zstd_span_desc_bytes(); span_first[n_reads + 1];
// span_count[1]; span_size / span_trail / span_dst [max_spans]; span_tmp: zstd_span_tmp_bytes(...) bytes.
size_t zstd_span_desc_bytes();
uint32_t zstd_span_max_spans(uint64_t stream_bytes, uint32_t n_reads);  // 0: too large
uint64_t zstd_span_tmp_bytes(uint64_t stream_bytes, uint32_t n_reads, uint32_t max_spans);
hipError_t launch_zstd_encode_spans(const ReadBatch& b, const uint32_t* orig_size, uint32_t key_elem, uint32_t hdr, const uint32_t* src_cap,
                                    const void* seq_tables, void* span_desc, uint32_t* span_first, uint32_t* span_count, uint32_t max_spans,
                                    uint8_t* span_tmp, uint64_t span_tmp_bytes, uint32_t* span_size, uint32_t* span_trail, uint32_t* span_dst,
                                    bool trailers, void* shared_regions, uint32_t shared_span_bytes, hipStream_t s);  // trailers: checkpoints and span index behind the frame
// shared_regions (nullable: every span builds its own table): zstd_span_region_bytes(n_reads) bytes of device scratch, 16-byte aligned -- the
// data bytes of a read with a control-byte region get ONE table (counted and built by extra wavefronts of the launch that codes the
// control-byte spans) and are packed one wavefront per 8 KB span by the launch behind it.
// shared_span_bytes: zstd_span_shared_bytes(the call's stream bytes) -- 0 (pass shared_regions = nullptr then): a batch of large buffers,
// a matter of throughput, where every span builds its own table in one launch.
size_t zstd_span_region_bytes(uint32_t n_reads);
uint32_t zstd_span_shared_bytes(uint64_t stream_bytes);
// The long-repeat matcher in front of the span launches (batches too small to fill the device run as spans): a probe over
// every read below max_raw bytes, the check of launch_zstd_encode's first launch, and its matcher instantiation for the reads
// whose distance holds.  deep_d[n_reads] is scratch; gate_out[i] = GATE_SKIP for the reads coded here (the spans skip them),
// gate_in[i] otherwise.  b: as for launch_zstd_encode (source streams in library-owned slots of src_cap[i] bytes).
hipError_t launch_zstd_encode_matcher(const ReadBatch& b, const uint32_t* orig_size, uint32_t key_elem, uint32_t hdr, const uint32_t* src_cap,
                                      const void* seq_tables, bool trailers, uint32_t max_raw, uint32_t* deep_d, const uint32_t* gate_in,
                                      uint32_t* gate_out, hipStream_t s);
// decode: result[i] = frame content size, E_ZSTD for a malformed frame, or `toosmall_code` when the
// frame's content size exceeds dst_cap[i].
// seq_dtables (device, from seq_dtables_build): decoding tables of the predefined LL / ML distributions.
hipError_t launch_zstd_decode(const ReadBatch& b, uint32_t toosmall_code, unsigned long long* dbg, const void* seq_dtables,
                              hipStream_t s);
// The same, and the frame's content -- the svb stream of int16 zig-zag samples, b.dst = its slot in the library's scratch --
// is decoded by the same wavefront straight away (svb_wave.h) into out + out_off[i] (out_size[i] bytes exactly): result[i]
// gets what launch_svb_decode(2, zigzag) would have reported after launch_zstd_decode, and no svb_decode launch follows.
// (experiments build only: measured slower than the two launches)
#ifdef VBZ_EXPERIMENTS
hipError_t launch_zstd_decode_svb_i16zz(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, uint8_t* out, const uint64_t* out_off,
                                        const uint32_t* out_size, hipStream_t s);
#endif
// Frames the reference wrote (libzstd: blocks of general sequences): the sequence chains of the reads with redo[i] != 0 walked one
// lane per frame ahead of the one-wavefront decoder (zstd_decode_ref.hip), which takes a frame's records when RefPre.ok says so.
constexpr uint32_t REF_MAXBLK = 4;
struct RefBlock
{
    uint32_t pos;             // where the block's header stands in the frame
    uint32_t nseq;
    uint32_t rec_lo, rec_hi;  // its first record in the workspace ({literal length, match length, offset, 0})
    uint32_t rep[3];          // the repeat offsets behind the block
    uint32_t end;             // the output position behind the block
};
struct RefPre
{
    uint32_t ok, nblk, pad[6];
    RefBlock blk[REF_MAXBLK];
};
// The literals of a reference-written frame's first block, decoded ahead of the one-wavefront decoder BESIDE the chain walk (the walk is a
// few hundred wavefronts bound by latency; the literals are seven tenths of such a frame's cycles and need nothing of the chains):
// blk = where the block header sits in the frame (0: not done), regen / csize as its literals header says, at = where in the
// destination slot the literals stand.  The decoder takes them when its own reading of the block says the same four numbers.
struct RefLits
{
    uint32_t blk, regen, csize, at;
    uint32_t tb, pcap;           // the literals stand in 64 stripes of pcap bytes from offset tb of the slot; lits_pos[64 r + q]: the first literal of stripe q
    uint32_t tail, pad;          // tail != 0: literal x also stands at tail - regen + x of the slot -- where the literals behind the last sequence
                                 // belong if the block's content ends at `tail` (the frame's content size for a last block, else a full block)
};
struct RefChains  // what launch_zstd_decode_only needs of them (pre == nullptr: none)
{
    const RefPre* pre = nullptr;
    const void* recs = nullptr;
    const RefLits* lits = nullptr;   // (nullable; only looked at for frames whose chains are walked) record k of read r at k * n_reads + r, k < REF_MAXBLK ...
    const uint32_t* lits_pos = nullptr;   // ... and 64 words per record
    uint32_t lits_units = 0;              // records per read this call has filled (k < lits_units)
};

size_t zstd_ref_pre_bytes(uint32_t n_reads);
const RefPre* zstd_ref_pre(const void* pre_meta);  // the per-frame hand-overs inside pre_meta
size_t zstd_ref_table_bytes(uint32_t n_reads);     // 0: a batch of this size keeps its tables in LDS
// pre_meta: zstd_ref_pre_bytes(n_reads); tables: zstd_ref_table_bytes(n_reads); recs: recs_cap records of 16 bytes.  *out: for the decoder.
hipError_t launch_zstd_ref_chain(const ReadBatch& b, const uint32_t* redo, void* pre_meta, void* tables, void* recs, uint64_t recs_cap, RefChains* out,
                                 hipStream_t s);
// The one-wavefront decoder for the reads with only[i] != 0 (the others are left alone); dbg: phase cycle counters (nullable).
hipError_t launch_zstd_decode_only(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, const uint32_t* only, RefChains chains,
                                   unsigned long long* dbg, hipStream_t s);
// Batched decoder for frames of the shape zstd_encode.hip writes (zstd_decode_fast.hip: one lane per frame for the headers, one lane
// per tree description, one wavefront per frame for nothing but the streams, one for the zero-run block); every frame that is not of
// that shape or fails a check, and every error verdict, goes through launch_zstd_decode_only at the end.  Same results as
// launch_zstd_decode.  meta: zstd_fast_meta_bytes(n_reads) bytes of device scratch.
// ref_*: scratch of launch_zstd_ref_chain (ref_pre == nullptr: frames of other writers go to the one-wavefront decoder as they are).
// dbg (nullable): phase cycle counters of the one-wavefront decoder, which then decodes EVERY frame (walked chains included).
size_t zstd_fast_meta_bytes(uint32_t n_reads);
const RefLits* zstd_ref_lits(const void* lit_meta, uint32_t n_reads);   // (diagnostics: zstd_ref_lit_units() records per read; blk != 0 = that block's literals were decoded ahead)
bool zstd_ref_literals_enabled();                                   // VBZ_HIP_REF_LITERALS
const uint32_t* zstd_fast_redo(const void* meta, uint32_t n_reads);  // after the call: redo[i] == 0 <=> frame i was decoded by the batched decoder
hipError_t launch_zstd_decode_fast(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, void* meta, void* ref_pre, void* ref_tables,
                                   void* ref_recs, uint64_t ref_recs_cap, void* ref_lits, uint32_t ref_units, unsigned long long* dbg, FastSide side,
                                   hipStream_t s);
// ref_lits (nullable): zstd_ref_lit_meta_bytes(n_reads) bytes for the literals decoded beside the walk (zstd_ref_lit_units() records per read;
// ref_units: how many of them this call fills -- blocks per frame that get a workgroup)
size_t zstd_ref_lit_meta_bytes(uint32_t n_reads);
uint32_t zstd_ref_lit_units();
const uint32_t* zstd_ref_lit_skip(const void* lit_meta, uint32_t n_reads);   // (diagnostics: 0 = the scan made the block a unit)
size_t seq_dtables_bytes();
void seq_dtables_build(void* host_buffer);
// The same for batches of few, large reads: frames that carry the encoder's span index are decoded one span per wavefront
// (verified; anything else, and every error verdict, comes from the ordinary decoder in a second launch gated by redo[]).
// content_bytes bounds the total frame content.  dspan_desc: max_spans x zstd_dspan_desc_bytes(); dspan_first[n_reads + 1];
// dspan_count[1]; dspan_status[4 * max_spans]; redo[n_reads].
size_t zstd_dspan_desc_bytes();
uint32_t zstd_dspan_max_spans(uint64_t content_bytes, uint32_t n_reads);  // 0: too large
hipError_t launch_zstd_decode_spans(const ReadBatch& b, uint32_t toosmall_code, const void* seq_dtables, void* dspan_desc, uint32_t* dspan_first,
                                    uint32_t* dspan_count, uint32_t max_spans, uint32_t* dspan_status, uint32_t* redo, hipStream_t s);

// ---- helpers (helpers.hip) ---------------------------------------------------------------------
// scratch slots for the intermediate svb streams: slot(i) = align16(ceil(raw_size[i]*num/den)+8)+48,
// off[i] = exclusive scan + 16, cap[i] = slot - 32; gate[i] = E_OOM if the slot exceeds `limit` bytes.
// gate_is_input: gate[] already holds per-read errors; those reads keep their error and get an empty slot.
hipError_t launch_plan_scratch(uint32_t n, const uint32_t* raw_size, uint32_t mul_num, uint32_t mul_den, uint64_t limit,
                               uint64_t* off, uint32_t* cap, uint32_t* gate, bool gate_is_input, hipStream_t s);
// seg_first[i] = number of segments of reads 0..i-1, a read of `size` bytes having max(1, ceil(size / unit_bytes)) of them
// (seg_first[n] = total); reads whose gate is an error get one segment.  max_segs = the size of the caller's segment tables
// (and grids): reads whose segments would not fit them (sources that alias each other can add up to more than the arena)
// keep one segment and get gate_out[i] = E_OOM; gate_out[i] = gate[i] (or 0) otherwise.  gate_out may be gate.
// scratch (nullable; used when n <= 1024): the scratch plan of launch_plan_scratch for the same reads in the same launch -- off / cap as
// there, gate[i] = gate_out[i] or E_OOM for a read whose slot does not fit.
struct ScratchPlan
{
    uint32_t num, den;
    uint64_t limit;
    uint64_t* off;
    uint32_t* cap;
    uint32_t* gate;
};
hipError_t launch_seg_plan(uint32_t n, const uint32_t* size, uint32_t unit_bytes, const uint32_t* gate, uint32_t max_segs, uint32_t* seg_first,
                           uint32_t* gate_out, const ScratchPlan* scratch, hipStream_t s);
hipError_t launch_count_nonzero(const uint32_t* a, uint32_t n, uint32_t* out, hipStream_t s);   // *out = the words of a[0..n) that are not zero
// canonical mode: see canon_classify_kernel (helpers.hip).  counts: two words of device memory.
hipError_t launch_canon_classify(uint32_t n, const uint32_t* raw_size, const uint32_t* gate, uint32_t min_bytes, uint32_t* gate_small, uint32_t* gate_large,
                                 uint32_t* counts, hipStream_t s);
// per-read routing: see route_reads_kernel (helpers.hip).  raw_size[i] = the read's raw (decoded) byte count.
hipError_t launch_route_reads(const ReadBatch& b, const uint32_t* raw_size, uint32_t min_bytes, uint32_t max_reads, uint64_t max_bytes, uint32_t* gate_small,
                              uint64_t* l_src_off, uint32_t* l_src_size, uint64_t* l_dst_off, uint32_t* l_dst_cap, uint32_t* l_gate, uint32_t* l_map,
                              uint32_t* l_count, uint32_t* cand, hipStream_t s);   // cand: route_cand_words() words of scratch
size_t route_cand_words();
hipError_t launch_route_results(const uint32_t* l_result, const uint32_t* l_map, const uint32_t* l_count, uint32_t max_reads, uint32_t* result, hipStream_t s);
// sized decode: read the 4-byte headers -> payload offsets/sizes, original sizes, gate errors (gate_in, nullable: reads that
// already carry an error keep it and are not looked at)
hipError_t launch_parse_sized(uint32_t n, const uint8_t* src, const uint64_t* src_off, const uint32_t* src_size,
                              const uint32_t* dst_cap, const uint32_t* gate_in, uint64_t* pay_off, uint32_t* pay_size, uint32_t* orig_size,
                              uint32_t* gate, hipStream_t s);
// descriptor table against the declared arenas: gate[i] = 0, E_INPUT_SIZE (source slot outside [0, src_bytes)) or
// E_DESTINATION_SIZE (destination slot outside [0, dst_bytes)); 64-bit arithmetic
// The single-buffer API's hand-back: `*result` bytes at `src` (if no error code and <= host_cap) into pinned host memory at host + 16
// words, the result word to host[0], the bytes copied to host[1], then host[2] = seq (system-scope release): what the host polls.
// ticket: one zeroed word of device memory the launch's workgroups count themselves on (left at zero).
hipError_t launch_hand_back(const uint32_t* result, const uint8_t* src, uint32_t* host, uint32_t host_cap, uint32_t seq, uint32_t* ticket, hipStream_t s);
hipError_t launch_validate_batch(uint32_t n, const uint64_t* src_off, const uint32_t* src_size, uint64_t src_bytes, const uint64_t* dst_off,
                                 const uint32_t* dst_cap, uint64_t dst_bytes, uint32_t* gate, hipStream_t s);
// integer_size == 0 && level == 0: per-read copy (reference vbz/vbz.cpp:130-133)
hipError_t launch_copy_bytes(const ReadBatch& b, uint32_t hdr, hipStream_t s);
hipError_t launch_synth_lengths(uint64_t seed, uint64_t first, uint32_t n, uint32_t* out_len, hipStream_t s);
hipError_t launch_synth_signal(uint64_t seed, uint64_t first, uint32_t n, uint8_t* dst, const uint64_t* off,
                               const uint32_t* len, hipStream_t s);
hipError_t launch_synth_u32(uint64_t seed, uint64_t first, uint32_t n, uint8_t* dst, const uint64_t* off,
                            const uint32_t* len, hipStream_t s);

// ---- wave / workgroup primitives ---------------------------------------------------------------
// For single-wave workgroups.  The LDS operations of one wave execute in issue order, so lanes of the same
// wave can hand data to each other through LDS without an s_barrier -- and, unlike __syncthreads(), without
// waiting for the wave's outstanding global stores.  This only pins the program order for the compiler.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding global
// loads AND stores (s_waitcnt vmcnt(0)), which serialises a tile loop on memory latency; the streaming kernels
// only ever exchange data through LDS, so they wait for LDS alone and keep their global accesses in flight.
__device__ __forceinline__ void wg_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The value of the lane below (lane 0 gets 0): one DPP move (wave_shr:1) instead of an LDS-crossbar shuffle.
__device__ __forceinline__ uint32_t wave_prev_lane_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}

// The value of the lane above (lane 63 gets 0): wave_shl:1.
__device__ __forceinline__ uint32_t wave_next_lane_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false);
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts (gfx9: no LDS crossbar
// traffic, six additions): after the four row_shr steps every 16-lane row holds its own scan, row_bcast:15 adds the
// last lane of rows 0 and 2 into rows 1 and 3, row_bcast:31 adds lane 31 into the upper half.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
#define VBZ_DPP_ADD(ctrl, rowmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xF, false)
    VBZ_DPP_ADD(0x111, 0xF);  // row_shr:1
    VBZ_DPP_ADD(0x112, 0xF);  // row_shr:2
    VBZ_DPP_ADD(0x114, 0xF);  // row_shr:4
    VBZ_DPP_ADD(0x118, 0xF);  // row_shr:8
    VBZ_DPP_ADD(0x142, 0xA);  // row_bcast:15 -> rows 1, 3
    VBZ_DPP_ADD(0x143, 0xC);  // row_bcast:31 -> rows 2, 3
#undef VBZ_DPP_ADD
    return v;
}

// exclusive scan over a 256-thread workgroup; wsum = 4 words of LDS; returns the exclusive prefix,
// `total` gets the workgroup sum.  Contains two (LDS-only) barriers.
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* wsum, uint32_t& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan_u32(v);
    if (lane == 63) wsum[w] = inc;
    wg_lds_barrier();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t s = wsum[k];
        base += (k < w) ? s : 0u;
        tot += s;
    }
    wg_lds_barrier();
    total = tot;
    return base + inc - v;
}

}  // namespace vbzhip
