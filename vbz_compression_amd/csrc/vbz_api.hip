// vbz_api.hip -- the C ABI of libvbz_hip.so (include/vbz.h, include/vbz_gpu.h).
//
// Host side of the drop-in boundary.  It mirrors the control flow of the reference's vbz/vbz.cpp
// (option validation, size rules, stage chaining, error codes -- cited per function) and drives the
// HIP kernels; there is no CPU implementation of either stage in this library.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <map>
#include <vector>

#include "../../include/vbz.h"
#include "../../include/vbz_gpu.h"
#include "vbz_kernels.h"

using namespace vbzhip;


// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct ProfEntry
{
    const char* name;
    uint32_t launches;
    double ms;
};

struct PendingEvent
{
    const char* name;
    hipEvent_t start, stop;
};

struct DevBuf
{
    void* p = nullptr;
    size_t cap = 0;
};

struct vbz_gpu_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string error;
    DevBuf scratch;   // intermediate svb streams of a batch
    DevBuf meta;      // per-read bookkeeping arrays
    DevBuf gmeta;     // ... of one launch group of a decompress call
    DevBuf route;     // per-read routing: the first group's gates, the second group's compact descriptors
    int routing = 1;       // VBZ_HIP_ROUTING=0: by batch shape only (2: experiment, the second group is not launched)
    vbz_gpu_ctx* large = nullptr;   // per-read routing: the second group's own stream and buffers (it runs beside the first group)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // A large batch on the one-workgroup path is coded as TWO HALVES on two streams (split_batch): the upper half on this child context
    vbz_gpu_ctx* half = nullptr;
    hipEvent_t ev_hfork = nullptr, ev_hjoin = nullptr, ev_hstagger = nullptr;
    DevBuf splitmeta;          // the scratch plan of the whole batch (slot offsets, capacities, gates), shared by the two halves
    uint32_t split_min = 16384;   // VBZ_HIP_SPLIT_MIN: batches of this many reads and more are split (0: never)
    int split_stagger = 0;        // VBZ_HIP_SPLIT_STAGGER=1: the upper half starts behind the lower half's first large launch (measured: 548 GB/s
                                  // against 565 when both halves start together -- the device interleaves the two queues by itself)
    bool last_split = false;      // vbz_gpu_decode_paths: the last decompress call ran as halves
    // Frames of OTHER writers (every vbz file in existence: the reference's libzstd frames) decode on one wavefront each behind a chain walk
    // of fixed latency, and two halves of such a call do not run beside each other (482 GB/s as one group, 420 as halves); the host
    // does not know whose frames a call holds.  Every decompress call leaves the number of frames the batched decoder did not take in
    // pinned memory (a counting launch and a 4-byte copy behind the call, no wait); the NEXT call looks at it if it has arrived and runs
    // as one group when most frames were foreign.  A matter of speed only: the decoded bytes do not depend on it.
    uint32_t* foreign_host = nullptr;   // pinned: [frames left to the one-wavefront decoder, frames looked at]
    DevBuf foreign_dev;
    hipEvent_t ev_foreign = nullptr;
    bool foreign_pending = false, mostly_foreign = false;
    int foreign_state = -1;   // what the last call that was looked at held: -1 not known, 0 no foreign frame at all, 1 some
    // Canonical encoding (vbz_gpu_set_canonical / VBZ_HIP_CANONICAL=1): a read's compressed bytes are a function of the read, the options
    // and the library version -- not of the batch it arrives in (see compress_canonical)
    bool canonical = false;
    uint32_t* canon_host = nullptr;   // pinned: the classification's count comes back here
    // single-buffer API staging
    DevBuf one_in, one_out, one_meta;
    DevBuf dbg;       // per-read phase timers (VBZ_HIP_PHASE_TIMING=1)
    DevBuf seqtab;    // encoding tables of the predefined sequence distributions
    DevBuf seqdtab;   // decoding tables of the same distributions
    DevBuf segmeta;   // segment / span tables of the large-read path
    DevBuf spanmeta, spantmp;  // span tables and temporary slots of the entropy stage in the large-read path
    DevBuf vgate;     // verdicts on the caller's descriptor table (one word per read)
    DevBuf encplan;   // per-read plans of the staged encoder (tables of both regions: zstd_encode.hip STAGE 1 / 2)
    bool staged_encode = true;  // VBZ_HIP_STAGED_ENCODE=0: the fused encoder kernel for every read
    bool shared_tables = true;  // VBZ_HIP_SHARED_TABLES=0: large-read path, every span of the data bytes with a table of its own (round 4's frames)
    DevBuf fastmeta;  // per-frame descriptors, stream tasks and weights of the batched own-frame decoder (zstd_decode_fast.hip)
    bool fast_decode = true;   // VBZ_HIP_FAST_DECODE=0: every frame through the one-wavefront decoder
    DevBuf refpre, reftab, refrecs;  // frames the reference wrote: per-frame hand-over, tables (large batches), records (zstd_decode_ref.hip)
    DevBuf reflits;                  // ... their literals decoded beside the walk (zstd_decode_fast.hip: ref_pieces_kernel)
    uint32_t last_lit_units = 0;     // (records per read the last call filled)
    int ref_chains = 1;        // VBZ_HIP_REF_CHAINS: 0 their sequence chains are walked by the one-wavefront decoder itself, 1 walked ahead of
                               // it in calls of REF_MIN_READS reads and more, 2 in every call
    FastSide side;             // the walk's own stream (beside the launches for this library's frames)
    bool last_walked = false;
    uint32_t last_frames = 0;  // vbz_gpu_decode_paths: the frames of the last zstd_frames call (0: none, or not on the batched path)
    uint32_t last_span_frames = 0;          // ... of the last call on the large-read path, and its redo[] (in spanmeta)
    const uint32_t* last_span_redo = nullptr;
    bool trailers = true;      // decoder hints (checkpoints, span index) in skippable frames behind the zstd frame
    int segmented = -1;  // -1: by batch shape; 0 / 1: forced (VBZ_HIP_SEGMENTED, for tests)
    bool zero_run_sequences = true;
    bool fuse_svb = false;     // VBZ_HIP_FUSE_SVB=1: the frame's wavefront decodes the svb stream too (measured slower: DESIGN.md 4.4)
    int long_repeats = 1;  // VBZ_HIP_LONG_REPEATS=0: no search for a repeat distance (experiments: 2 = probe only, 3 = second launch only)
    int phase_timing = 0;      // VBZ_HIP_PHASE_TIMING: 1 phase counters of the entropy kernels (one launch per frame), 2 / 3 of the encoder's planning / packing launch (staged, under load)
    bool trace = false;        // VBZ_HIP_TRACE=1: synchronise after every launch group and name it on stderr (to find a faulting kernel)
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    uint32_t one_seq = 0;      // the single-buffer API's hand-back flag: a number per call (run_one)
    bool profiling = false;
    std::vector<PendingEvent> pending;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
};

namespace {

void set_error(vbz_gpu_ctx* c, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf;
    fprintf(stderr, "vbz_hip: %s\n", buf);
}

bool ensure(vbz_gpu_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return true;
    if (b.p) {
        (void)hipStreamSynchronize(c->stream);  // earlier work may still use the old buffer
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = bytes > ((size_t)1 << 60) ? hipErrorOutOfMemory : hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        set_error(c, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        (void)hipGetLastError();   // (the runtime keeps the failure as the thread's last error: the caller's next HIP call must not trip over it)
        b.p = nullptr;
        return false;
    }
    b.cap = want;
    return true;
}

hipEvent_t get_event(vbz_gpu_ctx* c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct Timed  // records a pair of events around one kernel launch when profiling is on
{
    vbz_gpu_ctx* c;
    const char* name;
    hipEvent_t start = nullptr;
    Timed(vbz_gpu_ctx* ctx, const char* n) : c(ctx), name(n)
    {
        if (c->profiling) {
            start = get_event(c);
            (void)hipEventRecord(start, c->stream);
        }
    }
    ~Timed()
    {
        if (c->trace) {
            const hipError_t e = hipStreamSynchronize(c->stream);
            fprintf(stderr, "vbz_hip trace: %s %s\n", name, e == hipSuccess ? "ok" : hipGetErrorString(e));
        }
        if (start) {
            hipEvent_t stop = get_event(c);
            (void)hipEventRecord(stop, c->stream);
            c->pending.push_back({ name, start, stop });
        }
    }
};

void drain_profile(vbz_gpu_ctx* c)
{
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, p.start, p.stop);
        bool found = false;
        for (auto& e : c->prof)
            if (e.name == p.name || strcmp(e.name, p.name) == 0) {
                e.launches++;
                e.ms += ms;
                found = true;
                break;
            }
        if (!found) c->prof.push_back({ p.name, 1, (double)ms });
        c->event_pool.push_back(p.start);
        c->event_pool.push_back(p.stop);
    }
    c->pending.clear();
}

// debug aid: per-phase shader-clock cycles of the entropy kernels, averaged over the batch, on stderr
unsigned long long* dbg_begin(vbz_gpu_ctx* c, uint32_t n)
{
    if (!c->phase_timing) return nullptr;
    if (!ensure(c, c->dbg, (size_t)n * PHASE_SLOTS * 8)) return nullptr;
    (void)hipMemsetAsync(c->dbg.p, 0, (size_t)n * PHASE_SLOTS * 8, c->stream);
    return (unsigned long long*)c->dbg.p;
}

void dbg_end(vbz_gpu_ctx* c, uint32_t n, const char* what, unsigned long long* d)
{
    if (!d) return;
    std::vector<unsigned long long> h((size_t)n * PHASE_SLOTS);
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h.data(), d, (size_t)n * PHASE_SLOTS * 8, hipMemcpyDeviceToHost);
    double sum[PHASE_SLOTS] = {};
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < PHASE_SLOTS; ++k) sum[k] += (double)h[(size_t)i * PHASE_SLOTS + k];
    fprintf(stderr, "vbz_hip phase cycles/read (%s, n=%u):", what, n);
    for (int k = 0; k < PHASE_SLOTS; ++k) fprintf(stderr, " p%d=%.0f", k, sum[k] / n);
    fprintf(stderr, "\n");
}

// v1 codes 1-byte integers with the nibble codec (vbz/v1/vbz_streamvbyte.cpp:22-113)
bool half_codec(const CompressionOptions* o) { return o->vbz_version == 1 && o->integer_size == 1; }

bool valid_int_size(const CompressionOptions* o)  // vbz/vbz.cpp:44-50
{
    return o->integer_size == 0 || o->integer_size == 1 || o->integer_size == 2 || o->integer_size == 4;
}

// worst-case svb bytes per raw byte as a fraction, for what the device encoder can really emit
void svb_factor(unsigned integer_size, bool zigzag, uint32_t* num, uint32_t* den)
{
    if (integer_size == 2 && zigzag) { *num = 9; *den = 8; }     // keys n/4 + 2 bytes per value
    else if (integer_size == 1) { *num = 17; *den = 4; }         // keys n/4 + 4 bytes per value
    else if (integer_size == 2) { *num = 17; *den = 8; }
    else { *num = 17; *den = 16; }
}

size_t zstd_bound(size_t n)  // published ZSTD_COMPRESSBOUND (zstd.h), what vbz/vbz.cpp:109 calls
{
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

// The entry points run on the context's device and leave the caller's current device as they found it (a process
// that drives several GPUs from one thread must not have its later allocations land on another device).
struct DeviceGuard
{
    int saved = -1;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
        if (saved != device) (void)hipSetDevice(device);
        else saved = -1;
    }
    ~DeviceGuard()
    {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

#define HIPCHK(c, expr, what)                                                      \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            set_error((c), "%s failed: %s", (what), hipGetErrorString(e__));      \
            (void)hipGetLastError(); /* reported here: not left for the caller's next HIP call to trip over */ \
            return -1;                                                             \
        }                                                                          \
    } while (0)

// carve n-element arrays out of ctx->meta
struct MetaCarver
{
    uint8_t* p;
    size_t off = 0;
    explicit MetaCarver(void* base) : p((uint8_t*)base) {}
    template <typename T>
    T* take(size_t n)
    {
        off = (off + 15) & ~(size_t)15;
        T* r = (T*)(p + off);
        off += n * sizeof(T);
        return r;
    }
};

// One read per workgroup (svb) / per wavefront (entropy stage) fills the GPU when a batch has thousands of reads.  A batch of
// few, large reads (BASELINE configs[0] and [3]: one 400 k-sample read, one 10 M-element buffer; the HDF5 filter's one chunk
// per call) takes the segmented kernels instead: every read is spread over many workgroups.  The rule looks at the batch
// shape only (the sizes themselves live on the device): average read of half a megabyte or more.
constexpr uint64_t SEGMENTED_MIN_AVG = 512u << 10;
// ... and batches too small to fill the device with one wavefront per read: a call with one read of 100 k samples (the
// single-buffer API, the HDF5 filter) takes 0.39 / 0.25 ms on one wavefront and 0.16 / 0.16 ms as spans (tools/time_one_read.py),
// whatever the read's length; the one-wavefront kernels win from a few thousand reads per call on, or when the reads are
// so short that a read is hardly more than one span.
constexpr uint64_t SMALL_BATCH_MIN_AVG = 64u << 10, SMALL_BATCH_MAX_BYTES = 96u << 20;
// Decoding a handful of short reads -- a chunk per call through the HDF5 filter -- is a chain of nine launches on the one-wavefront
// path (scan, weights, streams, runs, the gated second attempt, ...: 0.17 - 0.23 ms for 1 - 64 reads of 5 - 10 k samples) and six with
// spans (0.13 - 0.19 ms, tools/time_small_batch.py); encoding such reads is faster on one wavefront each (0.11 - 0.13 against 0.12 - 0.20 ms).
constexpr uint64_t TINY_DECODE_MIN_AVG = 8u << 10;
constexpr uint32_t TINY_DECODE_MAX_READS = 64;

// The context's second stream (the chain walk beside the launches for own frames; the shared-table spans beside the control-byte
// spans): all three objects or none -- a context must never keep a stream without its events.
bool ensure_side(vbz_gpu_ctx* c)
{
    if (c->side.stream) return true;
    FastSide side;
    const bool ok = hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&side.fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&side.join, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        const char* why = hipGetErrorString(hipGetLastError());
        if (side.join) (void)hipEventDestroy(side.join);
        if (side.fork) (void)hipEventDestroy(side.fork);
        if (side.stream) (void)hipStreamDestroy(side.stream);
        set_error(c, "second stream: %s", why);
        return false;
    }
    c->side = side;
    return true;
}

bool use_segments(const vbz_gpu_ctx* c, uint64_t raw_arena_bytes, uint32_t n, bool decode)
{
    if (c->segmented >= 0) return c->segmented != 0;
    const uint64_t avg = raw_arena_bytes / n;
    return avg >= SEGMENTED_MIN_AVG || (avg >= SMALL_BATCH_MIN_AVG && raw_arena_bytes <= SMALL_BATCH_MAX_BYTES) ||
           (decode && n <= TINY_DECODE_MAX_READS && avg >= TINY_DECODE_MIN_AVG);
}

struct SegTables
{
    uint32_t* first = nullptr;  // [n + 1]
    uint32_t* val = nullptr;    // [max_segs]
    uint64_t* off = nullptr;    // [max_segs]
    uint32_t* run = nullptr;    // [max_segs]
    uint32_t* gate = nullptr;   // [n]: the input gate, plus E_OOM for reads whose segments do not fit the tables
    uint32_t max_segs = 0;
};

// segment tables for a batch whose raw bytes span `raw_arena_bytes`; sizes (device) are the reads' raw byte counts
// scratch (nullable): the group's scratch plan in the same launch (calls of up to 1024 reads: *scratch_done says whether it was)
int plan_segments(vbz_gpu_ctx* c, uint32_t n, const uint32_t* raw_size, const uint32_t* gate, uint64_t raw_arena_bytes, uint32_t unit, SegTables* t,
                  const ScratchPlan* scratch = nullptr, bool* scratch_done = nullptr)
{
    const uint64_t segs = raw_arena_bytes / unit + n + 1;
    if (segs > 0x7FFFFFFFull) {
        set_error(c, "batch too large for the segmented path");
        return -1;
    }
    t->max_segs = (uint32_t)segs;
    if (!ensure(c, c->segmeta, ((size_t)n + 1) * 8 + (size_t)segs * 16 + 256)) return -1;
    MetaCarver mc(c->segmeta.p);
    t->first = mc.take<uint32_t>((size_t)n + 1);
    t->gate = mc.take<uint32_t>(n);
    t->val = mc.take<uint32_t>(segs);
    t->off = mc.take<uint64_t>(segs);
    t->run = mc.take<uint32_t>(segs);
    Timed tm(c, "seg_plan");
    HIPCHK(c, launch_seg_plan(n, raw_size, unit, gate, t->max_segs, t->first, t->gate, scratch, c->stream), "seg_plan launch");
    if (scratch_done) *scratch_done = scratch != nullptr && n <= 1024;
    return 0;
}

ReadBatch to_rb(const vbz_gpu_batch* bt)
{
    ReadBatch rb;
    rb.n_reads = bt->n_reads;
    rb.src = (const uint8_t*)bt->src;
    rb.src_off = bt->src_off;
    rb.src_size = bt->src_size;
    rb.dst = (uint8_t*)bt->dst;
    rb.dst_off = bt->dst_off;
    rb.dst_cap = bt->dst_cap;
    rb.result = bt->result;
    rb.gate = nullptr;
    return rb;
}

// The scratch plan of a batch made ONCE for both halves of a split call (split_batch): the group then takes its slots from here
// instead of planning its own, and its scratch is the planning context's.
struct Preplanned
{
    void* scratch = nullptr;
    uint64_t* svb_off = nullptr;
    uint32_t *svb_cap = nullptr, *gate = nullptr;
    hipEvent_t after_first = nullptr;   // recorded on the group's stream behind its first large launch (the other half may wait for it)
};

// One launch group of a compress call: the reads of rb_in (those whose gate is closed left alone), on the one-workgroup path or
// on the large-read path.  src_bytes: extent of the raw bytes of the group's reads.
int compress_group(vbz_gpu_ctx* c, const ReadBatch& rb_in, uint64_t src_bytes, const CompressionOptions* o, int sized, bool segmented,
                   const Preplanned* pre = nullptr)
{
    const uint32_t n = rb_in.n_reads;
    if (n == 0) return 0;
    const uint32_t hdr = sized ? 4u : 0u;
    hipStream_t s = c->stream;
    ReadBatch rb = rb_in;
    struct { const uint32_t* src_size; uint64_t src_bytes; } view = { rb_in.src_size, src_bytes };
    const auto* bt = &view;   // (the body below was written against the batch descriptor)
    if (o->integer_size == 0 && o->zstd_compression_level == 0) {  // vbz.cpp:130-133
        Timed t(c, "copy_bytes");
        HIPCHK(c, launch_copy_bytes(rb, hdr, s), "copy launch");
        return 0;
    }
    SegTables seg;
    // svb into scratch, then the entropy stage into dst (vbz.cpp:163-207): the scratch slots are planned with the segments when they can be
    const bool both_stages = o->integer_size != 0 && o->zstd_compression_level != 0;
    uint32_t num = 1, den = 1;
    size_t scratch_need = 0;
    uint64_t* svb_off = nullptr;
    uint32_t *svb_cap = nullptr, *svb_size = nullptr, *gate = nullptr, *deep_d = nullptr;
    bool scratch_planned = false;
    MetaCarver mc(nullptr);
    if (both_stages) {
        svb_factor(o->integer_size, o->perform_delta_zig_zag, &num, &den);
        scratch_need = (size_t)(((unsigned __int128)bt->src_bytes * num + den - 1) / den) + (size_t)n * 96 + 256;
        if (!pre && !ensure(c, c->scratch, scratch_need)) return -1;
        if (!ensure(c, c->meta, (size_t)n * 40 + 256)) return -1;
        mc = MetaCarver(c->meta.p);
        svb_off = mc.take<uint64_t>(n);
        svb_cap = mc.take<uint32_t>(n);
        svb_size = mc.take<uint32_t>(n);
        gate = mc.take<uint32_t>(n);
        deep_d = mc.take<uint32_t>(n);
        if (pre) {   // (both stages, one-workgroup path: what split_batch asks for)
            svb_off = pre->svb_off;
            svb_cap = pre->svb_cap;
            gate = pre->gate;
            scratch_planned = true;
        }
    }
    void* const scratch_base = pre ? pre->scratch : c->scratch.p;
    const ScratchPlan splan = { num, den, c->scratch.cap, svb_off, svb_cap, gate };
    if (segmented && plan_segments(c, n, bt->src_size, rb_in.gate, bt->src_bytes, svb_seg_unit_bytes((int)o->integer_size), &seg,
                                   both_stages ? &splan : nullptr, &scratch_planned) != 0)
        return -1;
    if (segmented) rb.gate = seg.gate;   // (E_OOM for a read whose segments do not fit the tables)
    if (o->integer_size != 0 && o->zstd_compression_level == 0) {  // vbz.cpp:171-192: svb straight into dst
        Timed t(c, "svb_encode");
        if (segmented)
            HIPCHK(c, launch_svb_encode_seg(rb, (int)o->integer_size, o->perform_delta_zig_zag, hdr, true, seg.first, seg.max_segs, seg.val, seg.off, s),
                   "svb_encode (segmented) launch");
        else
            HIPCHK(c, launch_svb_encode(rb, (int)o->integer_size, o->perform_delta_zig_zag, hdr, true, half_codec(o), nullptr, nullptr, s), "svb_encode launch");
        return 0;
    }
    if (o->integer_size == 0) {  // zstd only
        Timed t(c, "zstd_encode");
        HIPCHK(c, launch_zstd_encode(rb, bt->src_size, 0, nullptr, hdr, nullptr, nullptr, nullptr, c->trailers, nullptr, nullptr, false, false, nullptr, s), "zstd_encode launch");
        return 0;
    }
    // the long-repeat matcher (every level: the reference's libzstd matches at every level; its workspace is the top of the
    // destination slots)
    const bool matcher = !segmented && c->zero_run_sequences && c->long_repeats;
    if (!scratch_planned) {
        if (rb.gate) HIPCHK(c, hipMemcpyAsync(gate, rb.gate, 4ull * n, hipMemcpyDeviceToDevice, s), "gate copy");
        Timed t(c, "plan_scratch");
        HIPCHK(c, launch_plan_scratch(n, bt->src_size, num, den, c->scratch.cap, svb_off, svb_cap, gate, rb.gate != nullptr, s), "plan launch");
    }
    ReadBatch a = rb;
    a.dst = (uint8_t*)scratch_base;
    a.dst_off = svb_off;
    a.dst_cap = svb_cap;
    a.result = svb_size;
    a.gate = gate;
    // the per-read plans of the entropy stage (one-wavefront path, staged encoder), in which the svb encoder leaves the data bytes'
    // histogram of int16 zig-zag reads (svb_kernels.hip CNT)
    unsigned long long* dbg = segmented ? nullptr : dbg_begin(c, n);
    const bool staged = !segmented && c->staged_encode && (!dbg || c->phase_timing == 3) && c->zero_run_sequences;
    const bool pre_filled = staged && svb_encode_fills_plans((int)o->integer_size, o->perform_delta_zig_zag, half_codec(o));
    void* plan = nullptr;
    if (staged) {
        if (!ensure(c, c->encplan, zstd_encode_plan_bytes(n))) return -1;
        plan = c->encplan.p;
    }
    {
        Timed t(c, "svb_encode");
        if (segmented)
            HIPCHK(c, launch_svb_encode_seg(a, (int)o->integer_size, o->perform_delta_zig_zag, 0, false, seg.first, seg.max_segs, seg.val, seg.off, s),
                   "svb_encode (segmented) launch");
        else
            HIPCHK(c, launch_svb_encode(a, (int)o->integer_size, o->perform_delta_zig_zag, 0, false, half_codec(o), (matcher && c->long_repeats != 3) ? deep_d : nullptr,
                                        pre_filled ? plan : nullptr, s),
                   "svb_encode launch");
        if (matcher && c->long_repeats == 3) (void)hipMemsetAsync(deep_d, 0, 4ull * n, s);
    }
    if (pre && pre->after_first) HIPCHK(c, hipEventRecord(pre->after_first, s), "event record");
    ReadBatch z = rb;
    z.src = (const uint8_t*)scratch_base;
    z.src_off = svb_off;
    z.src_size = svb_size;
    if (segmented) {  // few, large reads: one wavefront per span of a stream, then compaction
        const uint32_t max_spans = zstd_span_max_spans(scratch_need, n);
        if (!max_spans) {
            set_error(c, "batch too large for the span path");
            return -1;
        }
        const uint64_t tmp_bytes = zstd_span_tmp_bytes(scratch_need, n, max_spans);
        if (!ensure(c, c->spantmp, tmp_bytes)) return -1;
        // Shared tables (zstd_encode.hip, SpanRegion): the data bytes of a read get ONE table, built beside the control-byte spans, and are
        // packed in spans of 8 KB by a launch of their own.  For calls that are a matter of latency (zstd_span_shared_bytes) and for 8 / 16-bit
        // elements, whose control bytes are mostly zero and cheap to code: 0.176 -> 0.140 ms per call for a 100 k-sample read.  The
        // control-byte spans of 32-bit elements take as long as the table construction itself (a 40 MB buffer: 100 us in the first launch,
        // + 15 us counting + 34 us packing against 125 us for everything in one launch) -- they keep a table per span.
        // (canonical mode: whether the data bytes share a table must not follow from the call's size -- every span its own table)
        const uint32_t shspan = (c->shared_tables && !c->canonical && (o->integer_size == 1 || o->integer_size == 2)) ? zstd_span_shared_bytes(scratch_need) : 0u;
        const bool shared = shspan != 0;
        if (!ensure(c, c->spanmeta, (size_t)max_spans * (zstd_span_desc_bytes() + 12) + ((size_t)n + 2) * 4 + 256 + (shared ? zstd_span_region_bytes(n) + 64 : 0))) return -1;
        MetaCarver sm(c->spanmeta.p);
        uint8_t* regions = shared ? sm.take<uint8_t>(zstd_span_region_bytes(n)) : nullptr;
        uint8_t* desc = sm.take<uint8_t>((size_t)max_spans * zstd_span_desc_bytes());
        uint32_t* span_first = sm.take<uint32_t>((size_t)n + 1);
        uint32_t* span_count = sm.take<uint32_t>(1);
        uint32_t* span_size = sm.take<uint32_t>(max_spans);
        uint32_t* span_trail = sm.take<uint32_t>(max_spans);
        uint32_t* span_dst = sm.take<uint32_t>(max_spans);
        z.gate = gate;
        if (c->zero_run_sequences && c->long_repeats && o->integer_size != 0) {
            // reads that repeat at one distance go to the one-wavefront matcher after all, whatever their length: a read that
            // cycles a template (the reference's own perf generator: vbz/perf/test_data_generator.h:61-67) is 15-30 x smaller as
            // matches than as spans of Huffman blocks, which is worth one wavefront's time on it (libzstd, the reference's
            // coder, finds those matches at every level and every length: vbz/vbz.cpp:194-207).  The probe looks at the head of
            // a long read's data bytes only (PROBE_WINDOW in zstd_encode.hip): two short launches for a read that has no period.
            uint32_t* gate2 = mc.take<uint32_t>(n);
            Timed t(c, "zstd_encode_matcher");
            HIPCHK(c, launch_zstd_encode_matcher(z, bt->src_size, o->integer_size, hdr, svb_cap, c->seqtab.p, c->trailers, 0xFFFFFFFFu,
                                                 deep_d, gate, gate2, s),
                   "zstd_encode (matcher) launch");
            z.gate = gate2;
        }
        Timed t(c, "zstd_encode");
        HIPCHK(c, launch_zstd_encode_spans(z, bt->src_size, o->integer_size, hdr, c->zero_run_sequences ? svb_cap : nullptr,
                                           c->zero_run_sequences ? c->seqtab.p : nullptr, desc, span_first, span_count, max_spans,
                                           (uint8_t*)c->spantmp.p, tmp_bytes, span_size, span_trail, span_dst, c->trailers, regions, shspan, s),
               "zstd_encode (spans) launch");
        return 0;
    }
    {
        Timed t(c, "zstd_encode");
        // (phase timing 2: the planning launch's counters, 3: the packing launch's; both under load, the other launches as they are)
        HIPCHK(c, launch_zstd_encode(z, bt->src_size, o->integer_size, nullptr, hdr, c->phase_timing == 3 ? nullptr : dbg,
                                     c->zero_run_sequences ? svb_cap : nullptr, c->zero_run_sequences ? c->seqtab.p : nullptr, c->trailers,
                                     (matcher && !dbg && c->long_repeats != 2) ? deep_d : nullptr, plan, staged, pre_filled, c->phase_timing == 3 ? dbg : nullptr, s),
               "zstd_encode launch");
    }
    dbg_end(c, n,
            c->phase_timing == 3   ? "zstd_pack: setup region lookups+scan bits quads ends+headers sequences trailer"
            : c->phase_timing == 2 ? "zstd_encode planning launch: setup hist plan - store+sequences - | plan: sort merge lengths codes weights tree"
                                   : "zstd_encode: setup hist plan size hdr encode",
            dbg);
    return 0;
}

constexpr uint32_t REF_MIN_READS = 2560;   // calls of fewer reads: the one-wavefront decoder walks reference-written chains itself

// The frames of a launch group on the one-workgroup path (not the large-read path): frames of this library's shape on the batched
// decoder, the sequence chains of frames the reference wrote walked one lane per frame, everything else -- and every error verdict --
// from the one-wavefront decoder.  content_bytes bounds the frames' content in all.
int zstd_frames(vbz_gpu_ctx* c, const ReadBatch& z, uint32_t toosmall_code, uint64_t content_bytes, unsigned long long* dbg)
{
    const uint32_t n = z.n_reads;
    hipStream_t s = c->stream;
    Timed t(c, "zstd_decode");
    c->last_frames = 0;
    if (!c->fast_decode) {
        HIPCHK(c, launch_zstd_decode(z, toosmall_code, dbg, c->seqdtab.p, s), "zstd_decode launch");
        return 0;
    }
    if (!ensure(c, c->fastmeta, zstd_fast_meta_bytes(n))) return -1;
    // Frames the reference wrote: their chains are walked ahead of the one-wavefront decoder when the call is large enough for that to
    // pay (the walk is a launch of its own whose duration is one chain's latency, ~0.7 ms: measured break-even 2 500 such frames).
    // Records of walked chains: 16 bytes per sequence, claimed from one workspace of the call as the lanes go (an atomic counter; a frame
    // that finds no room is decoded as before).  libzstd on nanopore signal writes ~1 100 sequences per 126 KB of content -- records of
    // 14 % of the content's size --: the workspace is a quarter of the declared content + 1 MiB, at most 1 GiB (the declared size
    // is the caller's word, and the buffer stays with the context: ADVICE round 4; 16 384 reads of 100 k samples need 290 MB).
    // (a workload that has shown no foreign frame does not launch the walk at all: the -- empty -- launch sits on a side stream, gets its
    // few wavefronts late when the device is full, and the launches behind the join wait for it; should foreign frames appear after all,
    // the one-wavefront decoder walks their chains itself for one call and the next call knows)
    const bool walk = (c->ref_chains >= 2 || (c->ref_chains == 1 && n >= REF_MIN_READS)) && (c->foreign_state != 0 || c->ref_chains >= 2);
    const uint64_t recs_bytes = walk ? std::min<uint64_t>(((content_bytes >> 2) + (1ull << 20)) & ~15ull, 1ull << 30) : 0;
    if (walk && (!ensure(c, c->refpre, zstd_ref_pre_bytes(n)) || !ensure(c, c->reftab, zstd_ref_table_bytes(n)) || !ensure(c, c->refrecs, recs_bytes)))
        return -1;
    if (walk && !dbg && !ensure_side(c)) return -1;
    const bool lits = walk && zstd_ref_literals_enabled();
    if (lits && !ensure(c, c->reflits, zstd_ref_lit_meta_bytes(n))) return -1;
    // (blocks per frame that get a record and a workgroup of the pieces' kernel: a call of short reads has one block a frame -- three of four
    // workgroups of a grid for four would start only to end, 0.3 ms per 16 384 frames --, a call of long reads two or three of 128 KB and a rest)
    const uint64_t avg_content = content_bytes / n;
    // (content_bytes is the caller's bound: the decoded bytes of the call, 1.6 x the frames' content where an svb stage follows)
    const uint32_t lit_units = avg_content <= (96u << 10) ? 1u : (uint32_t)std::min<uint64_t>(avg_content / (256u << 10) + 2u, zstd_ref_lit_units());
    c->last_lit_units = lits ? lit_units : 0;
    HIPCHK(c, launch_zstd_decode_fast(z, toosmall_code, c->seqdtab.p, c->fastmeta.p, walk ? c->refpre.p : nullptr, c->reftab.p, c->refrecs.p,
                                      recs_bytes / 16, lits ? c->reflits.p : nullptr, lit_units, dbg, c->side, s),
           "zstd_decode (batched) launch");
    c->last_walked = walk;
    c->last_frames = dbg ? 0 : n;
    if (c->split_min != 0 && n >= c->split_min / 2 && !dbg) {   // (the context of a call that may be split: see foreign_host)
        if (!c->foreign_host && (hipHostMalloc((void**)&c->foreign_host, 16, hipHostMallocDefault) != hipSuccess ||
                                 hipEventCreateWithFlags(&c->ev_foreign, hipEventDisableTiming) != hipSuccess)) {
            (void)hipGetLastError();
            if (c->foreign_host) (void)hipHostFree(c->foreign_host);
            c->foreign_host = nullptr;
        }
        if (c->foreign_host && !c->foreign_pending && ensure(c, c->foreign_dev, 16)) {
            uint32_t* d = (uint32_t*)c->foreign_dev.p;
            if (launch_count_nonzero(zstd_fast_redo(c->fastmeta.p, n), n, d, s) == hipSuccess &&
                hipMemcpyAsync(c->foreign_host, d, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipEventRecord(c->ev_foreign, s) == hipSuccess) {
                c->foreign_host[1] = n;
                c->foreign_pending = true;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    return 0;
}

// One launch group of a decompress call: the reads of rb_in (dst_cap = the exact decoded byte counts; those whose gate is closed
// left alone), on the one-workgroup path or on the large-read path.  dst_bytes: extent of the decoded bytes of the group.
int decompress_group(vbz_gpu_ctx* c, const ReadBatch& rb_in, uint64_t dst_bytes, const CompressionOptions* o, bool segmented,
                     const Preplanned* pre = nullptr)
{
    const uint32_t n = rb_in.n_reads;
    if (n == 0) return 0;
    hipStream_t s = c->stream;
    if (!ensure(c, c->gmeta, (size_t)n * 24 + 512)) return -1;
    MetaCarver mc(c->gmeta.p);
    uint64_t* svb_off = mc.take<uint64_t>(n);
    uint32_t* gate = mc.take<uint32_t>(n);
    uint32_t* svb_cap = mc.take<uint32_t>(n);
    uint32_t* svb_size = mc.take<uint32_t>(n);
    ReadBatch rb = rb_in;
    if (o->integer_size == 0 && o->zstd_compression_level == 0) {
        Timed t(c, "copy_bytes");
        HIPCHK(c, launch_copy_bytes(rb, 0, s), "copy launch");
        return 0;
    }
    SegTables seg;
    // (both stages: the scratch slots are planned with the segments when they can be -- two launches less for a call of few reads)
    const bool both_stages = o->integer_size != 0 && o->zstd_compression_level != 0;
    uint32_t num = 1, den = 1;
    size_t scratch_need = 0;
    if (both_stages) {
        svb_factor(o->integer_size, false, &num, &den);  // any code length may appear in a foreign stream
        scratch_need = (size_t)(((unsigned __int128)dst_bytes * num + den - 1) / den) + (size_t)n * 96 + 256;
        if (!pre && !ensure(c, c->scratch, scratch_need)) return -1;
    }
    const ScratchPlan splan = { num, den, c->scratch.cap, svb_off, svb_cap, gate };
    bool scratch_planned = false;
    void* const scratch_base = pre ? pre->scratch : c->scratch.p;
    if (pre) {   // (both stages, one-workgroup path: split_batch) -- slots and gates planned once for the whole batch
        svb_off = pre->svb_off;
        svb_cap = pre->svb_cap;
        gate = pre->gate;
        rb.gate = gate;
        scratch_planned = true;
    }
    if (segmented && plan_segments(c, n, rb.dst_cap, rb.gate, dst_bytes, svb_seg_unit_bytes((int)o->integer_size), &seg, both_stages ? &splan : nullptr,
                                   &scratch_planned) != 0)
        return -1;
    const bool gate_in = !pre && (segmented || rb.gate != nullptr);
    if (gate_in) {   // (the caller's gate; E_OOM for a read whose segments do not fit the tables)
        if (!scratch_planned) HIPCHK(c, hipMemcpyAsync(gate, segmented ? seg.gate : rb.gate, 4ull * n, hipMemcpyDeviceToDevice, s), "gate copy");
        rb.gate = gate;
    }
    if (o->zstd_compression_level == 0) {
        Timed t(c, "svb_decode");
        if (segmented)
            HIPCHK(c, launch_svb_decode_seg(rb, (int)o->integer_size, o->perform_delta_zig_zag, seg.first, seg.max_segs, seg.val, seg.off, seg.run, s),
                   "svb_decode (segmented) launch");
        else
            HIPCHK(c, launch_svb_decode(rb, (int)o->integer_size, o->perform_delta_zig_zag, half_codec(o), s), "svb_decode launch");
        return 0;
    }
    if (o->integer_size == 0) {  // vbz.cpp:259-262: content larger than the destination -> DESTINATION_SIZE
        if (segmented) {
            Timed t(c, "zstd_decode");
            HIPCHK(c, launch_zstd_decode(rb, E_DESTINATION_SIZE, nullptr, c->seqdtab.p, s), "zstd_decode launch");
            return 0;
        }
        return zstd_frames(c, rb, E_DESTINATION_SIZE, dst_bytes, nullptr);
    }
    // entropy stage into scratch (sized for the largest svb stream the expected output can have),
    // then svb decode into dst (vbz.cpp:234-299)
    if (!scratch_planned) {
        Timed t(c, "plan_scratch");
        HIPCHK(c, launch_plan_scratch(n, rb.dst_cap, num, den, c->scratch.cap, svb_off, svb_cap, gate, gate_in, s), "plan launch");
    }
    ReadBatch z = rb;
    z.dst = (uint8_t*)scratch_base;
    z.dst_off = svb_off;
    z.dst_cap = svb_cap;
    z.result = svb_size;
    z.gate = gate;
    unsigned long long* dbg = segmented ? nullptr : dbg_begin(c, n);
    // the hot path -- int16 zig-zag samples, one wavefront per frame: optionally (VBZ_HIP_FUSE_SVB=1) the wavefront decodes the
    // svb stream it has just written while it is still in the caches, straight into the destination, and there is no
    // svb_decode launch (measured slower than the separate launch: profiles/r03_fused_svb_decode.md)
#ifdef VBZ_EXPERIMENTS
    if (!segmented && !dbg && c->fuse_svb && o->integer_size == 2 && o->perform_delta_zig_zag) {
        z.result = rb.result;
        Timed t(c, "zstd_decode");  // (zstd_decode_kernel<false, true>: the frame and its svb stream)
        HIPCHK(c, launch_zstd_decode_svb_i16zz(z, E_STREAM, c->seqdtab.p, rb.dst, rb.dst_off, rb.dst_cap, s), "zstd_decode + svb_decode launch");
        return 0;
    }
#endif
    if (segmented) {  // few, large reads: frames with a span index are decoded one span per wavefront
        const uint32_t max_spans = zstd_dspan_max_spans(scratch_need, n);
        if (!max_spans) {
            set_error(c, "batch too large for the span path");
            return -1;
        }
        if (!ensure(c, c->spanmeta, (size_t)max_spans * (zstd_dspan_desc_bytes() + 16) + ((size_t)n + 2) * 8 + 256)) return -1;
        MetaCarver sm(c->spanmeta.p);
        uint8_t* desc = sm.take<uint8_t>((size_t)max_spans * zstd_dspan_desc_bytes());
        uint32_t* dspan_first = sm.take<uint32_t>((size_t)n + 1);
        uint32_t* dspan_count = sm.take<uint32_t>(1);
        uint32_t* dspan_status = sm.take<uint32_t>((size_t)max_spans * 4);
        uint32_t* redo = sm.take<uint32_t>(n);
        Timed t(c, "zstd_decode");
        HIPCHK(c, launch_zstd_decode_spans(z, E_STREAM, c->seqdtab.p, desc, dspan_first, dspan_count, max_spans, dspan_status, redo, s),
               "zstd_decode (spans) launch");
        c->last_frames = 0;
        c->last_span_frames = n;
        c->last_span_redo = redo;
    } else {
        c->last_span_frames = 0;
        // a frame whose content cannot be a valid svb stream of the expected size: the reference would
        // decode it and then fail in the svb stage with a stream error
        if (zstd_frames(c, z, E_STREAM, dst_bytes, dbg) != 0) return -1;
    }
    if (pre && pre->after_first) HIPCHK(c, hipEventRecord(pre->after_first, s), "event record");
    dbg_end(c, n, "zstd_decode: parse flush seqtables chain place huftable header queue | general sequences: flush tables records literals matches", dbg);
    ReadBatch d = rb;
    d.src = (const uint8_t*)scratch_base;
    d.src_off = svb_off;
    d.src_size = svb_size;
    d.gate = gate;
    {
        Timed t(c, "svb_decode");
        if (segmented)
            HIPCHK(c, launch_svb_decode_seg(d, (int)o->integer_size, o->perform_delta_zig_zag, seg.first, seg.max_segs, seg.val, seg.off, seg.run, s),
                   "svb_decode (segmented) launch");
        else
            HIPCHK(c, launch_svb_decode(d, (int)o->integer_size, o->perform_delta_zig_zag, half_codec(o), s), "svb_decode launch");
    }
    return 0;
}

// ---- per-read routing ----------------------------------------------------------------------------------------------------
// The one-workgroup / one-wavefront kernels fill the GPU with thousands of reads, but ONE very long read among them (ultra-long
// nanopore reads exist: millions of samples) would run on one wavefront of 1024 for tens of milliseconds -- the reference
// treats every buffer alike (vbz/vbz.cpp:116-208), a batch must too.  The host does not know the sizes (they live on the
// device), so every call whose arena could hold a long read routes on the device: route_reads picks the reads of
// ROUTE_MIN_BYTES and more (at most ROUTE_MAX_READS of them, ROUTE_MAX_BYTES in all: more than that and the batch shape rule
// above has already sent the whole batch down the large-read path) into a compact second group that takes the large-read
// path with small grids; in the first group their gate says GATE_SKIP.  Without long reads the second group's launches
// find nothing to do (a few empty grids per call).
constexpr uint32_t ROUTE_MIN_BYTES = (uint32_t)SEGMENTED_MIN_AVG, ROUTE_MAX_READS = 16;
constexpr uint64_t ROUTE_MAX_BYTES = 64ull << 20;

struct Routed
{
    uint32_t* gate_small = nullptr;   // [n]
    ReadBatch large;                  // ROUTE_MAX_READS compact entries (gate: GATE_SKIP behind the routed ones)
    uint32_t* large_orig = nullptr;   // compress: raw sizes of the routed reads (= large.src_size); decompress: unused
    uint32_t* map = nullptr;          // [ROUTE_MAX_READS] index of each routed read in the batch
    uint32_t* count = nullptr;        // [1]
};

bool routing_applies(const vbz_gpu_ctx* c, const CompressionOptions* o, uint64_t raw_arena_bytes, uint32_t n)
{
    return c->routing && c->segmented < 0 && n > 1 && o->integer_size != 0 && !half_codec(o) && raw_arena_bytes >= ROUTE_MIN_BYTES;
}

// the second launch group's context (the large-read path beside the first group, on a stream of its own)
int ensure_large(vbz_gpu_ctx* c)
{
    if (!c->large) {
        c->large = vbz_gpu_create(c->device, nullptr);
        if (!c->large || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
            set_error(c, "could not create the context of the routed reads");
            return -1;
        }
        c->large->routing = 0;
        c->large->split_min = 0;
    }
    c->large->trailers = c->trailers;
    c->large->zero_run_sequences = c->zero_run_sequences;
    c->large->long_repeats = c->long_repeats;
    c->large->shared_tables = c->shared_tables;
    c->large->canonical = c->canonical;
    c->large->segmented = 1;
    return 0;
}

int route(vbz_gpu_ctx* c, const ReadBatch& rb, const uint32_t* raw_size, Routed* r)
{
    const uint32_t n = rb.n_reads;
    if (ensure_large(c) != 0) return -1;
    if (!ensure(c, c->route, (size_t)n * 4 + (size_t)ROUTE_MAX_READS * 48 + route_cand_words() * 4 + 512)) return -1;
    MetaCarver mc(c->route.p);
    r->gate_small = mc.take<uint32_t>(n);
    uint64_t* l_src_off = mc.take<uint64_t>(ROUTE_MAX_READS);
    uint64_t* l_dst_off = mc.take<uint64_t>(ROUTE_MAX_READS);
    uint32_t* l_src_size = mc.take<uint32_t>(ROUTE_MAX_READS);
    uint32_t* l_dst_cap = mc.take<uint32_t>(ROUTE_MAX_READS);
    uint32_t* l_gate = mc.take<uint32_t>(ROUTE_MAX_READS);
    uint32_t* l_result = mc.take<uint32_t>(ROUTE_MAX_READS);
    r->map = mc.take<uint32_t>(ROUTE_MAX_READS);
    r->count = mc.take<uint32_t>(4);
    uint32_t* cand = mc.take<uint32_t>(route_cand_words());
    r->large = rb;
    r->large.n_reads = ROUTE_MAX_READS;
    r->large.src_off = l_src_off;
    r->large.src_size = l_src_size;
    r->large.dst_off = l_dst_off;
    r->large.dst_cap = l_dst_cap;
    r->large.gate = l_gate;
    r->large.result = l_result;
    Timed t(c, "route");
    HIPCHK(c, launch_route_reads(rb, raw_size, ROUTE_MIN_BYTES, ROUTE_MAX_READS, ROUTE_MAX_BYTES, r->gate_small, l_src_off, l_src_size, l_dst_off, l_dst_cap,
                                 l_gate, r->map, r->count, cand, c->stream),
           "route launch");
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream), "event record");
    HIPCHK(c, hipStreamWaitEvent(c->large->stream, c->ev_fork, 0), "stream wait");
    return 0;
}

// the second group is done before anything enqueued behind the call runs
int route_join(vbz_gpu_ctx* c, const Routed& r, uint32_t* result)
{
    HIPCHK(c, launch_route_results(r.large.result, r.map, r.count, ROUTE_MAX_READS, result, c->large->stream), "route results launch");
    HIPCHK(c, hipEventRecord(c->ev_join, c->large->stream), "event record");
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0), "stream wait");
    return 0;
}


// ---- two halves of a batch in flight -------------------------------------------------------------------------------------
// One launch sequence per batch leaves the device to ONE kind of kernel at a time: instruction-bound launches (svb, packing) and
// latency-bound ones (planning, the stream decoder) each run alone, and every launch ends in a tail of partly idle CUs.  Two contexts
// coding alternate batches on streams of their own measured 572 GB/s where one reaches 545 (profiles/r05r_two_in_flight.txt).  The
// same inside ONE call: descriptors are validated, the sized headers parsed, long reads routed and the scratch slots of ALL reads
// planned once on the context's stream; then the reads are cut at the midpoint of the table, the lower half goes on on this
// stream, the upper half on a child context's stream (its own per-read tables; the scratch arena and its plan are shared -- every
// per-read table is an array, so a half is a pointer offset), and the child's stream is joined before the call returns.  A read's
// bytes do not depend on its neighbours, so the output is what the unsplit call writes (tests/test_gpu_split.py: sha256).
bool split_applies(const vbz_gpu_ctx* c, const CompressionOptions* o, uint32_t n)
{
    return c->split_min != 0 && n >= c->split_min && o->integer_size != 0 && o->zstd_compression_level != 0 && !half_codec(o) && !c->phase_timing && !c->trace;
}

struct Split
{
    uint32_t h = 0;              // reads [0, h) stay on the context, [h, n) go to the child
    Preplanned lo, hi;
};

// raw_size: the reads' raw (decoded) byte counts; num / den as the group would have chosen them
int split_plan(vbz_gpu_ctx* c, const ReadBatch& rb, const uint32_t* raw_size, uint64_t raw_bytes, uint32_t num, uint32_t den, Split* sp)
{
    const uint32_t n = rb.n_reads;
    if (!c->half) {
        c->half = vbz_gpu_create(c->device, nullptr);
        if (!c->half || hipEventCreateWithFlags(&c->ev_hfork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_hjoin, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_hstagger, hipEventDisableTiming) != hipSuccess) {
            set_error(c, "could not create the context of a batch's upper half");
            return -1;
        }
        c->half->routing = 0;
        c->half->split_min = 0;
        c->half->segmented = 0;
    }
    vbz_gpu_ctx* k = c->half;
    k->trailers = c->trailers;
    k->zero_run_sequences = c->zero_run_sequences;
    k->long_repeats = c->long_repeats;
    k->staged_encode = c->staged_encode;
    k->fast_decode = c->fast_decode;
    k->ref_chains = c->ref_chains;
    k->fuse_svb = c->fuse_svb;
    k->profiling = c->profiling;
    k->foreign_state = c->foreign_state;
    const size_t scratch_need = (size_t)(((unsigned __int128)raw_bytes * num + den - 1) / den) + (size_t)n * 96 + 256;
    if (!ensure(c, c->scratch, scratch_need) || !ensure(c, c->splitmeta, (size_t)n * 16 + 256)) return -1;
    MetaCarver mc(c->splitmeta.p);
    uint64_t* svb_off = mc.take<uint64_t>(n);
    uint32_t* svb_cap = mc.take<uint32_t>(n);
    uint32_t* gate = mc.take<uint32_t>(n);
    hipStream_t s = c->stream;
    if (rb.gate) HIPCHK(c, hipMemcpyAsync(gate, rb.gate, 4ull * n, hipMemcpyDeviceToDevice, s), "gate copy");
    {
        Timed t(c, "plan_scratch");
        HIPCHK(c, launch_plan_scratch(n, raw_size, num, den, c->scratch.cap, svb_off, svb_cap, gate, rb.gate != nullptr, s), "plan launch");
    }
    sp->h = n / 2;
    sp->lo.scratch = sp->hi.scratch = c->scratch.p;
    sp->lo.svb_off = svb_off;
    sp->lo.svb_cap = svb_cap;
    sp->lo.gate = gate;
    sp->hi.svb_off = svb_off + sp->h;
    sp->hi.svb_cap = svb_cap + sp->h;
    sp->hi.gate = gate + sp->h;
    if (c->split_stagger) sp->lo.after_first = c->ev_hstagger;
    HIPCHK(c, hipEventRecord(c->ev_hfork, s), "event record");
    HIPCHK(c, hipStreamWaitEvent(k->stream, c->ev_hfork, 0), "stream wait");
    return 0;
}

ReadBatch upper_half(const ReadBatch& rb, uint32_t h)
{
    ReadBatch u = rb;
    u.n_reads = rb.n_reads - h;
    u.src_off += h;
    u.src_size += h;
    u.dst_off += h;
    u.dst_cap += h;
    u.result += h;
    if (u.gate) u.gate += h;
    return u;
}

// the upper half's stream is joined whatever happened in between (it may still be writing the caller's arenas)
int split_join(vbz_gpu_ctx* c)
{
    HIPCHK(c, hipEventRecord(c->ev_hjoin, c->half->stream), "event record");
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_hjoin, 0), "stream wait");
    if (c->error.empty() && !c->half->error.empty()) c->error = c->half->error;
    return 0;
}

int compress_split(vbz_gpu_ctx* c, const ReadBatch& rb, uint64_t src_bytes, const CompressionOptions* o, int sized)
{
    uint32_t num, den;
    svb_factor(o->integer_size, o->perform_delta_zig_zag, &num, &den);
    Split sp;
    if (split_plan(c, rb, rb.src_size, src_bytes, num, den, &sp) != 0) return -1;
    ReadBatch lo = rb;
    lo.n_reads = sp.h;
    int rc = compress_group(c, lo, src_bytes, o, sized, false, &sp.lo);
    if (rc == 0 && c->split_stagger && hipStreamWaitEvent(c->half->stream, c->ev_hstagger, 0) != hipSuccess) rc = -1;
    if (rc == 0) rc = compress_group(c->half, upper_half(rb, sp.h), src_bytes, o, sized, false, &sp.hi);
    if (split_join(c) != 0) rc = -1;
    return rc;
}

int decompress_split(vbz_gpu_ctx* c, const ReadBatch& rb, uint64_t dst_bytes, const CompressionOptions* o)
{
    uint32_t num, den;
    svb_factor(o->integer_size, false, &num, &den);
    Split sp;
    if (split_plan(c, rb, rb.dst_cap, dst_bytes, num, den, &sp) != 0) return -1;
    ReadBatch lo = rb;
    lo.n_reads = sp.h;
    int rc = decompress_group(c, lo, dst_bytes, o, false, &sp.lo);
    if (rc == 0 && c->split_stagger && hipStreamWaitEvent(c->half->stream, c->ev_hstagger, 0) != hipSuccess) rc = -1;
    if (rc == 0) rc = decompress_group(c->half, upper_half(rb, sp.h), dst_bytes, o, false, &sp.hi);
    if (split_join(c) != 0) rc = -1;
    c->last_split = rc == 0;
    return rc;
}


// ---- canonical encoding ----------------------------------------------------------------------------------------------------------------
// The reference is a pure function of (input, options, libzstd version) per buffer (vbz/vbz.cpp:116-208).  This library's frames are
// standard zstd whatever path wrote them, but WHICH path a read takes follows, by default, from the shape of the call it arrives in
// (a batch of few large reads, a handful of reads, per-read routing with its limits, shared tables while the call is small): the same
// read can come out as different -- equally valid -- bytes from the HDF5 filter, the bulk re-packer and a batch of thousands.  In
// canonical mode the rule looks at the READ alone: a read of CANON_LARGE_BYTES raw bytes and more is coded as spans (every span its own
// table), every other read by one wavefront -- one kernel classifies the reads by their sizes on the device, the count of large ones
// comes back to the host (the one synchronisation of the call), and the two kinds run as two launch groups beside each other like
// per-read routing's.  Destination slots are taken to have the reference's capacity contract (vbz_max_compressed_size): the long-repeat
// coder's workspace is what a slot has above the worst-case frame.  Decoding needs no such mode.  tests/test_gpu_canonical.py holds
// vbz_compress, batches of 1 and of 4096, the HDF5 filter and the bulk re-packer to one sha256 per read.
constexpr uint32_t CANON_LARGE_BYTES = (uint32_t)SEGMENTED_MIN_AVG;

int compress_canonical(vbz_gpu_ctx* c, const ReadBatch& rb, uint64_t src_bytes, const CompressionOptions* o, int sized)
{
    const uint32_t n = rb.n_reads;
    hipStream_t s = c->stream;
    if (!c->canon_host && hipHostMalloc((void**)&c->canon_host, 16, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error(c, "pinned word for the canonical classification");
        return -1;
    }
    if (!ensure(c, c->route, (size_t)n * 8 + 64)) return -1;
    MetaCarver mc(c->route.p);
    uint32_t* counts = mc.take<uint32_t>(4);
    uint32_t* gate_small = mc.take<uint32_t>(n);
    uint32_t* gate_large = mc.take<uint32_t>(n);
    HIPCHK(c, launch_canon_classify(n, rb.src_size, rb.gate, CANON_LARGE_BYTES, gate_small, gate_large, counts, s), "classify launch");
    HIPCHK(c, hipMemcpyAsync(c->canon_host, counts, 8, hipMemcpyDeviceToHost, s), "count copy");
    HIPCHK(c, hipStreamSynchronize(s), "stream synchronize");
    const uint32_t nbig = c->canon_host[0];
    const bool split = split_applies(c, o, n);
    if (nbig == 0) return split ? compress_split(c, rb, src_bytes, o, sized) : compress_group(c, rb, src_bytes, o, sized, false);
    if (nbig == n) return compress_group(c, rb, src_bytes, o, sized, true);
    if (ensure_large(c) != 0) return -1;
    HIPCHK(c, hipEventRecord(c->ev_fork, s), "event record");
    HIPCHK(c, hipStreamWaitEvent(c->large->stream, c->ev_fork, 0), "stream wait");
    ReadBatch small = rb, large = rb;
    small.gate = gate_small;
    large.gate = gate_large;
    int rc = 0;   // (as in compress_batch_impl: the second stream is joined whatever happens)
    if (compress_group(c->large, large, src_bytes, o, sized, true) != 0) rc = -1;
    if (rc == 0 && (split ? compress_split(c, small, src_bytes, o, sized) : compress_group(c, small, src_bytes, o, sized, false)) != 0) rc = -1;
    if (rc != 0 && c->error.empty() && !c->large->error.empty()) c->error = c->large->error;
    if (hipEventRecord(c->ev_join, c->large->stream) != hipSuccess || hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess) rc = -1;
    return rc;
}

// The caller's descriptor table is untrusted (vbz_gpu.h): one thread per read checks its slots against the declared arenas before any
// other kernel forms an address from them; the verdicts are the gate every launch group of the call starts from.
int validate_descriptors(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, ReadBatch* rb)
{
    if (!ensure(c, c->vgate, (size_t)bt->n_reads * 4 + 64)) return -1;
    uint32_t* g = reinterpret_cast<uint32_t*>(c->vgate.p);
    HIPCHK(c, launch_validate_batch(bt->n_reads, bt->src_off, bt->src_size, bt->src_bytes, bt->dst_off, bt->dst_cap, bt->dst_bytes, g, c->stream), "validate launch");
    rb->gate = g;
    return 0;
}

// The arena extents a caller declares size the scratch: a value no device could hold is refused before any arithmetic is done with it
// (scratch bounds are a few times the extent: 2^46 bytes keeps every product inside 64 bits).
constexpr uint64_t EXTENT_MAX = 1ull << 46;
bool plausible_extents(vbz_gpu_ctx* c, const vbz_gpu_batch* bt)
{
    if (bt->n_reads != 0 && (!bt->src_off || !bt->src_size || !bt->dst_off || !bt->dst_cap || !bt->result || (!bt->src && bt->src_bytes != 0) ||
                             (!bt->dst && bt->dst_bytes != 0))) {
        set_error(c, "a table or arena pointer of the batch is NULL");
        return false;
    }
    if (bt->src_bytes <= EXTENT_MAX && bt->dst_bytes <= EXTENT_MAX) return true;
    set_error(c, "declared arena extents are not plausible (src_bytes %llu, dst_bytes %llu)", (unsigned long long)bt->src_bytes, (unsigned long long)bt->dst_bytes);
    return false;
}

// own_descriptors: the table is the library's own (the single-buffer API: one read, slots it has just allocated) -- nothing to validate
int compress_batch_impl(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, const CompressionOptions* o, int sized, bool own_descriptors = false)
{
    const uint32_t n = bt->n_reads;
    c->last_span_frames = 0;   // (vbz_gpu_decode_span_paths: its redo[] lives in spanmeta, which a span-mode compress call carves anew)
    if (n == 0) return 0;
    ReadBatch rb = to_rb(bt);
    if (!own_descriptors && validate_descriptors(c, bt, &rb) != 0) return -1;
    if (c->canonical && c->segmented < 0 && o->integer_size != 0 && o->zstd_compression_level != 0 && !half_codec(o))
        return compress_canonical(c, rb, bt->src_bytes, o, sized);
    const bool by_shape = o->integer_size != 0 && !half_codec(o) && use_segments(c, bt->src_bytes, n, false);
    const bool split = !by_shape && split_applies(c, o, n);
    if (by_shape || !routing_applies(c, o, bt->src_bytes, n))
        return split ? compress_split(c, rb, bt->src_bytes, o, sized) : compress_group(c, rb, bt->src_bytes, o, sized, by_shape);
    Routed r;
    if (route(c, rb, bt->src_size, &r) != 0) return -1;
    ReadBatch small = rb;
    small.gate = r.gate_small;
    // (the second group first: its launches are short.)  Whatever fails from here on, the second stream is joined before the call
    // returns -- it may still be writing the caller's arenas -- and its error message becomes the context's
    int rc = 0;
    if (c->routing != 2 && compress_group(c->large, r.large, ROUTE_MAX_BYTES, o, sized, true) != 0) rc = -1;
    if (rc == 0 && (split ? compress_split(c, small, bt->src_bytes, o, sized) : compress_group(c, small, bt->src_bytes, o, sized, false)) != 0) rc = -1;
    if (rc != 0 && c->error.empty() && !c->large->error.empty()) c->error = c->large->error;
    if (route_join(c, r, bt->result) != 0) rc = -1;
    return rc;
}

int decompress_batch_impl(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, const CompressionOptions* o, int sized, bool own_descriptors = false)
{
    const uint32_t n = bt->n_reads;
    c->last_frames = 0;
    c->last_span_frames = 0;
    if (n == 0) return 0;
    hipStream_t s = c->stream;
    ReadBatch rb = to_rb(bt);
    if (!own_descriptors && validate_descriptors(c, bt, &rb) != 0) return -1;
    if (sized) {  // vbz.cpp:332-366: strip the header, the original size becomes the exact destination size
        if (!ensure(c, c->meta, (size_t)n * 24 + 512)) return -1;
        MetaCarver mc(c->meta.p);
        uint64_t* pay_off = mc.take<uint64_t>(n);
        uint32_t* pay_size = mc.take<uint32_t>(n);
        uint32_t* orig_size = mc.take<uint32_t>(n);
        uint32_t* gate = mc.take<uint32_t>(n);
        Timed t(c, "parse_sized");
        HIPCHK(c, launch_parse_sized(n, rb.src, bt->src_off, bt->src_size, bt->dst_cap, rb.gate, pay_off, pay_size, orig_size, gate, s),
               "parse_sized launch");
        rb.src_off = pay_off;
        rb.src_size = pay_size;
        rb.dst_cap = orig_size;
        rb.gate = gate;
    }
    const bool by_shape = o->integer_size != 0 && !half_codec(o) && use_segments(c, bt->dst_bytes, n, true);
    if (c->foreign_pending && hipEventQuery(c->ev_foreign) == hipSuccess) {   // what the call before this one found (see foreign_host)
        c->mostly_foreign = 2ull * c->foreign_host[0] > c->foreign_host[1];
        c->foreign_state = c->foreign_host[0] != 0 ? 1 : 0;
        c->foreign_pending = false;
    } else if (c->foreign_pending) {
        (void)hipGetLastError();   // (hipErrorNotReady is not an error to leave behind)
    }
    const bool split = !by_shape && split_applies(c, o, n) && !c->mostly_foreign;
    c->last_split = false;
    if (by_shape || !routing_applies(c, o, bt->dst_bytes, n))
        return split ? decompress_split(c, rb, bt->dst_bytes, o) : decompress_group(c, rb, bt->dst_bytes, o, by_shape);
    Routed r;
    if (route(c, rb, rb.dst_cap, &r) != 0) return -1;   // by the decoded size
    ReadBatch small = rb;
    small.gate = r.gate_small;
    int rc = 0;   // (as in compress_batch_impl: the second stream is joined whatever happens)
    if (c->routing != 2 && decompress_group(c->large, r.large, ROUTE_MAX_BYTES, o, true) != 0) rc = -1;
    if (rc == 0 && (split ? decompress_split(c, small, bt->dst_bytes, o) : decompress_group(c, small, bt->dst_bytes, o, false)) != 0) rc = -1;
    if (rc != 0 && c->error.empty() && !c->large->error.empty()) c->error = c->large->error;
    if (route_join(c, r, bt->result) != 0) rc = -1;
    return rc;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// vbz_gpu.h
// ------------------------------------------------------------------------------------------------
extern "C" {

#ifdef VBZ_EXPERIMENTS
const char* vbz_gpu_version(void) { return "vbz_hip 0.6.0 gfx950 +experiments"; }
#else
const char* vbz_gpu_version(void) { return "vbz_hip 0.6.0 gfx950"; }
#endif

vbz_gpu_ctx* vbz_gpu_create(int device, void* stream)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error(nullptr, "no HIP device available (%s); this library has no CPU path", hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= count) {
        set_error(nullptr, "device %d out of range (%d devices)", device, count);
        return nullptr;
    }
    DeviceGuard dg(device);
    {
        int now = -1;
        if (hipGetDevice(&now) != hipSuccess || now != device) {
            set_error(nullptr, "hipSetDevice(%d) failed", device);
            return nullptr;
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            set_error(nullptr, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
            return nullptr;
        }
    }
    vbz_gpu_ctx* c = new vbz_gpu_ctx();
    c->device = device;
    // The knobs of the shipped library (README.md): every one selects between paths the suites run.  Known-slower variants and
    // the timed kernel instantiations are compiled into the experiments build only (-DVBZ_EXPERIMENTS: lib/libvbz_hip_x.so,
    // for tools/ and the tests that keep those variants honest); here their values do nothing.
    if (const char* e = getenv("VBZ_HIP_TRACE")) c->trace = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_ZERO_RUN_SEQUENCES")) c->zero_run_sequences = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_LONG_REPEATS")) c->long_repeats = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_FAST_DECODE")) c->fast_decode = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_REF_CHAINS")) c->ref_chains = atoi(e);
    if (const char* e = getenv("VBZ_HIP_STAGED_ENCODE")) c->staged_encode = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_SHARED_TABLES")) c->shared_tables = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_ROUTING")) c->routing = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_SEGMENTED")) c->segmented = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_TRAILERS")) c->trailers = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_CANONICAL")) c->canonical = atoi(e) != 0;
    if (const char* e = getenv("VBZ_HIP_SPLIT_MIN")) c->split_min = (uint32_t)strtoul(e, nullptr, 10);   // 0: a batch is never coded as two halves
    if (const char* e = getenv("VBZ_HIP_SPLIT_STAGGER")) c->split_stagger = atoi(e);
#ifdef VBZ_EXPERIMENTS
    if (const char* e = getenv("VBZ_HIP_PHASE_TIMING")) c->phase_timing = atoi(e);   // timed instantiations of the entropy kernels
    if (const char* e = getenv("VBZ_HIP_LONG_REPEATS")) c->long_repeats = atoi(e);   // 2: probe only, 3: second launch only
    if (const char* e = getenv("VBZ_HIP_FUSE_SVB")) c->fuse_svb = atoi(e) != 0;      // svb decode on the frame's wavefront (slower)
    if (const char* e = getenv("VBZ_HIP_ROUTING")) c->routing = atoi(e);             // 2: the second launch group is not launched
#endif
    {
        std::vector<uint8_t> host(seq_tables_bytes());
        seq_tables_build(host.data());
        if (!ensure(c, c->seqtab, host.size()) ||
            hipMemcpy(c->seqtab.p, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error(nullptr, "could not upload the sequence tables");
            vbz_gpu_destroy(c);
            return nullptr;
        }
        std::vector<uint8_t> hostd(seq_dtables_bytes());
        seq_dtables_build(hostd.data());
        if (!ensure(c, c->seqdtab, hostd.size()) ||
            hipMemcpy(c->seqdtab.p, hostd.data(), hostd.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error(nullptr, "could not upload the sequence decoding tables");
            vbz_gpu_destroy(c);
            return nullptr;
        }
    }
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            set_error(nullptr, "hipStreamCreate failed");
            c->stream = nullptr;
            vbz_gpu_destroy(c);  // frees the uploaded tables as well
            return nullptr;
        }
        c->own_stream = true;
    }
    return c;
}

void vbz_gpu_destroy(vbz_gpu_ctx* c)
{
    if (!c) return;
    DeviceGuard dg(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (DevBuf* b : { &c->scratch, &c->meta, &c->gmeta, &c->route, &c->one_in, &c->one_out, &c->one_meta, &c->dbg, &c->seqtab, &c->seqdtab, &c->segmeta, &c->spanmeta, &c->spantmp, &c->fastmeta, &c->vgate, &c->encplan, &c->refpre, &c->reftab, &c->refrecs, &c->reflits, &c->splitmeta, &c->foreign_dev })
        if (b->p) (void)hipFree(b->p);
    if (c->large) vbz_gpu_destroy(c->large);
    if (c->half) vbz_gpu_destroy(c->half);
    for (hipEvent_t e : { c->ev_hfork, c->ev_hjoin, c->ev_hstagger })
        if (e) (void)hipEventDestroy(e);
    if (c->side.fork) (void)hipEventDestroy(c->side.fork);
    if (c->side.join) (void)hipEventDestroy(c->side.join);
    if (c->side.stream) (void)hipStreamDestroy(c->side.stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->canon_host) (void)hipHostFree(c->canon_host);
    if (c->foreign_host) (void)hipHostFree(c->foreign_host);
    if (c->ev_foreign) (void)hipEventDestroy(c->ev_foreign);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* vbz_gpu_stream(vbz_gpu_ctx* c) { return c ? (void*)c->stream : nullptr; }

void vbz_gpu_set_trailers(vbz_gpu_ctx* c, int enable)
{
    if (c) c->trailers = enable != 0;
}
void vbz_gpu_set_canonical(vbz_gpu_ctx* c, int enable)
{
    if (c) c->canonical = enable != 0;
}
const char* vbz_gpu_last_error(vbz_gpu_ctx* c) { return c ? c->error.c_str() : "no context"; }

int vbz_gpu_synchronize(vbz_gpu_ctx* c)
{
    if (!c) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream), "stream synchronize");
    return 0;
}

int vbz_gpu_compress_batch(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, const CompressionOptions* o, int sized)
{
    if (!c || !bt || !o) return -1;
    DeviceGuard dg(c->device);
    if (!valid_int_size(o) || (o->integer_size != 0 && o->vbz_version > 1)) {
        set_error(c, "unsupported options (integer_size=%u version=%u)", o->integer_size, o->vbz_version);
        return -2;
    }
    if (!plausible_extents(c, bt)) return -2;
    return compress_batch_impl(c, bt, o, sized);
}

int vbz_gpu_decompress_batch(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, const CompressionOptions* o, int sized)
{
    if (!c || !bt || !o) return -1;
    DeviceGuard dg(c->device);
    if (!valid_int_size(o) || (o->integer_size != 0 && o->vbz_version > 1)) {
        set_error(c, "unsupported options (integer_size=%u version=%u)", o->integer_size, o->vbz_version);
        return -2;
    }
    if (!plausible_extents(c, bt)) return -2;
    return decompress_batch_impl(c, bt, o, sized);
}

int vbz_gpu_svb_compress_batch(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, int integer_size, int zigzag, int version)
{
    if (!c || !bt) return -1;
    DeviceGuard dg(c->device);
    if ((integer_size != 1 && integer_size != 2 && integer_size != 4) || version > 1 || version < 0) return -2;
    Timed t(c, "svb_encode");
    HIPCHK(c, launch_svb_encode(to_rb(bt), integer_size, zigzag != 0, 0, true, version == 1 && integer_size == 1, nullptr, nullptr, c->stream), "svb_encode launch");
    return 0;
}

int vbz_gpu_svb_decompress_batch(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, int integer_size, int zigzag, int version)
{
    if (!c || !bt) return -1;
    DeviceGuard dg(c->device);
    if ((integer_size != 1 && integer_size != 2 && integer_size != 4) || version > 1 || version < 0) return -2;
    Timed t(c, "svb_decode");
    HIPCHK(c, launch_svb_decode(to_rb(bt), integer_size, zigzag != 0, version == 1 && integer_size == 1, c->stream), "svb_decode launch");
    return 0;
}

int vbz_gpu_zstd_compress_batch(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, const uint32_t* key_bytes)
{
    if (!c || !bt) return -1;
    DeviceGuard dg(c->device);
    Timed t(c, "zstd_encode");
    HIPCHK(c, launch_zstd_encode(to_rb(bt), bt->src_size, 0, key_bytes, 0, nullptr, nullptr, nullptr, c->trailers, nullptr, nullptr, false, false, nullptr, c->stream), "zstd_encode launch");
    return 0;
}

int vbz_gpu_zstd_decompress_batch(vbz_gpu_ctx* c, const vbz_gpu_batch* bt)
{
    if (!c || !bt) return -1;
    DeviceGuard dg(c->device);
    if (!plausible_extents(c, bt)) return -2;
    return zstd_frames(c, to_rb(bt), E_ZSTD, bt->dst_bytes, nullptr);
}

int vbz_gpu_synth_lengths(vbz_gpu_ctx* c, uint64_t seed, uint64_t first, uint32_t n, uint32_t* out_len)
{
    if (!c) return -1;
    DeviceGuard dg(c->device);
    HIPCHK(c, launch_synth_lengths(seed, first, n, out_len, c->stream), "synth_lengths launch");
    return 0;
}

int vbz_gpu_synth_signal(vbz_gpu_ctx* c, uint64_t seed, uint64_t first, uint32_t n, void* dst, const uint64_t* off,
                         const uint32_t* len)
{
    if (!c) return -1;
    DeviceGuard dg(c->device);
    HIPCHK(c, launch_synth_signal(seed, first, n, (uint8_t*)dst, off, len, c->stream), "synth_signal launch");
    return 0;
}

int vbz_gpu_synth_u32(vbz_gpu_ctx* c, uint64_t seed, uint64_t first, uint32_t n, void* dst, const uint64_t* off, const uint32_t* len)
{
    if (!c) return -1;
    DeviceGuard dg(c->device);
    HIPCHK(c, launch_synth_u32(seed, first, n, (uint8_t*)dst, off, len, c->stream), "synth_u32 launch");
    return 0;
}

#ifdef VBZ_EXPERIMENTS
// Test aid (experiments build only): the first half of vbz_gpu_compress_batch for int16 zig-zag reads at level != 0 on the one-workgroup
// path -- scratch slots, svb_encode with its hand-over to the entropy stage (svb_kernels.hip CNT) -- and nothing of the entropy stage.
// Afterwards read i's svb stream stands at the bottom of its destination slot, result[i] = the stream's size, and plans_out (device,
// n_reads * vbz_gpu_x_plan_bytes() bytes) holds the per-read plans.  tests/test_gpu_handover.py holds them to a numpy statement of
// region_histogram's sample.
VBZ_EXPORT size_t vbz_gpu_x_plan_bytes(void) { return sizeof(EncPlan); }
VBZ_EXPORT int vbz_gpu_x_svb_handover(vbz_gpu_ctx* c, const vbz_gpu_batch* bt, void* plans_out)
{
    if (!c || !bt || !plans_out) return -1;
    DeviceGuard dg(c->device);
    const uint32_t n = bt->n_reads;
    if (n == 0) return 0;
    hipStream_t s = c->stream;
    ReadBatch rb = to_rb(bt);
    uint32_t num, den;
    svb_factor(2, true, &num, &den);
    const size_t scratch_need = (size_t)(((unsigned __int128)bt->src_bytes * num + den - 1) / den) + (size_t)n * 96 + 256;
    if (!ensure(c, c->scratch, scratch_need) || !ensure(c, c->meta, (size_t)n * 40 + 256) || !ensure(c, c->encplan, zstd_encode_plan_bytes(n))) return -1;
    MetaCarver mc(c->meta.p);
    uint64_t* svb_off = mc.take<uint64_t>(n);
    uint32_t* svb_cap = mc.take<uint32_t>(n);
    uint32_t* svb_size = mc.take<uint32_t>(n);
    uint32_t* gate = mc.take<uint32_t>(n);
    HIPCHK(c, launch_plan_scratch(n, bt->src_size, num, den, c->scratch.cap, svb_off, svb_cap, gate, false, s), "plan launch");
    ReadBatch a = rb;
    a.dst = (uint8_t*)c->scratch.p;
    a.dst_off = svb_off;
    a.dst_cap = svb_cap;
    a.result = svb_size;
    a.gate = gate;
    HIPCHK(c, launch_svb_encode(a, 2, true, 0, false, false, nullptr, c->encplan.p, s), "svb_encode launch");
    ReadBatch cp = rb;   // scratch slot -> bottom of the destination slot
    cp.src = (const uint8_t*)c->scratch.p;
    cp.src_off = svb_off;
    cp.src_size = svb_size;
    cp.gate = gate;
    HIPCHK(c, launch_copy_bytes(cp, 0, s), "copy launch");
    HIPCHK(c, hipMemcpyAsync(plans_out, c->encplan.p, (size_t)n * sizeof(EncPlan), hipMemcpyDeviceToDevice, s), "plan copy");
    return 0;
}
#endif

void vbz_gpu_profile_enable(vbz_gpu_ctx* c, int enable)
{
    if (!c) return;
    if (!enable) drain_profile(c);
    c->profiling = enable != 0;
    if (c->half) vbz_gpu_profile_enable(c->half, enable);
}

int vbz_gpu_decode_span_paths(vbz_gpu_ctx* c, uint32_t* by_spans)
{
    if (!c) return -1;
    DeviceGuard guard(c->device);
    if (by_spans) *by_spans = 0;
    if (c->last_span_frames == 0) return 0;
    std::vector<uint32_t> redo(c->last_span_frames);
    if (hipMemcpyAsync(redo.data(), c->last_span_redo, 4ull * redo.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    uint32_t nb = 0;
    for (uint32_t v : redo) nb += v == 0;
    if (by_spans) *by_spans = nb;
    return (int)c->last_span_frames;
}

static int decode_paths_one(vbz_gpu_ctx* c, uint32_t* batched, uint32_t* walked);

int vbz_gpu_decode_paths(vbz_gpu_ctx* c, uint32_t* batched, uint32_t* walked)
{
    if (!c) return -1;
    uint32_t b0 = 0, w0 = 0, b1 = 0, w1 = 0;
    const int n0 = decode_paths_one(c, &b0, &w0);
    const int n1 = (n0 >= 0 && c->last_split && c->half) ? decode_paths_one(c->half, &b1, &w1) : 0;   // (the upper half of a split call)
    if (batched) *batched = b0 + b1;
    if (walked) *walked = w0 + w1;
    return (n0 < 0 || n1 < 0) ? -1 : n0 + n1;
}

int vbz_gpu_decode_literals_ahead(vbz_gpu_ctx* c)
{
    if (!c) return -1;
    int total = 0;
    for (vbz_gpu_ctx* k : { c, (c->last_split && c->half) ? c->half : (vbz_gpu_ctx*)nullptr }) {
        if (!k || !k->last_frames || !k->last_walked || !zstd_ref_literals_enabled() || !k->reflits.p) continue;
        DeviceGuard guard(k->device);
        const uint32_t per = k->last_lit_units;
        if (per == 0) continue;
        std::vector<RefLits> lits((size_t)k->last_frames * per);
        if (hipMemcpyAsync(lits.data(), zstd_ref_lits(k->reflits.p, k->last_frames), sizeof(RefLits) * lits.size(), hipMemcpyDeviceToHost, k->stream) != hipSuccess ||
            hipStreamSynchronize(k->stream) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        if (k->trace) {   // per block ordinal: units the scan made, units whose literals stand
            std::vector<uint32_t> sk((size_t)k->last_frames * per);
            if (hipMemcpy(sk.data(), zstd_ref_lit_skip(k->reflits.p, k->last_frames), 4 * sk.size(), hipMemcpyDeviceToHost) == hipSuccess)
                for (uint32_t u = 0; u < per; ++u) {
                    uint32_t made = 0, done = 0;
                    for (uint32_t i = 0; i < k->last_frames; ++i) {
                        made += sk[(size_t)u * k->last_frames + i] == 0;
                        done += lits[(size_t)u * k->last_frames + i].blk != 0;
                    }
                    fprintf(stderr, "vbz_hip: literals beside the walk, unit %u: %u made by the scan, %u done\n", u, made, done);
                }
        }
        for (uint32_t i = 0; i < k->last_frames; ++i) {   // (frames with at least one block's literals ahead)
            bool any = false;
            for (uint32_t u = 0; u < per; ++u) any = any || lits[(size_t)u * k->last_frames + i].blk != 0;   // (record u of read i)
            total += any;
        }
    }
    return total;
}

static int decode_paths_one(vbz_gpu_ctx* c, uint32_t* batched, uint32_t* walked)
{
    DeviceGuard guard(c->device);
    if (batched) *batched = 0;
    if (walked) *walked = 0;
    const uint32_t n = c->last_frames;
    if (n == 0) return 0;
    std::vector<uint32_t> redo(n);
    std::vector<RefPre> pre(c->last_walked ? n : 0);
    if (hipMemcpyAsync(redo.data(), zstd_fast_redo(c->fastmeta.p, n), 4ull * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (!pre.empty() && hipMemcpyAsync(pre.data(), zstd_ref_pre(c->refpre.p), sizeof(RefPre) * (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    uint32_t nb = 0, nw = 0;
    for (uint32_t i = 0; i < n; ++i) {
        nb += redo[i] == 0;
        if (!pre.empty()) nw += pre[i].ok != 0;
    }
    if (c->trace && !pre.empty()) {  // why frames of other writers were left to the one-wavefront decoder (zstd_decode_ref.hip: BAIL)
        std::map<uint32_t, uint32_t> why;
        for (uint32_t i = 0; i < n; ++i)
            if (redo[i] && !pre[i].ok) ++why[pre[i].pad[0]];
        for (auto& e : why) fprintf(stderr, "vbz_hip: chains not walked: reason %u, %u frame(s)\n", e.first, e.second);
        double tb = 0, ch = 0, ns = 0;
        for (uint32_t i = 0; i < n; ++i)
            if (pre[i].ok) tb += pre[i].pad[1], ch += pre[i].pad[2], ns += pre[i].pad[3];
        if (nw) fprintf(stderr, "vbz_hip: walked chains: %u frames, last block: %.0f cycles for the tables, %.0f for the chain, %.0f sequences\n", nw, tb / nw, ch / nw, ns / nw);
    }
    if (batched) *batched = nb;
    if (walked) *walked = nw;
    return (int)n;
}

void vbz_gpu_profile_reset(vbz_gpu_ctx* c)
{
    if (!c) return;
    drain_profile(c);
    c->prof.clear();
    if (c->half) vbz_gpu_profile_reset(c->half);
}

int vbz_gpu_profile_read(vbz_gpu_ctx* c, const char** names, uint32_t* launches, double* total_ms, int cap)
{
    if (!c) return 0;
    drain_profile(c);
    if (c->half) {   // the upper halves of split calls: their launches count under the same labels
        drain_profile(c->half);
        for (auto& h : c->half->prof) {
            bool found = false;
            for (auto& e : c->prof)
                if (strcmp(e.name, h.name) == 0) {
                    e.launches += h.launches;
                    e.ms += h.ms;
                    found = true;
                    break;
                }
            if (!found) c->prof.push_back(h);
        }
        c->half->prof.clear();
    }
    int k = 0;
    for (auto& e : c->prof) {
        if (k < cap) {
            if (names) names[k] = e.name;
            if (launches) launches[k] = e.launches;
            if (total_ms) total_ms[k] = e.ms;
        }
        ++k;
    }
    return k;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// vbz.h : the reference's single-buffer API over host pointers
// ------------------------------------------------------------------------------------------------
namespace {

// The reference's calls are re-entrant and callers parallelise by calling from many threads (vbz.cpp has no
// statics).  A call here needs a context (stream, device buffers), so there is a pool of them: a caller takes an idle
// context or creates one -- up to VBZ_HIP_CONTEXTS (default 16) -- and calls from different threads overlap on the
// GPU, each on its own stream.  Beyond the limit callers wait for a context to come back.
std::mutex g_mutex;
std::condition_variable g_idle_cv;
std::vector<vbz_gpu_ctx*> g_idle;
unsigned g_created = 0;
unsigned g_limit = 0;        // 0: pool_limit(); lowered to the number of live contexts once a creation fails
bool g_ctx_failed = false;

unsigned pool_limit()
{
    unsigned n = 16;
    if (const char* e = getenv("VBZ_HIP_CONTEXTS")) n = (unsigned)atoi(e);
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}

vbz_gpu_ctx* take_ctx()
{
    std::unique_lock<std::mutex> lock(g_mutex);
    for (;;) {
        if (!g_idle.empty()) {
            vbz_gpu_ctx* c = g_idle.back();
            g_idle.pop_back();
            return c;
        }
        if (g_ctx_failed) return nullptr;
        if (g_created < (g_limit ? g_limit : pool_limit())) {
            ++g_created;
            lock.unlock();
            int dev = 0;
            if (const char* e = getenv("VBZ_HIP_DEVICE")) dev = atoi(e);
            vbz_gpu_ctx* c = vbz_gpu_create(dev, nullptr);
            lock.lock();
            if (c) return c;
            --g_created;
            if (g_created == 0) {
                g_ctx_failed = true;  // no usable device: every call reports a device failure
                g_idle_cv.notify_all();
                return nullptr;
            }
            // creation failed while other contexts live (e.g. hipMalloc under memory pressure): stop growing the pool
            // and wait for one of them to come back instead of retrying at once
            g_limit = g_created;
        }
        g_idle_cv.wait(lock);
    }
}

void give_ctx(vbz_gpu_ctx* c)
{
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        g_idle.push_back(c);
    }
    g_idle_cv.notify_one();
}

struct CtxLease
{
    vbz_gpu_ctx* c;
    CtxLease() : c(take_ctx()) {}
    ~CtxLease()
    {
        if (c) give_ctx(c);
    }
};

// vbz.h callers may be binaries compiled against the REFERENCE header and only re-linked: they test
// `ret >= VBZ_FIRST_ERROR` with the reference's value (-7).  A device failure is therefore reported with the
// reference's own VBZ_OUT_OF_MEMORY_ERROR ("the device or its memory is not available") plus a line on stderr;
// the detailed cause is in vbz_gpu_last_error / on stderr.  VBZ_DEVICE_ERROR (-8) only appears in vbz_gpu.h results.
vbz_size_t device_failure()
{
    fprintf(stderr, "vbz_hip: no usable gfx950 device (or a device operation failed); this library has no CPU path\n");
    return VBZ_OUT_OF_MEMORY_ERROR;
}

constexpr uint32_t ONE_PINNED_MAX = 1u << 20;   // reads / results up to this size go through pinned memory (run_one)

struct OneMeta  // device-side descriptors of a one-read batch
{
    uint64_t src_off, dst_off;
    uint32_t src_size, dst_cap, result, pad;
};

// run one read through the batch path: host src -> device -> kernels -> host dst
vbz_size_t run_one(bool compress, const void* src, vbz_size_t src_size, void* dst, vbz_size_t dst_cap, vbz_size_t dev_cap,
                   const CompressionOptions* o, int sized)
{
    CtxLease lease;
    vbz_gpu_ctx* c = lease.c;
    if (!c) return device_failure();
    DeviceGuard dg(c->device);
    // Device buffers: one_in = [OneMeta | 64 bytes | the read], one_out = the result.  Pinned host memory: [OneMeta | pad to 256 | the
    // read on its way in | 16 words of hand-back header + the result on its way out] -- reads and results of up to ONE_PINNED_MAX go
    // through it: ONE host-to-device copy from pinned memory (descriptors and read together), and on the way back no copy call and no
    // stream synchronisation at all: hand_back_kernel writes the result into the pinned area and raises a flag this thread polls
    // (a synchronisation costs 10-20 us, the two copies of a pageable call as much again; 100 KB over the link 2 us).  Larger calls
    // keep the plain copies (a host memcpy of megabytes costs more than it saves: DESIGN 4.7).
    const bool small_in = src_size <= ONE_PINNED_MAX, small_out = (compress ? dev_cap : dst_cap) <= ONE_PINNED_MAX;
    const size_t in_off = 256, out_off = in_off + (small_in ? (((size_t)src_size + 63) & ~(size_t)63) : 0);
    const uint32_t out_cap = small_out ? (compress ? dev_cap : dst_cap) : 0u;
    const size_t pin_need = out_off + 64 + out_cap + 64;
    if (!ensure(c, c->one_in, (size_t)src_size + 512) || !ensure(c, c->one_out, (size_t)dev_cap + 64)) return VBZ_OUT_OF_MEMORY_ERROR;
    if (c->pinned_cap < pin_need) {
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr;
        c->pinned_cap = 0;
        const size_t want = std::max<size_t>(pin_need, 1u << 20);
        // (host-coherent: the hand-back flag is polled while the kernel that raises it may still be running)
        if (hipHostMalloc(&c->pinned, want, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            c->pinned = nullptr;
            return VBZ_OUT_OF_MEMORY_ERROR;
        }
        c->pinned_cap = want;
    }
    uint8_t* pin = (uint8_t*)c->pinned;
    OneMeta* hm = (OneMeta*)pin;
    hm->src_off = 0;
    hm->dst_off = 0;
    hm->src_size = src_size;
    hm->dst_cap = dev_cap;
    hm->result = VBZ_DEVICE_ERROR;
    hm->pad = 0;
    hipStream_t s = c->stream;
    uint8_t* din = (uint8_t*)c->one_in.p;       // device: the descriptors, then the read at + 256
    bool ok = true;
    if (small_in) {
        if (src_size) memcpy(pin + in_off, src, src_size);
        ok &= hipMemcpyAsync(din, pin, in_off + src_size, hipMemcpyHostToDevice, s) == hipSuccess;
    } else {
        ok &= hipMemcpyAsync(din, pin, sizeof(OneMeta), hipMemcpyHostToDevice, s) == hipSuccess;
        ok &= hipMemcpyAsync(din + in_off, src, src_size, hipMemcpyHostToDevice, s) == hipSuccess;
    }
    if (!ok) {
        set_error(c, "host to device copy failed");
        return device_failure();
    }
    OneMeta* dm = (OneMeta*)din;
    vbz_gpu_batch bt;
    memset(&bt, 0, sizeof bt);
    bt.n_reads = 1;
    bt.src = din + in_off;
    bt.src_off = &dm->src_off;
    bt.src_size = &dm->src_size;
    bt.src_bytes = src_size;
    bt.dst = c->one_out.p;
    bt.dst_off = &dm->dst_off;
    bt.dst_cap = &dm->dst_cap;
    bt.dst_bytes = dev_cap;
    bt.result = &dm->result;
    int rc = compress ? compress_batch_impl(c, &bt, o, sized, true) : decompress_batch_impl(c, &bt, o, sized, true);
    if (rc != 0) {
        (void)hipStreamSynchronize(s);   // (the copy out of the pinned area may still be in flight: the next call rewrites it)
        return device_failure();
    }
    uint32_t result = VBZ_DEVICE_ERROR;
    uint32_t have = 0;   // bytes of the result that have arrived in pinned memory
    volatile uint32_t* hb = (volatile uint32_t*)(pin + out_off);
    if (small_out) {
        const uint32_t seq = ++c->one_seq ? c->one_seq : ++c->one_seq;   // (never 0: what the flag holds between calls)
        hb[2] = 0;
        if (launch_hand_back(&dm->result, (const uint8_t*)c->one_out.p, (uint32_t*)(pin + out_off), out_cap, seq, &dm->pad, s) != hipSuccess) {
            set_error(c, "hand-back launch failed: %s", hipGetErrorString(hipGetLastError()));
            (void)hipStreamSynchronize(s);
            return device_failure();
        }
        // poll the flag; now and then ask the stream whether it is still alive (a faulting kernel never raises the flag)
        bool done = false;
        for (uint64_t spin = 0; !done; ++spin) {
            if (__atomic_load_n((const uint32_t*)&hb[2], __ATOMIC_ACQUIRE) == seq) {
                done = true;
                break;
            }
            if ((spin & 0xFFFu) == 0xFFFu) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) {   // the stream has drained: the flag is there, or the launch chain failed
                    done = __atomic_load_n((const uint32_t*)&hb[2], __ATOMIC_ACQUIRE) == seq;
                    break;
                }
                if (q != hipErrorNotReady) break;
            }
        }
        if (!done) {
            (void)hipStreamSynchronize(s);
            set_error(c, "kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
            return device_failure();
        }
        result = hb[0];
        have = hb[1];
    } else {
        if (hipMemcpyAsync(&hm->result, &dm->result, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            set_error(c, "kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
            return device_failure();
        }
        result = hm->result;
    }
    if (result == VBZ_DEVICE_ERROR) return device_failure();
    if (result >= VBZ_FIRST_ERROR) return result;
    if (result > dst_cap) return VBZ_DESTINATION_SIZE_ERROR;
    if (have) memcpy(dst, pin + out_off + 64, have);
    if (result > have && (hipMemcpyAsync((uint8_t*)dst + have, (const uint8_t*)c->one_out.p + have, result - have, hipMemcpyDeviceToHost, s) != hipSuccess ||
                          hipStreamSynchronize(s) != hipSuccess)) {
        set_error(c, "device to host copy failed");
        return device_failure();
    }
    return result;
}

vbz_size_t max_svb_size(unsigned integer_size, vbz_size_t source_size)
{
    // vbz/v0/vbz_streamvbyte.cpp:7-18 == vbz/v1/vbz_streamvbyte.cpp:9-20
    if (source_size % integer_size != 0) return VBZ_INPUT_SIZE_ERROR;
    const uint32_t count = source_size / integer_size;
    return (vbz_size_t)((uint64_t)(count + 3) / 4 + 4ull * count);  // truncation to 32 bits as in the reference
}

}  // namespace

extern "C" {

bool vbz_is_error(vbz_size_t v) { return v >= VBZ_DEVICE_ERROR; }  // the reference's seven codes and the vbz_gpu.h extension

char const* vbz_error_string(vbz_size_t v)
{
    switch (v) {
    case VBZ_ZSTD_ERROR: return "VBZ_ZSTD_ERROR";
    case VBZ_INPUT_SIZE_ERROR: return "VBZ_INPUT_SIZE_ERROR";
    case VBZ_INTEGER_SIZE_ERROR: return "VBZ_INTEGER_SIZE_ERROR";
    case VBZ_DESTINATION_SIZE_ERROR: return "VBZ_DESTINATION_SIZE_ERROR";
    case VBZ_STREAMVBYTE_STREAM_ERROR: return "VBZ_STREAMVBYTE_STREAM_ERROR";
    case VBZ_VERSION_ERROR: return "VBZ_VERSION_ERROR";
    case VBZ_OUT_OF_MEMORY_ERROR: return "VBZ_OUT_OF_MEMORY_ERROR";
    case VBZ_DEVICE_ERROR: return "VBZ_DEVICE_ERROR";
    default: return "VBZ_UNKNOWN_ERROR";
    }
}

vbz_size_t vbz_max_compressed_size(vbz_size_t source_size, CompressionOptions const* o)
{
    if (!valid_int_size(o)) return VBZ_INTEGER_SIZE_ERROR;
    vbz_size_t max_size = source_size;
    if (o->integer_size != 0) {
        if (o->vbz_version > 1) return VBZ_VERSION_ERROR;
        max_size = max_svb_size(o->integer_size, max_size);
        if (vbz_is_error(max_size)) return max_size;
    }
    if (o->zstd_compression_level != 0) max_size = (vbz_size_t)zstd_bound(max_size);
    return max_size + 4;  // always room for the sized header (vbz.cpp:112-113)
}

vbz_size_t vbz_compress(void const* source, vbz_size_t source_size, void* destination, vbz_size_t destination_capacity,
                        CompressionOptions const* o)
{
    if (!valid_int_size(o)) return VBZ_INTEGER_SIZE_ERROR;
    if (o->integer_size != 0) {
        if (o->vbz_version > 1) return VBZ_VERSION_ERROR;
        const vbz_size_t max_svb = max_svb_size(o->integer_size, source_size);
        if (vbz_is_error(max_svb)) return max_svb;  // vbz.cpp:153-160
        if (o->zstd_compression_level == 0 && max_svb > destination_capacity) return VBZ_DESTINATION_SIZE_ERROR;
    } else if (o->zstd_compression_level == 0 && source_size > destination_capacity) {
        return VBZ_DESTINATION_SIZE_ERROR;
    }
    const vbz_size_t bound = vbz_max_compressed_size(source_size, o);
    const vbz_size_t dev_cap = destination_capacity < bound ? destination_capacity : bound;
    return run_one(true, source, source_size, destination, destination_capacity, dev_cap, o, 0);
}

vbz_size_t vbz_decompress(void const* source, vbz_size_t source_size, void* destination, vbz_size_t destination_size,
                          CompressionOptions const* o)
{
    if (!valid_int_size(o)) return VBZ_INTEGER_SIZE_ERROR;
    // the reference checks the version only after the zstd stage (vbz.cpp:282-290); an invalid version
    // can never succeed, so it is reported up front
    if (o->integer_size != 0 && o->vbz_version > 1) return VBZ_VERSION_ERROR;
    return run_one(false, source, source_size, destination, destination_size, destination_size, o, 0);
}

vbz_size_t vbz_compress_sized(void const* source, vbz_size_t source_size, void* destination, vbz_size_t destination_capacity,
                              CompressionOptions const* o)
{
    if (!valid_int_size(o)) return VBZ_INTEGER_SIZE_ERROR;
    if (destination_capacity < 4) return VBZ_DESTINATION_SIZE_ERROR;
    if (o->integer_size != 0) {
        if (o->vbz_version > 1) return VBZ_VERSION_ERROR;
        const vbz_size_t max_svb = max_svb_size(o->integer_size, source_size);
        if (vbz_is_error(max_svb)) return max_svb;  // NOTE: the reference returns this error + 4 (vbz.cpp:321-329)
        if (o->zstd_compression_level == 0 && max_svb > destination_capacity - 4) return VBZ_DESTINATION_SIZE_ERROR;
    } else if (o->zstd_compression_level == 0 && source_size > destination_capacity - 4) {
        return VBZ_DESTINATION_SIZE_ERROR;
    }
    const vbz_size_t bound = vbz_max_compressed_size(source_size, o);
    const vbz_size_t dev_cap = destination_capacity < bound ? destination_capacity : bound;
    return run_one(true, source, source_size, destination, destination_capacity, dev_cap, o, 1);
}

vbz_size_t vbz_decompress_sized(void const* source, vbz_size_t source_size, void* destination, vbz_size_t destination_capacity,
                                CompressionOptions const* o)
{
    if (!valid_int_size(o)) return VBZ_INTEGER_SIZE_ERROR;
    if (source_size < 4) return VBZ_INPUT_SIZE_ERROR;
    uint32_t original;
    memcpy(&original, source, 4);
    if (destination_capacity < original) return VBZ_DESTINATION_SIZE_ERROR;
    if (o->integer_size != 0 && o->vbz_version > 1) return VBZ_VERSION_ERROR;
    return run_one(false, source, source_size, destination, destination_capacity, destination_capacity, o, 1);
}

vbz_size_t vbz_decompressed_size(void const* source, vbz_size_t source_size, CompressionOptions const* o)
{
    if (!valid_int_size(o)) return VBZ_INTEGER_SIZE_ERROR;
    if (source_size < 4) return VBZ_INPUT_SIZE_ERROR;
    uint32_t original;
    memcpy(&original, source, 4);
    return original;
}

}  // extern "C"
