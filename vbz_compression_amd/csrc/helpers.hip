// helpers.hip -- small bookkeeping kernels around the two stages, and the on-device generator of the
// synthetic workload.  None of these is on the critical path.
#include "vbz_kernels.h"

namespace vbzhip {

namespace {

// ---- scratch planning: exclusive scan of per-read slot sizes ---------------------------------------
// slot(i) = align16(bound(raw_size[i])) + 48, bound = num*size/den + 8 (the worst-case svb size).
// off[i] = sum of earlier slots; cap[i] = slot - 32 (16 bytes of slack on each side stay unused).
// One 1024-thread workgroup per 1024 reads; a workgroup adds up the slots of all reads in front of its own (64 coalesced
// loads per thread for the 64th workgroup of a 65 536-read batch) instead of waiting for its predecessors: no hand-over
// between workgroups, no extra memory, 0.12 -> 0.05 ms per call against the one-workgroup loop it replaces.
__global__ __launch_bounds__(1024) void plan_scratch_kernel(uint32_t n, const uint32_t* raw_size, uint32_t mul_num,
                                                            uint32_t mul_den, uint64_t limit, uint64_t* off, uint32_t* cap,
                                                            uint32_t* gate, uint32_t gate_is_input)
{
    __shared__ uint64_t wsum[16], wpre[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t base = blockIdx.x * 1024u;
    // A read that has already failed, or is not in this launch group, gets an empty slot.  The gates are read here while
    // other workgroups write theirs (0 or E_OOM, below): both of those values mean "has a slot" to every reader, so all
    // workgroups add up the same sizes whatever the order they run in (a read that comes in with E_OOM keeps it, and a slot).
    auto slot_of = [&](uint32_t i, bool& gated) -> uint64_t {
        const uint32_t g = gate_is_input ? gate[i] : 0u;
        gated = g >= GATE_SKIP;
        const bool empty = gated && g != E_OOM;
        const uint64_t bound = empty ? 0 : ((uint64_t)raw_size[i] * mul_num + mul_den - 1) / mul_den + 8;
        return ((bound + 15) & ~15ull) + 48;
    };
    // everything in front of this workgroup's reads
    uint64_t before = 0;
    for (uint32_t i = (uint32_t)tid; i < base; i += 1024u) {
        bool g;
        before += slot_of(i, g);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
    if (lane == 0) wpre[w] = before;
    // this workgroup's reads
    const uint32_t i = base + (uint32_t)tid;
    uint64_t slot = 0;
    bool gated = false;
    if (i < n) slot = slot_of(i, gated);
    uint64_t inc = slot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint64_t pre = 0;
    for (int k = 0; k < 16; ++k) pre += wpre[k];
    for (int k = 0; k < w; ++k) pre += wsum[k];
    if (i < n) {
        const uint64_t o = pre + inc - slot;
        off[i] = o + 16;
        const uint64_t c = slot - 32;
        cap[i] = c > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)c;
        if (!gated) gate[i] = (o + slot > limit) ? E_OOM : 0u;
    }
}

// segment counts of a batch of few, large reads: exclusive scan by one 1024-thread workgroup.  max_segs is what the host sized
// the segment tables (and the grids) for -- the arena's extent; reads may alias their source bytes, so the sizes can add up to
// more: a read whose segments would not fit keeps ONE segment and gets gate_out[i] = E_OOM (a per-read error instead of
// writes behind the tables); gate_out[i] is the input gate otherwise (0 without one).
// sp (sp.off != nullptr; n <= 1024 only): the scratch plan of plan_scratch_kernel for the same reads in the same launch -- a call of few
// reads is a chain of short dependent launches, and this one saves two of them (the plan and the copy of the gates in front of it):
// sp.gate[i] = gate_out[i], or E_OOM for a read whose scratch slot does not fit.
__global__ __launch_bounds__(1024) void seg_plan_kernel(uint32_t n, const uint32_t* size, uint32_t unit, const uint32_t* gate, uint32_t max_segs,
                                                        uint32_t* seg_first, uint32_t* gate_out, ScratchPlan sp)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        uint32_t c = 0, g = 0;
        if (i < n) {
            g = gate ? gate[i] : 0u;
            const uint32_t sz = g >= GATE_SKIP || size[i] >= E_FIRST ? 0u : size[i];
            c = sz ? (uint32_t)(((uint64_t)sz + unit - 1) / unit) : 1u;
        }
        // a read fits if its segments end inside the tables with one segment left for every read behind it; the scan is
        // redone with the misfits' counts at one (they are rare: one retry settles it, the check is repeated to be sure)
        for (int attempt = 0; attempt < 2; ++attempt) {
            const uint32_t inc = wave_incl_scan_u32(c);
            if (lane == 63) wsum[w] = inc;
            __syncthreads();
            uint32_t pre = carry_s;
            for (int k = 0; k < w; ++k) pre += wsum[k];
            const uint32_t first = pre + inc - c;
            const bool fits = i >= n || (uint64_t)first + c + (n - 1 - i) <= max_segs;
            __syncthreads();
            if (attempt == 1 || !__syncthreads_or(!fits)) {
                if (i < n) {
                    seg_first[i] = first;
                    if (!fits) g = E_OOM;   // (second attempt only: cannot happen with c == 1 unless max_segs < n)
                    gate_out[i] = g;
                }
                if (tid == 1023) carry_s = pre + inc;
                break;
            }
            if (!fits) {
                c = 1;
                g = E_OOM;
            }
        }
        __syncthreads();
    }
    if (tid == 0) seg_first[n] = carry_s;
    if (sp.off == nullptr) return;
    // ---- the scratch plan (plan_scratch_kernel's arithmetic; n <= 1024: thread i has just written gate_out[i] itself)
    __shared__ uint64_t ssum[16];
    __syncthreads();
    const uint32_t i = (uint32_t)tid;
    uint64_t slot = 0;
    uint32_t g = 0;
    bool gated = false;
    if (i < n) {
        g = gate_out[i];
        gated = g >= GATE_SKIP;
        const bool empty = gated && g != E_OOM;
        const uint64_t bound = empty ? 0 : ((uint64_t)size[i] * sp.num + sp.den - 1) / sp.den + 8;
        slot = ((bound + 15) & ~15ull) + 48;
    }
    uint64_t inc = slot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) ssum[w] = inc;
    __syncthreads();
    uint64_t pre = 0;
    for (int k = 0; k < w; ++k) pre += ssum[k];
    if (i < n) {
        const uint64_t o = pre + inc - slot;
        sp.off[i] = o + 16;
        const uint64_t cc = slot - 32;
        sp.cap[i] = cc > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)cc;
        sp.gate[i] = gated ? g : ((o + slot > sp.limit) ? E_OOM : 0u);
    }
}

// ---- per-read routing (vbz_api.hip): the reads of `min_bytes` raw bytes and more, in batch order, at most max_reads of them and
// max_bytes in all, become the compact second launch group (descriptors l_*[max_reads], l_map = their batch indices, *l_count;
// l_gate = the read's input gate, GATE_SKIP behind the last routed one); gate_small[i] = GATE_SKIP for them, the input gate
// (or 0) for everybody else.  Two launches: every thread looks at one read and long reads put their index on a candidate
// list (they are rare); one workgroup sorts the list and takes from its front.
constexpr uint32_t ROUTE_CAND_MAX = 1024;

__global__ __launch_bounds__(256) void route_flag_kernel(ReadBatch b, const uint32_t* raw_size, uint32_t min_bytes, uint32_t* gate_small, uint32_t* cand,
                                                         uint32_t* cand_count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_reads) return;
    const uint32_t g = b.gate ? b.gate[i] : 0u;
    const uint32_t sz = raw_size[i];
    bool big = g < GATE_SKIP && sz >= min_bytes && sz < E_FIRST;
    if (big) {
        const uint32_t at = atomicAdd(cand_count, 1u);
        if (at < ROUTE_CAND_MAX) cand[at] = i;
        else big = false;   // (more long reads than the list holds: this one stays in the first group)
    }
    gate_small[i] = big ? GATE_SKIP : g;
}

__global__ __launch_bounds__(1024) void route_pick_kernel(ReadBatch b, const uint32_t* raw_size, uint32_t max_reads, uint64_t max_bytes, uint32_t* gate_small,
                                                          const uint32_t* cand, const uint32_t* cand_count, uint64_t* l_src_off, uint32_t* l_src_size,
                                                          uint64_t* l_dst_off, uint32_t* l_dst_cap, uint32_t* l_gate, uint32_t* l_map, uint32_t* l_count)
{
    __shared__ uint32_t key[ROUTE_CAND_MAX];
    const uint32_t tid = threadIdx.x;
    const uint32_t n = *cand_count < ROUTE_CAND_MAX ? *cand_count : ROUTE_CAND_MAX;
    for (uint32_t j = tid; j < max_reads; j += 1024) {
        l_gate[j] = GATE_SKIP;
        l_src_off[j] = 0;
        l_dst_off[j] = 0;
        l_src_size[j] = 0;
        l_dst_cap[j] = 0;
        l_map[j] = 0;
    }
    if (tid == 0) *l_count = 0;
    if (n == 0) return;
    key[tid] = tid < n ? cand[tid] : 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t k = 2; k <= ROUTE_CAND_MAX; k <<= 1)   // bitonic sort, ascending (the list came in any order)
        for (uint32_t d = k >> 1; d > 0; d >>= 1) {
            const uint32_t o = tid ^ d;
            if (o > tid) {
                const uint32_t x = key[tid], y = key[o];
                const bool up = (tid & k) == 0;
                if ((x > y) == up) {
                    key[tid] = y;
                    key[o] = x;
                }
            }
            __syncthreads();
        }
    // More long reads than the list holds: WHICH of them got onto it depends on the order the atomics arrived in, so nothing is
    // routed (the same batch must give the same compressed bytes every time; such a batch is nearly one the shape rule takes anyway)
    const bool overflow = *cand_count > ROUTE_CAND_MAX;
    if (tid == 0) {   // (a handful of entries: one thread walks them)
        uint64_t bytes = 0;
        uint32_t taken = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t i = key[j];
            const uint32_t g = b.gate ? b.gate[i] : 0u;
            if (!overflow && taken < max_reads && bytes + raw_size[i] <= max_bytes) {
                bytes += raw_size[i];
                l_src_off[taken] = b.src_off[i];
                l_src_size[taken] = b.src_size[i];
                l_dst_off[taken] = b.dst_off[i];
                l_dst_cap[taken] = b.dst_cap[i];
                l_gate[taken] = g;
                l_map[taken] = i;
                ++taken;
            } else {
                gate_small[i] = g;   // not routed after all
            }
        }
        *l_count = taken;
    }
}

// ---- canonical mode (vbz_api.hip): which reads take the large-read path is decided by each read's OWN size, whatever the batch around it.
// gate_small[i] = GATE_SKIP for reads of min_bytes raw bytes and more (the input gate, or 0, for the others); gate_large[i] the reverse
// (a read that carries an error is reported by the first group); counts[0] = the reads of min_bytes and more.
__global__ __launch_bounds__(256) void canon_classify_kernel(uint32_t n, const uint32_t* raw_size, const uint32_t* gate, uint32_t min_bytes, uint32_t* gate_small,
                                                             uint32_t* gate_large, uint32_t* counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = (i < n && gate) ? gate[i] : 0u;
    const uint32_t sz = i < n ? raw_size[i] : 0u;
    const bool big = i < n && g < GATE_SKIP && sz >= min_bytes && sz < E_FIRST;
    if (i < n) {
        gate_small[i] = big ? GATE_SKIP : g;
        gate_large[i] = big ? g : GATE_SKIP;
    }
    const uint64_t m = __ballot(big);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counts[0], (uint32_t)__popcll(m));
}

// how many words of a[0..n) are not zero (the frames the batched decoder left to the one-wavefront decoder: vbz_api.hip, foreign_note)
__global__ __launch_bounds__(256) void count_nonzero_kernel(const uint32_t* a, uint32_t n, uint32_t* out)
{
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += a[i] != 0;
    c = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(c), 63);   // (all 64 lanes are here: the wavefront's total)
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// results of the second launch group back to the reads they belong to
__global__ void route_results_kernel(const uint32_t* l_result, const uint32_t* l_map, const uint32_t* l_count, uint32_t max_reads, uint32_t* result)
{
    const uint32_t j = threadIdx.x;
    if (j < max_reads && j < *l_count) result[l_map[j]] = l_result[j];
}

// The descriptor table of a batch is untrusted like the data (include/vbz_gpu.h): a read whose source or destination slot does not
// lie inside the arena the caller declared gets its error here, before any kernel forms an address from it.
__global__ void validate_batch_kernel(uint32_t n, const uint64_t* src_off, const uint32_t* src_size, uint64_t src_bytes,
                                      const uint64_t* dst_off, const uint32_t* dst_cap, uint64_t dst_bytes, uint32_t* gate)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t so = src_off[i], d0 = dst_off[i];
    uint32_t g = 0;
    if (so > src_bytes || (uint64_t)src_size[i] > src_bytes - so) g = E_INPUT_SIZE;
    else if (d0 > dst_bytes || (uint64_t)dst_cap[i] > dst_bytes - d0) g = E_DESTINATION_SIZE;
    gate[i] = g;
}

__global__ void parse_sized_kernel(uint32_t n, const uint8_t* src, const uint64_t* src_off, const uint32_t* src_size,
                                   const uint32_t* dst_cap, const uint32_t* gate_in, uint64_t* pay_off, uint32_t* pay_size, uint32_t* orig_size,
                                   uint32_t* gate)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sz = src_size[i];
    uint32_t g = 0, orig = 0;
    if (gate_in && gate_in[i] >= GATE_SKIP) {
        g = gate_in[i];
    } else if (sz < 4) {
        g = E_INPUT_SIZE;  // vbz/vbz.cpp:345-348
    } else {
        const uint8_t* p = src + src_off[i];
        orig = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        if (dst_cap[i] < orig) g = E_DESTINATION_SIZE;  // vbz/vbz.cpp:353-356
    }
    pay_off[i] = src_off[i] + 4;
    pay_size[i] = sz >= 4 ? sz - 4 : 0;
    orig_size[i] = orig;
    gate[i] = g;
}

// integer_size == 0 && level == 0: plain copy (vbz/vbz.cpp:130-133 copy_buffer)
__global__ __launch_bounds__(256) void copy_bytes_kernel(ReadBatch b, uint32_t hdr)
{
    const uint32_t r = blockIdx.x;
    if (b.gate && b.gate[r] >= GATE_SKIP) {
        if (threadIdx.x == 0 && b.gate[r] != GATE_SKIP) b.result[r] = b.gate[r];
        return;
    }
    const uint32_t n = b.src_size[r];
    if ((uint64_t)n + hdr > b.dst_cap[r]) {
        if (threadIdx.x == 0) b.result[r] = E_DESTINATION_SIZE;
        return;
    }
    const uint8_t* s = b.src + b.src_off[r];
    uint8_t* d = b.dst + b.dst_off[r];
    if (hdr && threadIdx.x < 4) d[threadIdx.x] = (uint8_t)(n >> (8 * threadIdx.x));
    d += hdr;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
    if (threadIdx.x == 0) b.result[r] = n + hdr;
}

// The single-buffer API's hand-back (vbz_api.hip run_one): HAND_BACK_WGS workgroups copy a read's result -- `*result` bytes at `src`, if
// they are no error code and fit `host_cap` -- straight into pinned HOST memory (one workgroup alone moved 84 KB over the link in 9.5 us:
// the writes a CU keeps in flight, not the link, set that rate); the workgroup that finishes last (a ticket in device memory, which
// it takes back to zero for the next call) writes the result word and then raises the flag the host polls: no device-to-host copy
// call, no stream synchronisation on the way back.  host[0] = result, host[1] = bytes copied, host[2] = flag (the call's sequence
// number), the bytes from host + 16 on.
constexpr uint32_t HAND_BACK_WGS = 16;
__global__ __launch_bounds__(256) void hand_back_kernel(const uint32_t* result, const uint8_t* src, uint32_t* host, uint32_t host_cap, uint32_t seq,
                                                        uint32_t* ticket)
{
    __shared__ uint32_t last_s;
    const uint32_t r = *result;
    uint32_t n = 0;
    if (r < E_FIRST && r <= host_cap) n = r;
    uint8_t* d = reinterpret_cast<uint8_t*>(host + 16);
    const uint32_t nv = n >> 4;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nv; i += HAND_BACK_WGS * 256u) {   // (src is the context's own allocation: 16-byte aligned)
        const uint4 v = *reinterpret_cast<const uint4*>(src + 16ull * i);
        *reinterpret_cast<uint4*>(d + 16ull * i) = v;
    }
    if (blockIdx.x == 0)
        for (uint32_t i = (nv << 4) + threadIdx.x; i < n; i += 256) d[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last_s = atomicAdd(ticket, 1u) == HAND_BACK_WGS - 1u ? 1u : 0u;
    __syncthreads();
    if (last_s && threadIdx.x == 0) {
        *ticket = 0;
        host[0] = r;
        host[1] = n;
        __threadfence_system();
        __hip_atomic_store(&host[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- synthetic workload -----------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ uint64_t synth_key(uint64_t seed, uint64_t r) { return mix64(seed * 0x100000001B3ull + r); }

__global__ void synth_lengths_kernel(uint64_t seed, uint64_t first, uint32_t n, uint32_t* out_len)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_len[i] = 90000u + (uint32_t)(mix64(synth_key(seed, first + i) ^ 0xC2B2AE3D27D4EB4Full) % 20001u);
}

__global__ __launch_bounds__(256) void synth_signal_kernel(uint64_t seed, uint64_t first, uint8_t* dst, const uint64_t* off,
                                                           const uint32_t* len)
{
    const uint32_t r = blockIdx.x;
    const uint64_t key = synth_key(seed, first + r);
    int16_t* out = reinterpret_cast<int16_t*>(dst + off[r]);
    const uint32_t n = len[r];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t hs = mix64(key ^ ((uint64_t)(i / 32) * 0xD6E8FEB86659FD93ull));
        const int32_t level = 200 + (int32_t)(hs % 321u);
        uint64_t hn = mix64(key ^ ((uint64_t)i * 0xA24BAED4963EE407ull) ^ 0x5555555555555555ull);
        int32_t noise = -60;
#pragma unroll
        for (int k = 0; k < 8; ++k) noise += (int32_t)((hn >> (4 * k)) & 15u);
        int32_t x = level + noise;
        x = x < -4096 ? -4096 : (x > 4095 ? 4095 : x);
        out[i] = (int16_t)x;
    }
}

__global__ __launch_bounds__(256) void synth_u32_kernel(uint64_t seed, uint64_t first, uint8_t* dst, const uint64_t* off,
                                                        const uint32_t* len)
{
    const uint32_t r = blockIdx.x;
    const uint64_t key = synth_key(seed, first + r);
    uint32_t* out = reinterpret_cast<uint32_t*>(dst + off[r]);
    const uint32_t n = len[r];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t h = mix64(key ^ ((uint64_t)i * 0xA24BAED4963EE407ull));
        const uint32_t sel = (uint32_t)(h & 127u);
        const uint32_t s = sel < 90 ? 24 : (sel < 115 ? 16 : (sel < 125 ? 8 : 0));
        out[i] = (uint32_t)(h >> 32) >> s;
    }
}

}  // namespace

hipError_t launch_plan_scratch(uint32_t n, const uint32_t* raw_size, uint32_t mul_num, uint32_t mul_den, uint64_t limit,
                               uint64_t* off, uint32_t* cap, uint32_t* gate, bool gate_is_input, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(plan_scratch_kernel, dim3((n + 1023u) / 1024u), dim3(1024), 0, s, n, raw_size, mul_num, mul_den, limit, off, cap, gate,
                       gate_is_input ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_seg_plan(uint32_t n, const uint32_t* size, uint32_t unit_bytes, const uint32_t* gate, uint32_t max_segs, uint32_t* seg_first,
                           uint32_t* gate_out, const ScratchPlan* scratch, hipStream_t s)
{
    ScratchPlan sp = {};
    if (scratch && n <= 1024) sp = *scratch;
    hipLaunchKernelGGL(seg_plan_kernel, dim3(1), dim3(1024), 0, s, n, size, unit_bytes, gate, max_segs, seg_first, gate_out, sp);
    return hipGetLastError();
}

hipError_t launch_route_reads(const ReadBatch& b, const uint32_t* raw_size, uint32_t min_bytes, uint32_t max_reads, uint64_t max_bytes, uint32_t* gate_small,
                              uint64_t* l_src_off, uint32_t* l_src_size, uint64_t* l_dst_off, uint32_t* l_dst_cap, uint32_t* l_gate, uint32_t* l_map,
                              uint32_t* l_count, uint32_t* cand, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    uint32_t* cand_count = cand + ROUTE_CAND_MAX;
    hipError_t e = hipMemsetAsync(cand_count, 0, 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(route_flag_kernel, dim3((b.n_reads + 255) / 256), dim3(256), 0, s, b, raw_size, min_bytes, gate_small, cand, cand_count);
    hipLaunchKernelGGL(route_pick_kernel, dim3(1), dim3(1024), 0, s, b, raw_size, max_reads, max_bytes, gate_small, cand, cand_count, l_src_off, l_src_size,
                       l_dst_off, l_dst_cap, l_gate, l_map, l_count);
    return hipGetLastError();
}

size_t route_cand_words() { return ROUTE_CAND_MAX + 4; }

hipError_t launch_count_nonzero(const uint32_t* a, uint32_t n, uint32_t* out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, 4, s);
    if (e != hipSuccess) return e;
    const uint32_t blocks = n < 256u * 64u ? (n + 255u) / 256u : 64u;
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, s, a, n, out);
    return hipGetLastError();
}

hipError_t launch_canon_classify(uint32_t n, const uint32_t* raw_size, const uint32_t* gate, uint32_t min_bytes, uint32_t* gate_small, uint32_t* gate_large,
                                 uint32_t* counts, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(counts, 0, 8, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(canon_classify_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, raw_size, gate, min_bytes, gate_small, gate_large, counts);
    return hipGetLastError();
}

hipError_t launch_route_results(const uint32_t* l_result, const uint32_t* l_map, const uint32_t* l_count, uint32_t max_reads, uint32_t* result, hipStream_t s)
{
    hipLaunchKernelGGL(route_results_kernel, dim3(1), dim3(64), 0, s, l_result, l_map, l_count, max_reads, result);
    return hipGetLastError();
}

hipError_t launch_parse_sized(uint32_t n, const uint8_t* src, const uint64_t* src_off, const uint32_t* src_size,
                              const uint32_t* dst_cap, const uint32_t* gate_in, uint64_t* pay_off, uint32_t* pay_size, uint32_t* orig_size,
                              uint32_t* gate, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(parse_sized_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, src, src_off, src_size, dst_cap, gate_in, pay_off,
                       pay_size, orig_size, gate);
    return hipGetLastError();
}

hipError_t launch_hand_back(const uint32_t* result, const uint8_t* src, uint32_t* host, uint32_t host_cap, uint32_t seq, uint32_t* ticket, hipStream_t s)
{
    hipLaunchKernelGGL(hand_back_kernel, dim3(HAND_BACK_WGS), dim3(256), 0, s, result, src, host, host_cap, seq, ticket);
    return hipGetLastError();
}

hipError_t launch_validate_batch(uint32_t n, const uint64_t* src_off, const uint32_t* src_size, uint64_t src_bytes, const uint64_t* dst_off,
                                 const uint32_t* dst_cap, uint64_t dst_bytes, uint32_t* gate, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(validate_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, src_off, src_size, src_bytes, dst_off, dst_cap, dst_bytes, gate);
    return hipGetLastError();
}

hipError_t launch_copy_bytes(const ReadBatch& b, uint32_t hdr, hipStream_t s)
{
    if (b.n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(b.n_reads), dim3(256), 0, s, b, hdr);
    return hipGetLastError();
}

hipError_t launch_synth_lengths(uint64_t seed, uint64_t first, uint32_t n, uint32_t* out_len, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_lengths_kernel, dim3((n + 255) / 256), dim3(256), 0, s, seed, first, n, out_len);
    return hipGetLastError();
}

hipError_t launch_synth_signal(uint64_t seed, uint64_t first, uint32_t n, uint8_t* dst, const uint64_t* off, const uint32_t* len,
                               hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_signal_kernel, dim3(n), dim3(256), 0, s, seed, first, dst, off, len);
    return hipGetLastError();
}

hipError_t launch_synth_u32(uint64_t seed, uint64_t first, uint32_t n, uint8_t* dst, const uint64_t* off, const uint32_t* len,
                            hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_u32_kernel, dim3(n), dim3(256), 0, s, seed, first, dst, off, len);
    return hipGetLastError();
}

}  // namespace vbzhip
