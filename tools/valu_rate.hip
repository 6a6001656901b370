// valu_rate.hip -- how many wave64 instructions per cycle one SIMD of an MI355X CU issues, by instruction class.
//
//     hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate > profiles/r06_valu_rate.txt
//
// DESIGN's "VALU busy %" columns divide a kernel's wave-level VALU instructions x C cycles by the SIMD-cycles of the launch; C (the
// cycles a wave64 instruction of that class holds its SIMD) is what this measures.  Each kernel runs W wavefronts per SIMD (workgroups
// of 256 threads = one wavefront per SIMD, W workgroups per CU, 256 CUs), every wavefront executing ITER x 32 instructions of ONE class
// on independent registers (no dependent chains inside a group of eight), bracketed by s_memtime on every wavefront; the figure printed
// is  (wave instructions one SIMD issued) / (the longest wavefront's s_memtime span), with the span also converted to shader cycles through
// the wall time of the launch (hipEvents) and the clock rocm-smi reports -- both are shown because s_memtime's unit is the thing in doubt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2048;
constexpr int GROUP = 32;   // instructions per loop trip

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

struct Stamp { unsigned long long t0, t1; };

// One kernel per class.  a[8] are the independent accumulators; the asm bodies only name %0..%7 (accumulators) and %8, %9 (operands).
#define KERNEL(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(Stamp* st, unsigned* sink, unsigned seed)                         \
    {                                                                                                              \
        __shared__ unsigned lds[4096];                                                                             \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13,  \
                 a7 = a0 * 17;                                                                                     \
        unsigned b = seed | 1u, c = (threadIdx.x * 4u) & 0x3FFCu, d = threadIdx.x * 9u;                                                  \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0;                                                  \
        __syncthreads();                                                                                           \
        unsigned long long t0 = __builtin_readcyclecounter();                                                     \
        for (int it = 0; it < ITER; ++it) {                                                                        \
            asm volatile(ASM ASM ASM ASM                                                                           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)          \
                         : "v"(b), "v"(c), "v"(d)                                                                  \
                         : "memory", "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                           \
        }                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
        unsigned long long t1 = __builtin_readcyclecounter();                                                     \
        if ((threadIdx.x & 63) == 0) {                                                                             \
            Stamp s = { t0, t1 };                                                                                  \
            st[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;                                                           \
        }                                                                                                          \
        sink[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ lds[threadIdx.x];           \
    }

// eight independent instructions of one class (x 4 in the KERNEL macro = GROUP)
#define A8(OP) OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n" OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7\n"
#define A8_3(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"

KERNEL(k_add_u32, A8("v_add_u32"))
KERNEL(k_xor_b32, A8("v_xor_b32"))
KERNEL(k_lshlrev_b32, A8("v_lshlrev_b32"))
KERNEL(k_pk_add_u16, A8("v_pk_add_u16"))
KERNEL(k_pk_lshlrev_b16, A8("v_pk_lshlrev_b16"))
KERNEL(k_perm_b32, A8_3("v_perm_b32"))
KERNEL(k_alignbit_b32, A8_3("v_alignbit_b32"))
KERNEL(k_bfe_u32, A8_3("v_bfe_u32"))
KERNEL(k_lshl_or_b32, A8_3("v_lshl_or_b32"))
KERNEL(k_mad_u32_u24, A8_3("v_mad_u32_u24"))
KERNEL(k_mul_lo_u32, A8("v_mul_lo_u32"))
KERNEL(k_bcnt, A8("v_bcnt_u32_b32"))
KERNEL(k_cndmask, "v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n"
                  "v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n")
// DPP move (row_shr:1): what the wave scans are made of
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                  "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                  "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                  "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
// LDS: every lane its own dword (no bank conflict) / its own byte of a dword shared by four lanes
#define L8(OP, OFFS) OP " %9, %0 offset:" #OFFS "\n" OP " %9, %1 offset:" #OFFS "+256\n" OP " %9, %2 offset:" #OFFS "+512\n" OP " %9, %3 offset:" #OFFS "+768\n" \
                     OP " %9, %4 offset:" #OFFS "+1024\n" OP " %9, %5 offset:" #OFFS "+1280\n" OP " %9, %6 offset:" #OFFS "+1536\n" OP " %9, %7 offset:" #OFFS "+1792\n"
KERNEL(k_ds_write_b32, L8("ds_write_b32", 0))
#define L8B(OP) OP " %10, %0 offset:0\n" OP " %10, %1 offset:1\n" OP " %10, %2 offset:2\n" OP " %10, %3 offset:3\n" \
                OP " %10, %4 offset:4\n" OP " %10, %5 offset:5\n" OP " %10, %6 offset:6\n" OP " %10, %7 offset:7\n"
KERNEL(k_ds_write_b8, L8B("ds_write_b8"))   // the encoder's pattern: lane l writes bytes 9 l + k
KERNEL(k_ds_or_b32, L8("ds_or_b32", 0))
KERNEL(k_ds_add_u32, L8("ds_add_u32", 0))
#define R8(OP) OP " %0, %9 offset:0\n" OP " %1, %9 offset:256\n" OP " %2, %9 offset:512\n" OP " %3, %9 offset:768\n" \
               OP " %4, %9 offset:1024\n" OP " %5, %9 offset:1280\n" OP " %6, %9 offset:1536\n" OP " %7, %9 offset:1792\n" "s_waitcnt lgkmcnt(0)\n"
KERNEL(k_ds_read_b32, R8("ds_read_b32"))
KERNEL(k_ds_read_u8, R8("ds_read_u8"))

KERNEL(k_sub_u32, A8("v_sub_u32"))
KERNEL(k_and_b32, A8("v_and_b32"))
KERNEL(k_or_b32, A8("v_or_b32"))
KERNEL(k_min_u32, A8("v_min_u32"))
KERNEL(k_lshrrev_b32, A8("v_lshrrev_b32"))
KERNEL(k_ashrrev_i32, A8("v_ashrrev_i32"))
KERNEL(k_mul_u32_u24, A8("v_mul_u32_u24"))
KERNEL(k_add3_u32, A8_3("v_add3_u32"))
KERNEL(k_and_or_b32, A8_3("v_and_or_b32"))
KERNEL(k_or3_b32, A8_3("v_or3_b32"))
KERNEL(k_xad_u32, A8_3("v_xad_u32"))
KERNEL(k_lshl_add_u32, A8_3("v_lshl_add_u32"))
KERNEL(k_add_lshl_u32, A8_3("v_add_lshl_u32"))
KERNEL(k_alignbyte_b32, A8_3("v_alignbyte_b32"))
KERNEL(k_pk_sub_i16, A8("v_pk_sub_i16"))
KERNEL(k_pk_ashrrev_i16, A8("v_pk_ashrrev_i16"))
KERNEL(k_pk_min_u16, A8("v_pk_min_u16"))
KERNEL(k_mbcnt_lo, A8("v_mbcnt_lo_u32_b32"))
KERNEL(k_mov_b32, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n"
                 "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\n")
// compares and selects: vcc written inside the group (a v_cndmask that reads a vcc nobody wrote measured 23 cycles)
KERNEL(k_cmp_vcc, "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n"
                  "v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n")
KERNEL(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %8, %1, vcc\n"
                      "v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %8, %2, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %8, %3, vcc\n")
KERNEL(k_cmp_sgpr_cndmask, "v_cmp_lt_u32 s[20:21], %0, %8\n v_cmp_lt_u32 s[22:23], %1, %8\n v_cmp_lt_u32 s[24:25], %2, %8\n v_cmp_lt_u32 s[26:27], %3, %8\n"
                           "v_cndmask_b32 %0, %8, %0, s[20:21]\n v_cndmask_b32 %1, %8, %1, s[22:23]\n v_cndmask_b32 %2, %8, %2, s[24:25]\n v_cndmask_b32 %3, %8, %3, s[26:27]\n")
KERNEL(k_cmp_addc, "v_cmp_lt_u32 vcc, %0, %8\n v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n"
                   "v_cmp_lt_u32 vcc, %2, %8\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                   "v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 5\n v_readlane_b32 s26, %6, 7\n v_readlane_b32 s27, %7, 9\n")
//KERNEL(k_ds_write_b64, "ds_write_b64 %9, %[p01] offset:0\n ds_write_b64 %9, %[p23] offset:2048\n ds_write_b64 %9, %[p45] offset:4096\n ds_write_b64 %9, %[p67] offset:6144\n"
                       //"ds_write_b64\n ds_write_b64 %9, %[p23] offset:10240\n ds_write_b64 %9, %[p45] offset:12288\n ds_write_b64 %9, %[p67] offset:14336\n")


#define RW8(OP) OP " %0, %9 offset:0\n" OP " %2, %9 offset:2048\n" OP " %4, %9 offset:4096\n" OP " %6, %9 offset:6144\n" \
                OP " %0, %9 offset:8192\n" OP " %2, %9 offset:10240\n" OP " %4, %9 offset:12288\n" OP " %6, %9 offset:14336\n" "s_waitcnt lgkmcnt(0)\n"
KERNEL(k_ds_bpermute, "ds_bpermute_b32 %0, %9, %0\n ds_bpermute_b32 %1, %9, %1\n ds_bpermute_b32 %2, %9, %2\n ds_bpermute_b32 %3, %9, %3\n"
                      "ds_bpermute_b32 %4, %9, %4\n ds_bpermute_b32 %5, %9, %5\n ds_bpermute_b32 %6, %9, %6\n ds_bpermute_b32 %7, %9, %7\n s_waitcnt lgkmcnt(0)\n")

// 64-bit shift: accumulators in pairs
__global__ __launch_bounds__(256) void k_lshlrev_b64(Stamp* st, unsigned* sink, unsigned seed)
{
    unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 17;
    unsigned b = seed & 1u;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#define S8 "v_lshlrev_b64 %0, %8, %0\n v_lshlrev_b64 %1, %8, %1\n v_lshlrev_b64 %2, %8, %2\n v_lshlrev_b64 %3, %8, %3\n" \
           "v_lshlrev_b64 %4, %8, %4\n v_lshlrev_b64 %5, %8, %5\n v_lshlrev_b64 %6, %8, %6\n v_lshlrev_b64 %7, %8, %7\n"
        asm volatile(S8 S8 S8 S8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        Stamp s = { t0, t1 };
        st[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
// scalar ALU (one scalar unit per CU, shared by the four SIMDs)
__global__ __launch_bounds__(256) void k_s_add_u32(Stamp* st, unsigned* sink, unsigned seed)
{
    unsigned a0 = seed, a1 = seed * 3, a2 = seed * 5, a3 = seed * 7;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#define SA "s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n"
        asm volatile(SA SA SA SA : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3) : "s"(seed) : "scc");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        Stamp s = { t0, t1 };
        st[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
// two classes interleaved: does a VALU instruction issue beside an LDS / a scalar one of ANOTHER wavefront? (mix of whole wavefronts:
// even workgroups run class A, odd ones class B; reported per class)

typedef void (*Kern)(Stamp*, unsigned*, unsigned);
struct Case { const char* name; Kern k; };

int main(int argc, char** argv)
{
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    int clk_khz = 0, wall_khz = 0;
    CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, dev));
    CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
    printf("# %s, %d CUs, clockRate %d kHz, wallClockRate %d kHz; %d x %d instructions per wavefront\n", p.name, cus, clk_khz, wall_khz, ITER, GROUP);
    printf("# columns: class, wavefronts per SIMD, ms per launch, wave instructions per SIMD per microsecond, cycles per wave instruction per SIMD at the\n"
           "#          nominal clock (clockRate), the same by the wavefronts' own s_memtime spans (ticks per instruction: their unit is what is in doubt)\n");
    const Case cases[] = {
        { "v_add_u32", k_add_u32 }, { "v_xor_b32", k_xor_b32 }, { "v_lshlrev_b32", k_lshlrev_b32 }, { "v_pk_add_u16", k_pk_add_u16 },
        { "v_pk_lshlrev_b16", k_pk_lshlrev_b16 }, { "v_perm_b32", k_perm_b32 }, { "v_alignbit_b32", k_alignbit_b32 }, { "v_bfe_u32", k_bfe_u32 },
        { "v_lshl_or_b32", k_lshl_or_b32 }, { "v_mad_u32_u24", k_mad_u32_u24 }, { "v_mul_lo_u32", k_mul_lo_u32 }, { "v_bcnt_u32_b32", k_bcnt },
        { "v_cndmask_b32 (vcc never written)", k_cndmask }, { "sub_u32", k_sub_u32 }, { "and_b32", k_and_b32 }, { "or_b32", k_or_b32 }, { "min_u32", k_min_u32 }, { "lshrrev_b32", k_lshrrev_b32 }, { "ashrrev_i32", k_ashrrev_i32 }, { "mul_u32_u24", k_mul_u32_u24 }, { "add3_u32", k_add3_u32 }, { "and_or_b32", k_and_or_b32 }, { "or3_b32", k_or3_b32 }, { "xad_u32", k_xad_u32 }, { "lshl_add_u32", k_lshl_add_u32 }, { "add_lshl_u32", k_add_lshl_u32 }, { "alignbyte_b32", k_alignbyte_b32 }, { "pk_sub_i16", k_pk_sub_i16 }, { "pk_ashrrev_i16", k_pk_ashrrev_i16 }, { "pk_min_u16", k_pk_min_u16 }, { "mbcnt_lo", k_mbcnt_lo }, { "mov_b32", k_mov_b32 }, { "bitop3", k_bitop3 }, { "cmp + vcc", k_cmp_vcc }, { "cmp + cndmask", k_cmp_cndmask }, { "cmp + sgpr + cndmask", k_cmp_sgpr_cndmask }, { "cmp + addc", k_cmp_addc }, { "readlane", k_readlane }, { "ds_bpermute", k_ds_bpermute },  { "v_mov_b32_dpp row_shr:1", k_mov_dpp }, { "v_lshlrev_b64", k_lshlrev_b64 }, { "s_add_u32", k_s_add_u32 },
        { "ds_write_b32", k_ds_write_b32 }, { "ds_write_b8", k_ds_write_b8 }, { "ds_or_b32", k_ds_or_b32 }, { "ds_add_u32", k_ds_add_u32 },
        { "ds_read_b32", k_ds_read_b32 }, { "ds_read_u8", k_ds_read_u8 },
    };
    const int maxw = 8;
    Stamp* st;
    unsigned* sink;
    CHECK(hipMalloc(&st, sizeof(Stamp) * cus * maxw * 4));
    CHECK(hipMalloc(&sink, 4 * cus * maxw * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Stamp> h(cus * maxw * 4);
    for (const Case& c : cases) {
        for (int w : { 1, 2, 4, 8 }) {
            const int grid = cus * w;
            float best = 1e30f;
            double ticks = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(c.k, dim3(grid), dim3(256), 0, 0, st, sink, 12345u + rep);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 0) continue;
                if (ms < best) {
                    best = ms;
                    CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * grid * 4, hipMemcpyDeviceToHost));
                    // median span of a wavefront
                    std::vector<unsigned long long> sp;
                    for (int i = 0; i < grid * 4; ++i) sp.push_back(h[i].t1 - h[i].t0);
                    std::sort(sp.begin(), sp.end());
                    ticks = (double)sp[sp.size() / 2];
                }
            }
            const double instr_per_simd = (double)w * ITER * GROUP;
            const double us = best * 1e3;
            printf("%-26s %d  %8.4f ms  %8.1f /us  %6.2f cycles  %6.2f ticks (median wavefront: %.0f ticks for %d instructions, x %d wavefronts)\n", c.name, w, best,
                   instr_per_simd / us, us * (clk_khz / 1e3) / instr_per_simd, ticks / (ITER * GROUP) / w * 1.0, ticks, ITER * GROUP, w);
        }
    }
    return 0;
}
