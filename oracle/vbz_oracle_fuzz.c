/*
 * vbz_oracle_fuzz.c -- CPU ORACLE (test infrastructure): replay of the reference's fuzz target over one input.
 *
 * Restates the decompress half of LLVMFuzzerTestOneInput (reference vbz/fuzzing/vbz_fuzz.cpp:138-161): for one option
 * set, find the smallest power of two p <= 1 MiB whose vbz_max_compressed_size(p) exceeds the input size -- an ERROR
 * value counts as "exceeds", exactly as the unsigned comparison there does -- and call vbz_decompress and
 * vbz_decompress_sized with every destination size 0..p.  The reference only requires "no crash"; the oracle's
 * verdicts (bytes produced or error code) are what the GPU path is compared with (tests/test_gpu_parity.py).
 */
#include <stdlib.h>
#include <string.h>

#include "vbz_oracle.h"

/* vbz_fuzz.cpp:145-154 */
uint32_t vbo_fuzz_max_destination(uint32_t size, const VboOptions* o)
{
    for (uint64_t cand = 1; cand <= 1024u * 1024u; cand *= 2) {
        const vbo_size_t m = vbo_max_compressed_size((vbo_size_t)cand, o);
        if (m > size) return (uint32_t)cand;
    }
    return 0;
}

/* vbz_fuzz.cpp:101-136,156-160.  results[2*g] = vbo_decompress(data, size, dst, g), results[2*g+1] =
 * vbo_decompress_sized(data, size, dst, g) for g = 0..max_destination.  Returns 0, or -1 if out of memory. */
int vbo_fuzz_decompress_sweep(const void* data, uint32_t size, const VboOptions* o, uint32_t max_destination, uint32_t* results)
{
    uint8_t* dst = (uint8_t*)malloc((size_t)max_destination + 64);
    if (!dst) return -1;
    for (uint32_t g = 0; g <= max_destination; ++g) {
        results[2 * g] = vbo_decompress(data, size, dst, g, o);
        results[2 * g + 1] = vbo_decompress_sized(data, size, dst, g, o);
    }
    free(dst);
    return 0;
}
