// Host harness (tests only): exposes the serial table-construction functions of
// vbz_compression_amd/csrc/zstd_entropy.h to ctypes so the CPU suite can check them against
// what libzstd emits for the same histogram.  Built by tests/test_entropy_host.py with g++.
#include <cstring>
#include "zstd_entropy.h"

using namespace vbzhip;

extern "C" {

int h_huf_build(const uint32_t* count, uint32_t maxSymbolValue, uint32_t maxNbBits, uint8_t* nbBits, uint16_t* code)
{
    static HufBuildWksp w;
    return (int)huf_build(count, maxSymbolValue, maxNbBits, nbBits, code, &w);
}

uint32_t h_optimal_table_log(uint32_t maxTableLog, uint32_t srcSize, uint32_t maxSymbolValue, uint32_t minus)
{
    return optimal_table_log(maxTableLog, srcSize, maxSymbolValue, minus);
}

int h_huf_write_tree(uint8_t* dst, int cap, const uint8_t* nbBits, uint32_t maxSymbolValue, uint32_t huffLog)
{
    static FseWeightWksp w;
    uint8_t weights[260];
    return huf_write_tree(dst, cap, nbBits, maxSymbolValue, huffLog, weights, &w);
}
}
