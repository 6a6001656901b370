// Host harness (tests only): exposes the serial table-construction functions of
// vbz_compression_amd/csrc/zstd_entropy.h to ctypes so the CPU suite can check them against
// what libzstd emits for the same histogram.  Built by tests/test_entropy_host.py with g++.
#include <cstring>
#include "zstd_entropy.h"
#include "zstd_reference_huffman.h"   // libzstd's construction: the yardstick (BSD notice inside)

using namespace vbzhip;

extern "C" {

int h_huf_build(const uint32_t* count, uint32_t maxSymbolValue, uint32_t maxNbBits, uint8_t* nbBits, uint16_t* code)
{
    static HufBuildWksp w;
    return (int)huf_build(count, maxSymbolValue, maxNbBits, nbBits, code, &w);
}

int h_huf_build_pm(const uint32_t* count, uint32_t maxSymbolValue, uint32_t maxNbBits, uint8_t* nbBits, uint16_t* code)
{
    static HufPmWksp w;
    return (int)huf_build_pm(count, maxSymbolValue, maxNbBits, nbBits, code, &w);
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

// ---- prototype of the "zero-run sequences" block (tests only): the serial statement of what the device
// encoder emits for the control-byte region.  One compressed block: raw literals + sequences whose
// matches are all offset-1 runs (repeat offset 1 of a fresh frame), LL/ML predefined, OF RLE(code 0).
extern "C" int h_encode_zero_run_frame(const uint8_t* k, uint32_t K, uint8_t* out, uint32_t cap, uint32_t rmin)
{
    static SeqCTables T;
    seq_build_default_ctables(&T);
    // tokenise: literals = bytes not in the tail of a zero run of length >= rmin
    static uint8_t lit[1 << 17];
    static uint32_t LL[1 << 14], ML[1 << 14];
    uint32_t nlit = 0, nseq = 0, litsince = 0;
    for (uint32_t p = 0; p < K;) {
        if (k[p] == 0) {
            uint32_t e = p;
            while (e < K && k[e] == 0) ++e;
            if (e - p >= rmin) {
                lit[nlit++] = 0;
                ++litsince;
                LL[nseq] = litsince;
                ML[nseq] = e - p - 1;
                ++nseq;
                litsince = 0;
                p = e;
                continue;
            }
            for (; p < e; ++p) { lit[nlit++] = 0; ++litsince; }
            continue;
        }
        lit[nlit++] = k[p++];
        ++litsince;
    }
    if (nseq == 0 || K > (128u << 10)) return -1;
    uint8_t* op = out;
    // frame header
    const uint32_t magic = 0xFD2FB528u;
    memcpy(op, &magic, 4); op += 4;
    if (K < 256) { *op++ = 0x20; *op++ = (uint8_t)K; }
    else if (K < 65536 + 256) { *op++ = 0x60; uint16_t v = (uint16_t)(K - 256); memcpy(op, &v, 2); op += 2; }
    else { *op++ = 0xA0; memcpy(op, &K, 4); op += 4; }
    uint8_t* bh = op; op += 3;
    // raw literals header
    if (nlit < 32) *op++ = (uint8_t)(nlit << 3);
    else if (nlit < 4096) { *op++ = (uint8_t)((nlit << 4) | 4); *op++ = (uint8_t)(nlit >> 4); }
    else { *op++ = (uint8_t)((nlit << 4) | 12); *op++ = (uint8_t)(nlit >> 4); *op++ = (uint8_t)(nlit >> 12); }
    memcpy(op, lit, nlit); op += nlit;
    // sequences header
    if (nseq < 128) *op++ = (uint8_t)nseq;
    else if (nseq < 0x7F00) { *op++ = (uint8_t)((nseq >> 8) + 128); *op++ = (uint8_t)nseq; }
    else { *op++ = 255; uint16_t v = (uint16_t)(nseq - 0x7F00); memcpy(op, &v, 2); op += 2; }
    *op++ = 0x10;  // LL predefined, OF RLE, ML predefined
    *op++ = 0;     // OF code 0: repeat offset 1 (== 1 in a fresh frame)
    BitW bw; bw.acc = 0; bw.nbits = 0; bw.p = op; bw.end = out + cap;
    uint32_t stLL, stML, c, ex, nb;
    auto init2 = [&](uint32_t& st, const uint16_t* stab, const uint32_t* dnb, const int32_t* dfs, uint32_t sym) {
        uint32_t nbo = (dnb[sym] + (1u << 15)) >> 16;
        uint32_t v = (nbo << 16) - dnb[sym];
        st = stab[(int32_t)(v >> nbo) + dfs[sym]];
    };
    auto enc = [&](uint32_t& st, const uint16_t* stab, const uint32_t* dnb, const int32_t* dfs, uint32_t sym) {
        uint32_t nbo = (st + dnb[sym]) >> 16;
        bitw_add(bw, st, nbo);
        st = stab[(int32_t)(st >> nbo) + dfs[sym]];
    };
    {
        uint32_t lc, lex, lnb, mc, mex, mnb;
        seq_ll_code(LL[nseq - 1], &lc, &lex, &lnb);
        seq_ml_code(ML[nseq - 1], &mc, &mex, &mnb);
        init2(stML, T.ml_state, T.ml_dnb, T.ml_dfs, mc);
        init2(stLL, T.ll_state, T.ll_dnb, T.ll_dfs, lc);
        bitw_add(bw, lex, lnb); bitw_flush(bw);
        bitw_add(bw, mex, mnb); bitw_flush(bw);
    }
    for (int n = (int)nseq - 2; n >= 0; --n) {
        uint32_t lc, lex, lnb, mc, mex, mnb;
        seq_ll_code(LL[n], &lc, &lex, &lnb);
        seq_ml_code(ML[n], &mc, &mex, &mnb);
        enc(stML, T.ml_state, T.ml_dnb, T.ml_dfs, mc);
        enc(stLL, T.ll_state, T.ll_dnb, T.ll_dfs, lc);
        bitw_flush(bw);
        bitw_add(bw, lex, lnb); bitw_flush(bw);
        bitw_add(bw, mex, mnb); bitw_flush(bw);
    }
    (void)c; (void)ex; (void)nb;
    bitw_add(bw, stML, SEQ_DEF_LOG); bitw_flush(bw);
    bitw_add(bw, stLL, SEQ_DEF_LOG); bitw_flush(bw);
    bitw_add(bw, 1, 1); bitw_flush(bw);
    if (bw.nbits > 0) *bw.p++ = (uint8_t)bw.acc;
    op = bw.p;
    const uint32_t bsize = (uint32_t)(op - bh - 3);
    const uint32_t hv = (bsize << 3) | (2u << 1) | 1u;
    bh[0] = (uint8_t)hv; bh[1] = (uint8_t)(hv >> 8); bh[2] = (uint8_t)(hv >> 16);
    return (int)(op - out);
}
