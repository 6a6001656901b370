// zstd_entropy.h -- serial statements of the table construction of the zstd-format entropy stage.
//
// Plain integer functions over caller-provided workspaces, compiled for gfx950 and under g++ (the CPU unit tests:
// tests/test_entropy_host.py).  The device encoder runs wave-parallel forms of them (zstd_encode.hip: huf_build_wave,
// huf_write_tree_wave) that must give the same bytes; the rare cases they leave alone fall back to these on one lane.
//
//   huf_build_pm / huf_package_merge   code lengths: optimal under the length limit (package-merge) -- what the device builds
//   (libzstd's own construction, the yardstick of the CPU tests, lives in tests/host/zstd_reference_huffman.h: test infrastructure)
//   huf_write_tree, fse_*              Huffman tree description (weights, FSE-compressed or direct): libzstd's HUF_writeCTable
//                                      byte for byte (facebook/zstd lib/compress/huf_compress.c, fse_compress.c, the
//                                      library the reference links as zstd/1.4.8: CMakeLists.txt:92-93)
//   Seq*                               predefined LL / ML encoding tables and value -> code maps of the sequences section
//
// The zstd format is RFC 8878.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define VBZ_HD __host__ __device__ __forceinline__
#define VBZ_HDN __host__ __device__
#else
#define VBZ_HD inline
#define VBZ_HDN inline
#endif

namespace vbzhip {

#ifndef VBZ_HUF_MAX_BITS
#define VBZ_HUF_MAX_BITS 11
#endif
constexpr int HUF_MAX_BITS = VBZ_HUF_MAX_BITS;      // zstd's HUF_TABLELOG_DEFAULT: literal codes are at most 11 bits
constexpr int HUF_ABS_MAX_BITS = 12;  // HUF_TABLELOG_MAX
constexpr uint32_t HUF_NO_SYMBOL = 0xF0F0F0F0u;

VBZ_HD int hb32(uint32_t v)  // index of the highest set bit, v != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 31 - __clz((int)v);
#else
    return 31 - __builtin_clz(v);
#endif
}

// The accuracy log libzstd picks for a table over `srcSize` symbols with alphabet 0 .. maxSymbolValue (its FSE_optimalTableLog rule, RFC 8878
// leaves the choice to the encoder): the asked-for log (0 = 11), but no more than the source can fill (log2(srcSize - 1) - minus) and no
// less than the alphabet needs (the smaller of log2(srcSize) + 1 and log2(maxSymbolValue) + 2), inside [5, 12].  The tree description's
// bytes depend on it, and those are held to libzstd's byte for byte (tests/test_entropy_host.py).
VBZ_HD uint32_t optimal_table_log(uint32_t maxTableLog, uint32_t srcSize, uint32_t maxSymbolValue, uint32_t minus)
{
    const uint32_t asked = maxTableLog ? maxTableLog : 11u;
    const uint32_t fill = (uint32_t)hb32(srcSize - 1) - minus;
    const uint32_t by_size = (uint32_t)hb32(srcSize) + 1u, by_alphabet = (uint32_t)hb32(maxSymbolValue) + 2u;
    const uint32_t floor_ = by_size < by_alphabet ? by_size : by_alphabet;
    uint32_t log = asked < fill ? asked : fill;
    log = log > floor_ ? log : floor_;
    return log < 5u ? 5u : (log > 12u ? 12u : log);
}

// ------------------------------------------------------------------------------------------------
// Optimal length-limited code lengths by package-merge (Larmore & Hirschberg 1990), in the form the device
// encoder runs across a wavefront (zstd_encode.hip: huf_build_wave) -- this is its serial statement.
//   w[0..n) ascending, n >= 2, 2^L >= n.  len[i] = code length of w[i], sum of 2^-len[i] == 1, len[i] <= L,
//   sum of w[i] * len[i] minimal under that limit (libzstd builds the unlimited tree and repairs it with a
//   heuristic, HUF_setMaxHeight: never shorter than this, sometimes longer).
// Level 1 holds the leaves; level j holds the leaves merged with the packages (sums of consecutive pairs) of
// level j-1, cut to the first 2n-2 items, a leaf in front of a package of the same weight.  The code takes the
// first 2n-2 items of level L; a package taken at level j takes its two items at level j-1; a leaf's length is
// the number of levels it is taken at.  Since every list is sorted, "taken" is a prefix of the leaves at every
// level: all that has to be kept per level is which of its items are packages.
struct HufPmWksp
{
    uint32_t leaf[256];     // weights of the present symbols, ascending (padded with 0xFFFFFFFF)
    uint32_t pkg[256];      // packages of the previous level, ascending (padded with 0xFFFFFFFF)
    uint32_t merged[512];   // the current level's list
    uint32_t isPkg[HUF_MAX_BITS + 1][16];  // per level: bit p = item p of the list is a package
    uint8_t sym[256];       // symbol of leaf i
};

VBZ_HDN void huf_package_merge(HufPmWksp* k, uint32_t n, uint32_t L, uint8_t* len)
{
    const uint32_t X0 = 2 * n - 2;
    uint32_t m = n >> 1;
    for (uint32_t t = 0; t < 256; ++t) k->pkg[t] = t < m ? k->leaf[2 * t] + k->leaf[2 * t + 1] : 0xFFFFFFFFu;
    for (uint32_t j = 2; j <= L; ++j) {
        for (int q = 0; q < 16; ++q) k->isPkg[j][q] = 0;
        uint32_t a = 0, b = 0, p = 0;
        while (p < X0 && (a < n || b < m)) {
            const bool takeLeaf = a < n && (b >= m || k->leaf[a] <= k->pkg[b]);
            if (takeLeaf) k->merged[p] = k->leaf[a++];
            else {
                k->merged[p] = k->pkg[b++];
                k->isPkg[j][p >> 5] |= 1u << (p & 31);
            }
            ++p;
        }
        m = p >> 1;
        for (uint32_t t = 0; t < 256; ++t) k->pkg[t] = t < m ? k->merged[2 * t] + k->merged[2 * t + 1] : 0xFFFFFFFFu;
    }
    for (uint32_t i = 0; i < n; ++i) len[i] = 0;
    uint32_t X = X0;
    for (uint32_t j = L; j >= 2 && X; --j) {
        uint32_t pk = 0;
        for (uint32_t p = 0; p < X; ++p) pk += (k->isPkg[j][p >> 5] >> (p & 31)) & 1u;
        const uint32_t nl = X - pk;
        for (uint32_t i = 0; i < nl; ++i) len[i]++;
        X = 2 * pk;
    }
    for (uint32_t i = 0; i < X; ++i) len[i]++;   // level 1: leaves only
}

// Length-limited canonical Huffman code from a histogram, lengths by package-merge, codes numbered the way zstd
// numbers them (HUF_buildCTable_wksp: within a length by increasing symbol, longest codes lowest).
//   count[0..maxSymbolValue], at least two non-zero counts, every count < 2^24.  Returns the table log.
// Order of equal counts: the higher symbol sorts first (gets the longer code).
VBZ_HDN uint32_t huf_build_pm(const uint32_t* count, uint32_t maxSymbolValue, uint32_t maxNbBits, uint8_t* nbBits, uint16_t* code,
                              HufPmWksp* k)
{
    uint32_t n = 0;
    for (uint32_t s = 0; s <= maxSymbolValue; ++s) {   // insertion sort of (count << 8 | 255 - s), ascending
        if (!count[s]) continue;
        const uint32_t key = (count[s] << 8) | (255u - s);
        uint32_t pos = n++;
        while (pos > 0 && (((k->leaf[pos - 1] << 8) | (255u - k->sym[pos - 1])) > key)) {
            k->leaf[pos] = k->leaf[pos - 1];
            k->sym[pos] = k->sym[pos - 1];
            --pos;
        }
        k->leaf[pos] = count[s];
        k->sym[pos] = (uint8_t)s;
    }
    for (uint32_t i = n; i < 256; ++i) k->leaf[i] = 0xFFFFFFFFu;
    uint8_t len[256];
    huf_package_merge(k, n, maxNbBits, len);
    maxNbBits = len[0];
    uint16_t nbPerRank[HUF_ABS_MAX_BITS + 2], valPerRank[HUF_ABS_MAX_BITS + 2];
    for (int i = 0; i < HUF_ABS_MAX_BITS + 2; ++i) nbPerRank[i] = valPerRank[i] = 0;
    for (uint32_t s = 0; s <= maxSymbolValue; ++s) nbBits[s] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        nbBits[k->sym[i]] = len[i];
        nbPerRank[len[i]]++;
    }
    {
        uint16_t min = 0;
        for (int r = (int)maxNbBits; r > 0; r--) {
            valPerRank[r] = min;
            min = (uint16_t)(min + nbPerRank[r]);
            min >>= 1;
        }
    }
    for (uint32_t s = 0; s <= maxSymbolValue; ++s) code[s] = nbBits[s] ? valPerRank[nbBits[s]]++ : 0;
    return maxNbBits;
}

// ------------------------------------------------------------------------------------------------
// FSE compression of the Huffman weights (zstd HUF_compressWeights / FSE_normalizeCount /
// FSE_writeNCount / FSE_buildCTable / FSE_compress_usingCTable) -- alphabet <= 13, tableLog <= 6
// ------------------------------------------------------------------------------------------------
struct FseWeightWksp
{
    uint32_t count[16];
    int16_t norm[16];
    uint16_t stateTable[64];
    uint8_t tableSymbol[64];
    uint32_t cumul[18];
    int32_t deltaFindState[16];
    uint32_t deltaNbBits[16];
    // the device's wave-cooperative writer (zstd_encode.hip: huf_write_tree_wave) only:
    uint16_t emit[256];    // what each step of the two state chains puts out: bits | count << 8
    uint32_t bitbuf[52];   // the packed bit stream (at most 253 x 6 + 13 bits)
    int32_t verdict;
};

struct BitW  // forward bit writer (little-endian), zstd BIT_CStream_t
{
    uint64_t acc;
    uint32_t nbits;
    uint8_t* p;
    uint8_t* end;
};

VBZ_HD void bitw_add(BitW& b, uint32_t v, uint32_t n)
{
    b.acc |= (uint64_t)(v & ((1u << n) - 1u)) << b.nbits;
    b.nbits += n;
}

VBZ_HD void bitw_flush(BitW& b)
{
    while (b.nbits >= 8 && b.p < b.end) {
        *b.p++ = (uint8_t)b.acc;
        b.acc >>= 8;
        b.nbits -= 8;
    }
}

VBZ_HDN int fse_normalize_m2(int16_t* norm, uint32_t tableLog, const uint32_t* count, uint32_t total, uint32_t maxSymbolValue,
                            int16_t lowProbCount)
{
    const int16_t NOT_YET_ASSIGNED = -2;
    uint32_t s, distributed = 0, ToDistribute;
    const uint32_t lowThreshold = total >> tableLog;
    uint32_t lowOne = (total * 3) >> (tableLog + 1);
    for (s = 0; s <= maxSymbolValue; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NOT_YET_ASSIGNED;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0) return 0;
    if ((total / ToDistribute) > lowOne) {
        lowOne = (total * 3) / (ToDistribute * 2);
        for (s = 0; s <= maxSymbolValue; s++) {
            if ((norm[s] == NOT_YET_ASSIGNED) && (count[s] <= lowOne)) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        }
        ToDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSymbolValue + 1) {
        uint32_t maxV = 0, maxC = 0;
        for (s = 0; s <= maxSymbolValue; s++)
            if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] = (int16_t)(norm[maxV] + (int16_t)ToDistribute);
        return 0;
    }
    if (total == 0) {
        for (s = 0; ToDistribute > 0; s = (s + 1) % (maxSymbolValue + 1))
            if (norm[s] > 0) { ToDistribute--; norm[s]++; }
        return 0;
    }
    {
        const uint64_t vStepLog = 62 - tableLog;
        const uint64_t mid = (1ull << (vStepLog - 1)) - 1;
        const uint64_t rStep = ((((uint64_t)1 << vStepLog) * ToDistribute) + mid) / total;
        uint64_t tmpTotal = mid;
        for (s = 0; s <= maxSymbolValue; s++) {
            if (norm[s] == NOT_YET_ASSIGNED) {
                const uint64_t end = tmpTotal + ((uint64_t)count[s] * rStep);
                const uint32_t sStart = (uint32_t)(tmpTotal >> vStepLog);
                const uint32_t sEnd = (uint32_t)(end >> vStepLog);
                const uint32_t weight = sEnd - sStart;
                if (weight < 1) return -1;
                norm[s] = (int16_t)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

// returns tableLog, 0 for "rle" (one symbol holds everything), -1 on failure.
// lowProbCount: -1 ("less than one" cells) or 1; zstd >= 1.4.7 uses 1 for Huffman weights.
VBZ_HDN int fse_normalize(int16_t* norm, uint32_t tableLog, const uint32_t* count, uint32_t total, uint32_t maxSymbolValue,
                         int16_t lowProbCount)
{
    const uint32_t rtbTable[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    const uint64_t scale = 62 - tableLog;
    const uint64_t step = ((uint64_t)1 << 62) / total;
    const uint64_t vStep = 1ull << (scale - 20);
    int stillToDistribute = 1 << tableLog;
    uint32_t s, largest = 0;
    int16_t largestP = 0;
    const uint32_t lowThreshold = total >> tableLog;
    for (s = 0; s <= maxSymbolValue; s++) {
        if (count[s] == total) return 0;
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) {
            norm[s] = lowProbCount;
            stillToDistribute--;
        } else {
            int16_t proba = (int16_t)((count[s] * step) >> scale);
            if (proba < 8) {
                uint64_t restToBeat = vStep * rtbTable[proba];
                proba = (int16_t)(proba + (((count[s] * step) - ((uint64_t)proba << scale)) > restToBeat));
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba;
            stillToDistribute -= proba;
        }
    }
    if (-stillToDistribute >= (norm[largest] >> 1)) {
        if (fse_normalize_m2(norm, tableLog, count, total, maxSymbolValue, lowProbCount) != 0) return -1;
    } else {
        norm[largest] = (int16_t)(norm[largest] + (int16_t)stillToDistribute);
    }
    return (int)tableLog;
}

// FSE table description (zstd FSE_writeNCount); returns bytes written or -1
VBZ_HDN int fse_write_ncount(uint8_t* out, int cap, const int16_t* norm, uint32_t maxSymbolValue, uint32_t tableLog)
{
    uint8_t* const ostart = out;
    uint8_t* const oend = out + cap;
    const int tableSize = 1 << tableLog;
    uint32_t bitStream = 0;
    int bitCount = 0;
    uint32_t symbol = 0;
    const uint32_t alphabetSize = maxSymbolValue + 1;
    int previousIs0 = 0;
    bitStream += (tableLog - 5) << bitCount;
    bitCount += 4;
    int remaining = tableSize + 1;
    int threshold = tableSize;
    int nbBits = (int)tableLog + 1;
    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            uint32_t start = symbol;
            while ((symbol < alphabetSize) && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24;
                bitStream += 0xFFFFu << bitCount;
                if (out + 2 > oend) return -1;
                out[0] = (uint8_t)bitStream;
                out[1] = (uint8_t)(bitStream >> 8);
                out += 2;
                bitStream >>= 16;
            }
            while (symbol >= start + 3) {
                start += 3;
                bitStream += 3u << bitCount;
                bitCount += 2;
            }
            bitStream += (symbol - start) << bitCount;
            bitCount += 2;
            if (bitCount > 16) {
                if (out + 2 > oend) return -1;
                out[0] = (uint8_t)bitStream;
                out[1] = (uint8_t)(bitStream >> 8);
                out += 2;
                bitStream >>= 16;
                bitCount -= 16;
            }
        }
        {
            int count = norm[symbol++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (uint32_t)count << bitCount;
            bitCount += nbBits;
            bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return -1;
            while (remaining < threshold) {
                nbBits--;
                threshold >>= 1;
            }
        }
        if (bitCount > 16) {
            if (out + 2 > oend) return -1;
            out[0] = (uint8_t)bitStream;
            out[1] = (uint8_t)(bitStream >> 8);
            out += 2;
            bitStream >>= 16;
            bitCount -= 16;
        }
    }
    if (remaining != 1) return -1;
    if (out + 2 > oend) return -1;
    out[0] = (uint8_t)bitStream;
    out[1] = (uint8_t)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (int)(out - ostart);
}

// ------------------------------------------------------------------------------------------------
// Generic FSE encoding table (zstd FSE_buildCTable_wksp) for a normalized distribution.
//   stateTable[1 << tableLog], deltaNbBits/deltaFindState[maxSymbolValue + 1]
//   scratch: tableSymbol[1 << tableLog], cumul[maxSymbolValue + 2]
// ------------------------------------------------------------------------------------------------
VBZ_HDN void fse_build_ctable(const int16_t* norm, uint32_t maxSymbolValue, uint32_t tableLog, uint16_t* stateTable,
                              uint32_t* deltaNbBits, int32_t* deltaFindState, uint8_t* tableSymbol, uint32_t* cumul)
{
    const uint32_t tableSize = 1u << tableLog, tableMask = tableSize - 1;
    const uint32_t step = (tableSize >> 1) + (tableSize >> 3) + 3;
    uint32_t highThreshold = tableSize - 1;
    cumul[0] = 0;
    for (uint32_t u = 1; u <= maxSymbolValue + 1; u++) {
        if (norm[u - 1] == -1) {
            cumul[u] = cumul[u - 1] + 1;
            tableSymbol[highThreshold--] = (uint8_t)(u - 1);
        } else {
            cumul[u] = cumul[u - 1] + (uint32_t)norm[u - 1];
        }
    }
    cumul[maxSymbolValue + 1] = tableSize + 1;
    {
        uint32_t position = 0;
        for (uint32_t symbol = 0; symbol <= maxSymbolValue; symbol++) {
            const int freq = norm[symbol];
            for (int occ = 0; occ < freq; occ++) {
                tableSymbol[position] = (uint8_t)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    for (uint32_t u = 0; u < tableSize; u++) {
        const uint8_t s = tableSymbol[u];
        stateTable[cumul[s]++] = (uint16_t)(tableSize + u);
    }
    uint32_t total = 0;
    for (uint32_t s = 0; s <= maxSymbolValue; s++) {
        const int nc = norm[s];
        if (nc == 0) {
            deltaNbBits[s] = ((tableLog + 1) << 16) - (1u << tableLog);
            deltaFindState[s] = 0;
        } else if (nc == -1 || nc == 1) {
            deltaNbBits[s] = (tableLog << 16) - (1u << tableLog);
            deltaFindState[s] = (int32_t)total - 1;
            total++;
        } else {
            const uint32_t maxBitsOut = tableLog - (uint32_t)hb32((uint32_t)nc - 1);
            const uint32_t minStatePlus = (uint32_t)nc << maxBitsOut;
            deltaNbBits[s] = (maxBitsOut << 16) - minStatePlus;
            deltaFindState[s] = (int32_t)total - nc;
            total += (uint32_t)nc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Sequences (RFC 8878 3.1.1.3.2): predefined distributions and value -> code maps
// ------------------------------------------------------------------------------------------------
constexpr int SEQ_LL_SYMS = 36, SEQ_ML_SYMS = 53, SEQ_DEF_LOG = 6;

struct SeqCTables  // encoding tables of the predefined literal-length / match-length distributions
{
    uint16_t ll_state[64];
    uint16_t ml_state[64];
    uint32_t ll_dnb[SEQ_LL_SYMS];
    int32_t ll_dfs[SEQ_LL_SYMS];
    uint32_t ml_dnb[SEQ_ML_SYMS];
    int32_t ml_dfs[SEQ_ML_SYMS];
};

VBZ_HDN void seq_default_norms(int16_t* ll, int16_t* ml)
{
    const int16_t LLD[36] = { 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1 };
    const int16_t MLD[53] = { 1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                              1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1 };
    for (int i = 0; i < 36; ++i) ll[i] = LLD[i];
    for (int i = 0; i < 53; ++i) ml[i] = MLD[i];
}

VBZ_HDN void seq_build_default_ctables(SeqCTables* t)
{
    int16_t ll[36], ml[53];
    uint8_t sym[64];
    uint32_t cumul[56];
    seq_default_norms(ll, ml);
    fse_build_ctable(ll, 35, SEQ_DEF_LOG, t->ll_state, t->ll_dnb, t->ll_dfs, sym, cumul);
    fse_build_ctable(ml, 52, SEQ_DEF_LOG, t->ml_state, t->ml_dnb, t->ml_dfs, sym, cumul);
}

// literal length -> (code, extra bits, number of extra bits)
VBZ_HD void seq_ll_code(uint32_t v, uint32_t* code, uint32_t* extra, uint32_t* nbits)
{
    if (v < 16) { *code = v; *extra = 0; *nbits = 0; return; }
    uint32_t c, base, nb;
    if (v < 24) { c = 16 + ((v - 16) >> 1); nb = 1; base = 16 + ((v - 16) & ~1u); }
    else if (v < 32) { c = 20 + ((v - 24) >> 2); nb = 2; base = 24 + ((v - 24) & ~3u); }
    else if (v < 48) { c = 22 + ((v - 32) >> 3); nb = 3; base = 32 + ((v - 32) & ~7u); }
    else if (v < 64) { c = 24; nb = 4; base = 48; }
    else { const uint32_t h = (uint32_t)hb32(v); c = h + 19; nb = h; base = 1u << h; }
    *code = c; *extra = v - base; *nbits = nb;
}

// match length (>= 3) -> (code, extra bits, number of extra bits)
VBZ_HD void seq_ml_code(uint32_t ml, uint32_t* code, uint32_t* extra, uint32_t* nbits)
{
    const uint32_t v = ml - 3;
    if (v < 32) { *code = v; *extra = 0; *nbits = 0; return; }
    uint32_t c, base, nb;
    if (v < 40) { c = 32 + ((v - 32) >> 1); nb = 1; base = 32 + ((v - 32) & ~1u); }
    else if (v < 48) { c = 36 + ((v - 40) >> 2); nb = 2; base = 40 + ((v - 40) & ~3u); }
    else if (v < 64) { c = 38 + ((v - 48) >> 3); nb = 3; base = 48 + ((v - 48) & ~7u); }
    else if (v < 96) { c = 40 + ((v - 64) >> 4); nb = 4; base = 64 + ((v - 64) & ~15u); }
    else if (v < 128) { c = 42; nb = 5; base = 96; }
    else { const uint32_t h = (uint32_t)hb32(v); c = h + 36; nb = h; base = 1u << h; }
    *code = c; *extra = v - base; *nbits = nb;
}

// FSE-compress the weight list (zstd HUF_compressWeights). Returns bytes, 0 = not compressible,
// 1 = single repeated weight (caller falls back), -1 = error.
// `counted`: w->count[0..12] already holds the histogram of the weights (the wave fills it in parallel)
VBZ_HDN int huf_compress_weights(uint8_t* dst, int cap, const uint8_t* weights, uint32_t wtSize, FseWeightWksp* w,
                                 bool counted = false)
{
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    uint32_t maxSymbolValue = HUF_ABS_MAX_BITS;
    if (wtSize <= 1) return 0;
    uint32_t maxCount = 0;
    if (!counted) {
        for (int i = 0; i < 16; ++i) w->count[i] = 0;
        for (uint32_t i = 0; i < wtSize; ++i) w->count[weights[i]]++;
    }
    while (w->count[maxSymbolValue] == 0) maxSymbolValue--;
    for (uint32_t s = 0; s <= maxSymbolValue; ++s)
        if (w->count[s] > maxCount) maxCount = w->count[s];
    if (maxCount == wtSize) return 1;
    if (maxCount == 1) return 0;
    uint32_t tableLog = optimal_table_log(6, wtSize, maxSymbolValue, 2);
    if (fse_normalize(w->norm, tableLog, w->count, wtSize, maxSymbolValue, 1) <= 0) return -1;
    {
        int h = fse_write_ncount(op, (int)(oend - op), w->norm, maxSymbolValue, tableLog);
        if (h < 0) return -1;
        op += h;
    }
    fse_build_ctable(w->norm, maxSymbolValue, tableLog, w->stateTable, w->deltaNbBits, w->deltaFindState, w->tableSymbol, w->cumul);
    // FSE_compress_usingCTable: two interleaved states, symbols consumed from the end
    BitW bw;
    bw.acc = 0;
    bw.nbits = 0;
    bw.p = op;
    bw.end = oend;
    const uint8_t* ip = weights + wtSize;
    uint32_t st1, st2;
#define FSE_INIT2(st, sym)                                                              \
    do {                                                                                \
        uint32_t nbo = (w->deltaNbBits[sym] + (1u << 15)) >> 16;                        \
        uint32_t v = (nbo << 16) - w->deltaNbBits[sym];                                 \
        st = w->stateTable[(int32_t)(v >> nbo) + w->deltaFindState[sym]];               \
    } while (0)
#define FSE_ENC(st, sym)                                                                \
    do {                                                                                \
        uint32_t nbo = (st + w->deltaNbBits[sym]) >> 16;                                \
        bitw_add(bw, st, nbo);                                                          \
        st = w->stateTable[(int32_t)(st >> nbo) + w->deltaFindState[sym]];              \
    } while (0)
    if (wtSize & 1) {
        --ip; FSE_INIT2(st1, *ip);
        --ip; FSE_INIT2(st2, *ip);
        --ip; FSE_ENC(st1, *ip);
        bitw_flush(bw);
    } else {
        --ip; FSE_INIT2(st2, *ip);
        --ip; FSE_INIT2(st1, *ip);
    }
    while (ip > weights) {
        --ip; FSE_ENC(st2, *ip);
        --ip; FSE_ENC(st1, *ip);
        bitw_flush(bw);
    }
#undef FSE_INIT2
#undef FSE_ENC
    bitw_add(bw, st2, tableLog);
    bitw_flush(bw);
    bitw_add(bw, st1, tableLog);
    bitw_flush(bw);
    bitw_add(bw, 1, 1);  // end mark
    bitw_flush(bw);
    if (bw.nbits > 0) {
        if (bw.p >= bw.end) return -1;
        *bw.p++ = (uint8_t)bw.acc;
    }
    if (bw.p >= bw.end) return -1;  // zstd: BIT_closeCStream reports overflow when the buffer filled up
    op = bw.p;
    return (int)(op - dst);
}

// Huffman tree description (zstd HUF_writeCTable). nbBits[0..maxSymbolValue], huffLog = table log.
// Returns bytes written (<= 129) or -1.  `weights` is a 256-byte scratch.
// `prepared`: weights[0..maxSymbolValue) and w->count[] were already filled by the caller.
VBZ_HDN int huf_write_tree(uint8_t* dst, int cap, const uint8_t* nbBits, uint32_t maxSymbolValue, uint32_t huffLog,
                           uint8_t* weights, FseWeightWksp* w, bool prepared = false)
{
    if (cap < 1) return -1;
    if (!prepared)
        for (uint32_t n = 0; n < maxSymbolValue; n++) weights[n] = nbBits[n] ? (uint8_t)(huffLog + 1 - nbBits[n]) : 0;
    {
        int hSize = huf_compress_weights(dst + 1, cap - 1, weights, maxSymbolValue, w, prepared);
        if (hSize < 0) return -1;
        if ((hSize > 1) && ((uint32_t)hSize < maxSymbolValue / 2)) {
            dst[0] = (uint8_t)hSize;
            return hSize + 1;
        }
    }
    if (maxSymbolValue > 128) return -1;  // cannot be described raw: caller stores the block uncompressed
    if ((int)((maxSymbolValue + 1) / 2) + 1 > cap) return -1;
    dst[0] = (uint8_t)(128 + (maxSymbolValue - 1));
    weights[maxSymbolValue] = 0;
    for (uint32_t n = 0; n < maxSymbolValue; n += 2) dst[(n / 2) + 1] = (uint8_t)((weights[n] << 4) + weights[n + 1]);
    return (int)((maxSymbolValue + 1) / 2) + 1;
}

}  // namespace vbzhip
