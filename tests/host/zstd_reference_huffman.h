// zstd_reference_huffman.h -- libzstd's Huffman table construction, restated for the CPU tests ONLY (tests/test_entropy_host.py holds the
// product's package-merge construction to it: never longer, same tree description for the same lengths).  Not part of the product: nothing
// under vbz_compression_amd/ includes this file.
//
// These two functions follow facebook/zstd lib/compress/huf_compress.c (HUF_setMaxHeight, HUF_buildCTable_wksp; zstd 1.4.8, the library the
// reference links: CMakeLists.txt:92-93) closely, variable for variable.  zstd is dual-licensed; under the BSD licence:
//
//   BSD License -- For Zstandard software
//   Copyright (c) 2016-present, Facebook, Inc. All rights reserved.
//
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following conditions
//   are met:
//    * Redistributions of source code must retain the above copyright notice, this list of conditions and the following disclaimer.
//    * Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the following disclaimer in
//      the documentation and/or other materials provided with the distribution.
//    * Neither the name Facebook nor the names of its contributors may be used to endorse or promote products derived from this software
//      without specific prior written permission.
//
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT
//   LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED. IN NO EVENT SHALL THE COPYRIGHT
//   HOLDER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT NOT
//   LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND ON
//   ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE
//   USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
#pragma once

#include "zstd_entropy.h"

namespace vbzhip {

struct HufNode
{
    uint32_t count;
    uint16_t parent;
    uint8_t byte;
    uint8_t nbBits;
};

// workspace for huf_build: node[0] is the sentinel in front of the 512-entry node table
struct HufBuildWksp
{
    HufNode node[513];
    uint32_t rankBase[33];
    uint32_t rankCur[33];
};


// Enforce a maximum code length on a sorted node list (zstd HUF_setMaxHeight).
VBZ_HDN uint32_t huf_set_max_height(HufNode* huffNode, uint32_t lastNonNull, uint32_t maxNbBits)
{
    const uint32_t largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= maxNbBits) return largestBits;
    int totalCost = 0;
    const uint32_t baseCost = 1u << (largestBits - maxNbBits);
    int n = (int)lastNonNull;
    while (huffNode[n].nbBits > maxNbBits) {
        totalCost += (int)(baseCost - (1u << (largestBits - huffNode[n].nbBits)));
        huffNode[n].nbBits = (uint8_t)maxNbBits;
        n--;
    }
    while (huffNode[n].nbBits == maxNbBits) n--;
    totalCost >>= (largestBits - maxNbBits);
    uint32_t rankLast[HUF_ABS_MAX_BITS + 2];
    for (int i = 0; i < HUF_ABS_MAX_BITS + 2; ++i) rankLast[i] = HUF_NO_SYMBOL;
    {
        uint32_t currentNbBits = maxNbBits;
        for (int pos = n; pos >= 0; pos--) {
            if (huffNode[pos].nbBits >= currentNbBits) continue;
            currentNbBits = huffNode[pos].nbBits;
            rankLast[maxNbBits - currentNbBits] = (uint32_t)pos;
        }
    }
    while (totalCost > 0) {
        uint32_t nBitsToDecrease = (uint32_t)hb32((uint32_t)totalCost) + 1;
        for (; nBitsToDecrease > 1; nBitsToDecrease--) {
            uint32_t highPos = rankLast[nBitsToDecrease];
            uint32_t lowPos = rankLast[nBitsToDecrease - 1];
            if (highPos == HUF_NO_SYMBOL) continue;
            if (lowPos == HUF_NO_SYMBOL) break;
            uint32_t highTotal = huffNode[highPos].count;
            uint32_t lowTotal = 2 * huffNode[lowPos].count;
            if (highTotal <= lowTotal) break;
        }
        while ((nBitsToDecrease <= HUF_ABS_MAX_BITS) && (rankLast[nBitsToDecrease] == HUF_NO_SYMBOL)) nBitsToDecrease++;
        totalCost -= 1 << (nBitsToDecrease - 1);
        if (rankLast[nBitsToDecrease - 1] == HUF_NO_SYMBOL) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
        huffNode[rankLast[nBitsToDecrease]].nbBits++;
        if (rankLast[nBitsToDecrease] == 0) {
            rankLast[nBitsToDecrease] = HUF_NO_SYMBOL;
        } else {
            rankLast[nBitsToDecrease]--;
            if (huffNode[rankLast[nBitsToDecrease]].nbBits != maxNbBits - nBitsToDecrease)
                rankLast[nBitsToDecrease] = HUF_NO_SYMBOL;
        }
    }
    while (totalCost < 0) {
        if (rankLast[1] == HUF_NO_SYMBOL) {
            while (huffNode[n].nbBits == maxNbBits) n--;
            huffNode[n + 1].nbBits--;
            rankLast[1] = (uint32_t)(n + 1);
            totalCost++;
            continue;
        }
        huffNode[rankLast[1] + 1].nbBits--;
        rankLast[1]++;
        totalCost++;
    }
    return maxNbBits;
}

// Build a length-limited canonical Huffman code from a histogram (zstd HUF_buildCTable_wksp).
//   count[0..maxSymbolValue], count[maxSymbolValue] != 0, at least two non-zero counts.
//   nbBits[s] / code[s] for s <= maxSymbolValue (0 bits = symbol absent).  Returns the table log.
VBZ_HDN uint32_t huf_build(const uint32_t* count, uint32_t maxSymbolValue, uint32_t maxNbBits, uint8_t* nbBits,
                           uint16_t* code, HufBuildWksp* w)
{
    HufNode* const huffNode0 = w->node;
    HufNode* const huffNode = huffNode0 + 1;
    for (int i = 0; i < 513; ++i) {
        huffNode0[i].count = 0;
        huffNode0[i].parent = 0;
        huffNode0[i].byte = 0;
        huffNode0[i].nbBits = 0;
    }
    // sort by decreasing count (bucketed by log2, insertion inside a bucket; ties keep symbol order)
    for (int i = 0; i < 33; ++i) w->rankBase[i] = 0;
    for (uint32_t n = 0; n <= maxSymbolValue; n++) w->rankBase[hb32(count[n] + 1)]++;
    for (int n = 30; n > 0; n--) w->rankBase[n - 1] += w->rankBase[n];
    for (int n = 0; n < 32; n++) w->rankCur[n] = w->rankBase[n];
    for (uint32_t n = 0; n <= maxSymbolValue; n++) {
        const uint32_t c = count[n];
        const uint32_t r = (uint32_t)hb32(c + 1) + 1;
        uint32_t pos = w->rankCur[r]++;
        while ((pos > w->rankBase[r]) && (c > huffNode[pos - 1].count)) {
            huffNode[pos] = huffNode[pos - 1];
            pos--;
        }
        huffNode[pos].count = c;
        huffNode[pos].byte = (uint8_t)n;
    }
    int nonNullRank = (int)maxSymbolValue;
    while (huffNode[nonNullRank].count == 0) nonNullRank--;
    int lowS = nonNullRank;
    int nodeNb = 256;
    const int nodeRoot = nodeNb + lowS - 1;
    int lowN = nodeNb;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (uint16_t)nodeNb;
    nodeNb++;
    lowS -= 2;
    for (int n = nodeNb; n <= nodeRoot; n++) huffNode[n].count = 1u << 30;
    huffNode0[0].count = 1u << 31;  // sentinel in front of the list
    while (nodeNb <= nodeRoot) {
        const int n1 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        const int n2 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        huffNode[nodeNb].count = huffNode[n1].count + huffNode[n2].count;
        huffNode[n1].parent = huffNode[n2].parent = (uint16_t)nodeNb;
        nodeNb++;
    }
    huffNode[nodeRoot].nbBits = 0;
    for (int n = nodeRoot - 1; n >= 256; n--) huffNode[n].nbBits = (uint8_t)(huffNode[huffNode[n].parent].nbBits + 1);
    for (int n = 0; n <= nonNullRank; n++) huffNode[n].nbBits = (uint8_t)(huffNode[huffNode[n].parent].nbBits + 1);
    maxNbBits = huf_set_max_height(huffNode, (uint32_t)nonNullRank, maxNbBits);
    uint16_t nbPerRank[HUF_ABS_MAX_BITS + 2];
    uint16_t valPerRank[HUF_ABS_MAX_BITS + 2];
    for (int i = 0; i < HUF_ABS_MAX_BITS + 2; ++i) nbPerRank[i] = valPerRank[i] = 0;
    for (int n = 0; n <= nonNullRank; n++) nbPerRank[huffNode[n].nbBits]++;
    {
        uint16_t min = 0;
        for (int n = (int)maxNbBits; n > 0; n--) {
            valPerRank[n] = min;
            min = (uint16_t)(min + nbPerRank[n]);
            min >>= 1;
        }
    }
    for (uint32_t n = 0; n <= maxSymbolValue; n++) nbBits[n] = 0;
    for (int n = 0; n <= nonNullRank; n++) nbBits[huffNode[n].byte] = huffNode[n].nbBits;
    for (uint32_t n = 0; n <= maxSymbolValue; n++) code[n] = nbBits[n] ? valPerRank[nbBits[n]]++ : 0;
    return maxNbBits;
}


}  // namespace vbzhip
