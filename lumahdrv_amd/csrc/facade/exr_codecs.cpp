// exr_codecs.cpp -- PIZ and PXR24 chunk decoders for ExrInterface::readFrame (exr_interface.cpp).
//
// Written from the published descriptions of the two schemes (OpenEXR "Technical Introduction" and the structure of its
// ImfPizCompressor / ImfHuf / ImfWav / ImfPxr24Compressor): nothing of OpenEXR is in this image, so the decoders are
// validated against an independent encoder restated in Python (tests/test_exr.py), not against OpenEXR-written files.
//
// PIZ chunk (up to 32 scan lines):  u16 minNonZero, u16 maxNonZero, bitmap bytes [minNonZero..maxNonZero] (which 16-bit
// values occur; 0 always does), i32 length, Huffman stream.  The decoded 16-bit words are, per channel (all rows of
// channel 0, then channel 1, ...), the 2-D wavelet coefficients of the LUT-compacted pixel words (FLOAT / UINT channels:
// two interleaved 16-bit planes, transformed separately); inverse wavelet, inverse LUT, re-interleave by scan line.
// PXR24 chunk (up to 16 scan lines): zlib stream of byte planes -- per scan line and channel, the most significant byte of
// every sample, then the next byte, ... -- of horizontally delta-coded samples; FLOAT samples carry their top 24 bits.
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../../include/luma/luma_exception.h"
#include "exr_codecs.h"

namespace lumaexr {

namespace {

// ------------------------------------------------------------------------------------------------ Huffman
const int ENC_BITS = 16, ENC_SIZE = (1 << ENC_BITS) + 1;  // symbols 0..65535 plus the run-length symbol
const int SHORT_ZERO_RUN = 59, LONG_ZERO_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZERO_RUN - SHORT_ZERO_RUN;  // 6
const int MAX_CODE_LEN = 58;
const int FAST_BITS = 12;

struct BitReader {
    const unsigned char *p, *end;
    uint64_t acc = 0;
    int n = 0;  // valid bits in acc (MSB-first stream)
    BitReader(const unsigned char *b, const unsigned char *e) : p(b), end(e) {}
    bool fill(int need)
    {
        while (n < need) {
            if (p >= end)
                return false;
            acc = (acc << 8) | *p++;
            n += 8;
        }
        return true;
    }
    uint32_t take(int bits)  // caller has filled
    {
        n -= bits;
        return (uint32_t)((acc >> n) & ((1ull << bits) - 1));
    }
};

struct HufTable {
    std::vector<unsigned char> len;       // code length per symbol (0 = unused)
    uint64_t base[MAX_CODE_LEN + 1];       // first code of each length
    uint32_t count[MAX_CODE_LEN + 1];
    uint32_t first[MAX_CODE_LEN + 1];      // index into `syms` of the first symbol of each length
    std::vector<uint32_t> syms;            // symbols ordered by (length, symbol)
    std::vector<uint32_t> fast;            // FAST_BITS-bit prefix -> (symbol << 6) | length, 0 = longer code
};

void unpack_code_lengths(BitReader &br, uint32_t im, uint32_t iM, HufTable &t)
{
    t.len.assign(ENC_SIZE, 0);
    for (uint32_t s = im; s <= iM; s++) {
        if (!br.fill(6))
            throw LumaException("EXR/PIZ: truncated Huffman table");
        const uint32_t l = br.take(6);
        if (l == (uint32_t)LONG_ZERO_RUN) {
            if (!br.fill(8))
                throw LumaException("EXR/PIZ: truncated Huffman table");
            const uint32_t run = br.take(8) + SHORTEST_LONG_RUN;
            if (s + run > iM + 1)
                throw LumaException("EXR/PIZ: corrupt Huffman table");
            s += run - 1;  // those symbols keep length 0
        } else if (l >= (uint32_t)SHORT_ZERO_RUN) {
            const uint32_t run = l - SHORT_ZERO_RUN + 2;
            if (s + run > iM + 1)
                throw LumaException("EXR/PIZ: corrupt Huffman table");
            s += run - 1;
        } else {
            t.len[s] = (unsigned char)l;
        }
    }
}

// canonical codes: the longest codes get the numerically smallest values; within a length, codes ascend with the symbol
void build_codes(HufTable &t)
{
    uint64_t n[MAX_CODE_LEN + 1];
    memset(n, 0, sizeof n);
    memset(t.count, 0, sizeof t.count);
    for (int s = 0; s < ENC_SIZE; s++)
        if (t.len[s]) {
            n[t.len[s]]++;
            t.count[t.len[s]]++;
        }
    uint64_t c = 0;
    for (int l = MAX_CODE_LEN; l > 0; l--) {
        const uint64_t next = (c + n[l]) >> 1;
        t.base[l] = c;
        c = next;
    }
    uint32_t pos = 0;
    for (int l = 1; l <= MAX_CODE_LEN; l++) {
        t.first[l] = pos;
        pos += t.count[l];
    }
    t.syms.assign(pos, 0);
    std::vector<uint32_t> fill(t.first, t.first + MAX_CODE_LEN + 1);
    for (int s = 0; s < ENC_SIZE; s++)
        if (t.len[s])
            t.syms[fill[t.len[s]]++] = (uint32_t)s;
    // a table whose codes collide or overflow their length cannot come from a Huffman tree: reject instead of mis-decoding
    for (int l = 1; l <= MAX_CODE_LEN; l++)
        if (t.count[l] && t.base[l] + t.count[l] > (1ull << l))
            throw LumaException("EXR/PIZ: invalid Huffman code lengths");
    t.fast.assign((size_t)1 << FAST_BITS, 0);
    for (int l = 1; l <= FAST_BITS; l++)
        for (uint32_t k = 0; k < t.count[l]; k++) {
            const uint64_t code = t.base[l] + k;
            const uint32_t sym = t.syms[t.first[l] + k];
            const uint32_t lo = (uint32_t)(code << (FAST_BITS - l)), span = 1u << (FAST_BITS - l);
            for (uint32_t j = 0; j < span; j++) {
                if (t.fast[lo + j])
                    throw LumaException("EXR/PIZ: ambiguous Huffman codes");
                t.fast[lo + j] = (sym << 6) | (uint32_t)l;
            }
        }
}

void huf_decode(const unsigned char *in, size_t nIn, std::vector<uint16_t> &out, size_t nOut)
{
    out.assign(nOut, 0);
    if (nIn == 0) {
        if (nOut)
            throw LumaException("EXR/PIZ: empty Huffman stream");
        return;
    }
    if (nIn < 20)
        throw LumaException("EXR/PIZ: truncated Huffman header");
    auto u32 = [&](size_t o) { return (uint32_t)in[o] | ((uint32_t)in[o + 1] << 8) | ((uint32_t)in[o + 2] << 16) | ((uint32_t)in[o + 3] << 24); };
    const uint32_t im = u32(0), iM = u32(4), nBits = u32(12);
    if (im >= (uint32_t)ENC_SIZE || iM >= (uint32_t)ENC_SIZE || im > iM)
        throw LumaException("EXR/PIZ: corrupt Huffman header");
    HufTable t;
    BitReader tb(in + 20, in + nIn);
    unpack_code_lengths(tb, im, iM, t);
    const unsigned char *dataStart = tb.p;  // the table is byte-aligned at its end: leftover bits are padding
    if ((uint64_t)nBits > 8ull * (uint64_t)(in + nIn - dataStart))
        throw LumaException("EXR/PIZ: Huffman bit count exceeds the data");
    build_codes(t);
    const uint32_t rlc = iM;  // run-length symbol
    BitReader br(dataStart, in + nIn);
    uint64_t left = nBits;  // bits of the stream not yet consumed
    size_t o = 0;
    while (left > 0) {
        // ---- one symbol: table lookup on the next FAST_BITS bits, bit-by-bit extension for longer codes
        const int want = left < (uint64_t)FAST_BITS ? (int)left : FAST_BITS;
        br.fill(want);
        const int have = br.n < want ? br.n : want;
        if (have <= 0)
            throw LumaException("EXR/PIZ: truncated Huffman data");
        const uint32_t peek = (uint32_t)((br.acc >> (br.n - have)) & ((1u << have) - 1)) << (FAST_BITS - have);
        uint32_t sym = 0;
        int l = 0;
        const uint32_t e = t.fast[peek];
        if (e && (int)(e & 63) <= have) {
            l = (int)(e & 63);
            sym = e >> 6;
        } else {
            // The format allows code lengths up to 58 bits; this reader holds at most 56 bits (a 64-bit accumulator refilled by
            // whole bytes).  A Huffman code of length L needs a symbol count ratio of at least Fibonacci(L) between the most
            // and the least frequent symbol, i.e. >= 1.4e11 samples in one chunk for L = 57 -- a chunk is at most 32 scan
            // lines; OpenEXR itself never emits such codes.  Longer codes are reported as invalid data, not mis-decoded.
            const int maxl = left < 56 ? (int)left : 56;
            for (l = FAST_BITS + 1; l <= maxl; l++) {
                if (!br.fill(l))
                    throw LumaException("EXR/PIZ: truncated Huffman data");
                const uint64_t code = (br.acc >> (br.n - l)) & ((1ull << l) - 1);
                if (t.count[l] && code >= t.base[l] && code - t.base[l] < t.count[l]) {
                    sym = t.syms[t.first[l] + (uint32_t)(code - t.base[l])];
                    break;
                }
            }
            if (l > maxl)
                throw LumaException("EXR/PIZ: invalid Huffman code");
        }
        br.n -= l;
        left -= (uint64_t)l;
        if (sym == rlc) {
            if (left < 8 || !br.fill(8))
                throw LumaException("EXR/PIZ: truncated run length");
            const uint32_t run = br.take(8);
            left -= 8;
            if (o == 0 || o + run > nOut)
                throw LumaException("EXR/PIZ: run exceeds the block");
            const uint16_t v = out[o - 1];
            for (uint32_t k = 0; k < run; k++)
                out[o++] = v;
        } else {
            if (o >= nOut)
                throw LumaException("EXR/PIZ: too much Huffman data");
            out[o++] = (uint16_t)sym;
        }
    }
    if (o != nOut)
        throw LumaException("EXR/PIZ: not enough Huffman data");
}

// ------------------------------------------------------------------------------------------------ wavelet
inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b)
{
    const int16_t ls = (int16_t)l, hs = (int16_t)h;
    const int hi = hs;
    const int ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hi);
}

inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b)
{
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}

void wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx)
{
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n)
        p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    while (p >= 1) {
        uint16_t *py = in;
        uint16_t *const ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                if (w14) {
                    wdec14(*px, *p10, i00, i10);
                    wdec14(*p01, *p11, i01, i11);
                    wdec14(i00, i01, *px, *p01);
                    wdec14(i10, i11, *p10, *p11);
                } else {
                    wdec16(*px, *p10, i00, i10);
                    wdec16(*p01, *p11, i01, i11);
                    wdec16(i00, i01, *px, *p01);
                    wdec16(i10, i11, *p10, *p11);
                }
            }
            if (nx & p) {  // odd column at this level: 1-D step in y
                uint16_t *p10 = px + oy1;
                if (w14)
                    wdec14(*px, *p10, i00, *p10);
                else
                    wdec16(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {  // odd line at this level: 1-D step in x
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1;
                if (w14)
                    wdec14(*px, *p01, i00, *p01);
                else
                    wdec16(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

}  // namespace

// raw_out: the block in the standard scan-line layout (per line: channel 0's samples, channel 1's, ...), little-endian
void piz_decode_block(const unsigned char *in, size_t nIn, const std::vector<ChannelLayout> &chans, int width, int lines,
                      std::vector<unsigned char> &raw_out)
{
    size_t nWords = 0;
    for (const ChannelLayout &c : chans)
        nWords += (size_t)width * lines * (c.bytes / 2);
    if (nIn < 4)
        throw LumaException("EXR/PIZ: truncated chunk");
    const unsigned minNZ = in[0] | (in[1] << 8), maxNZ = in[2] | (in[3] << 8);
    if (maxNZ >= 8192)
        throw LumaException("EXR/PIZ: corrupt bitmap range");
    std::vector<unsigned char> bitmap(8192, 0);
    size_t p = 4;
    if (minNZ <= maxNZ) {
        const size_t nb = (size_t)maxNZ - minNZ + 1;
        if (nb > nIn - p)
            throw LumaException("EXR/PIZ: truncated bitmap");
        memcpy(&bitmap[minNZ], in + p, nb);
        p += nb;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; i++)
        if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7))))
            lut[k++] = (uint16_t)i;
    const uint16_t maxValue = (uint16_t)(k - 1);
    if (nIn - p < 4)
        throw LumaException("EXR/PIZ: truncated chunk");
    const int32_t length = (int32_t)((uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16) | ((uint32_t)in[p + 3] << 24));
    p += 4;
    if (length < 0 || (size_t)length > nIn - p)
        throw LumaException("EXR/PIZ: corrupt Huffman length");
    std::vector<uint16_t> words;
    huf_decode(in + p, (size_t)length, words, nWords);
    // inverse wavelet per channel and 16-bit plane, then the inverse LUT
    size_t start = 0;
    std::vector<size_t> chanStart;
    for (const ChannelLayout &c : chans) {
        chanStart.push_back(start);
        const int size = c.bytes / 2;
        for (int j = 0; j < size; j++)
            wav2_decode(&words[start + j], width, size, lines, width * size, maxValue);
        start += (size_t)width * lines * size;
    }
    for (size_t i = 0; i < words.size(); i++)
        words[i] = lut[words[i]];
    // re-interleave: line by line, channel by channel
    raw_out.resize(nWords * 2);
    size_t o = 0;
    for (int y = 0; y < lines; y++)
        for (size_t ci = 0; ci < chans.size(); ci++) {
            const size_t nrow = (size_t)width * (chans[ci].bytes / 2);
            const uint16_t *s = &words[chanStart[ci] + (size_t)y * nrow];
            for (size_t i = 0; i < nrow; i++) {
                raw_out[o++] = (unsigned char)(s[i] & 0xff);
                raw_out[o++] = (unsigned char)(s[i] >> 8);
            }
        }
}

void pxr24_decode_block(const unsigned char *in, size_t nIn, const std::vector<ChannelLayout> &chans, int width, int lines,
                        std::vector<unsigned char> &raw_out)
{
    size_t packed = 0, full = 0;
    for (const ChannelLayout &c : chans) {
        packed += (size_t)width * lines * (c.type == 2 ? 3 : c.bytes);  // FLOAT travels as 24 bits
        full += (size_t)width * lines * c.bytes;
    }
    std::vector<unsigned char> tmp(packed);
    uLongf got = (uLongf)packed;
    if (uncompress(tmp.data(), &got, in, (uLong)nIn) != Z_OK || got != packed)
        throw LumaException("EXR/PXR24: zlib decompression failed");
    raw_out.resize(full);
    size_t ip = 0, o = 0;
    for (int y = 0; y < lines; y++)
        for (const ChannelLayout &c : chans) {
            const int planes = c.type == 2 ? 3 : c.bytes;
            const unsigned char *pl[4] = {0, 0, 0, 0};
            for (int b = 0; b < planes; b++) {
                pl[b] = &tmp[ip];
                ip += (size_t)width;
            }
            uint32_t pixel = 0;
            for (int x = 0; x < width; x++) {
                uint32_t diff = 0;
                for (int b = 0; b < planes; b++)
                    diff = (diff << 8) | pl[b][x];
                if (c.type == 2) {
                    pixel += diff << 8;  // 24 significant bits in the top of the float
                    memcpy(&raw_out[o], &pixel, 4);
                    o += 4;
                } else if (c.type == 1) {
                    pixel = (pixel + diff) & 0xffffu;
                    const uint16_t v = (uint16_t)pixel;
                    memcpy(&raw_out[o], &v, 2);
                    o += 2;
                } else {
                    pixel += diff;
                    memcpy(&raw_out[o], &pixel, 4);
                    o += 4;
                }
            }
        }
}

}  // namespace lumaexr
