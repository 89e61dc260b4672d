// exr_interface.cpp -- minimal OpenEXR scan-line reader / writer (see include/exr_interface.h).
// File format per the OpenEXR file layout specification; no OpenEXR code or headers involved.
#include "../../../include/exr_interface.h"
#include "exr_codecs.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../../include/luma/luma_test_pattern.h"

namespace {

typedef std::vector<unsigned char> Bytes;

struct Channel {
    std::string name;
    int type;  // 0 UINT, 1 HALF, 2 FLOAT
    int xs, ys;
    int size() const { return type == 1 ? 2 : 4; }
};

struct Reader {
    const Bytes &b;
    size_t p;
    explicit Reader(const Bytes &bytes) : b(bytes), p(0) {}
    void need(size_t n) const
    {
        if (n > b.size() || p > b.size() - n)  // no p + n: p may come from an untrusted 64-bit chunk offset
            throw LumaException("EXR: unexpected end of file");
    }
    int32_t i32()
    {
        need(4);
        int32_t v;
        memcpy(&v, &b[p], 4);
        p += 4;
        return v;
    }
    uint64_t u64()
    {
        need(8);
        uint64_t v;
        memcpy(&v, &b[p], 8);
        p += 8;
        return v;
    }
    std::string str()
    {
        std::string s;
        for (;;) {
            need(1);
            char c = (char)b[p++];
            if (!c)
                break;
            s.push_back(c);
            if (s.size() > 255)
                throw LumaException("EXR: malformed string");
        }
        return s;
    }
};

Bytes slurp(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        throw LumaException((std::string("Cannot open image file \"") + path + "\".").c_str());
    Bytes d;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0)
        d.insert(d.end(), buf, buf + n);
    fclose(f);
    return d;
}

// ZIP / RLE post-processing shared by both: predictor + byte interleave
void unpredict_interleave(Bytes &tmp, Bytes &out)
{
    for (size_t i = 1; i < tmp.size(); i++)
        tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);
    out.resize(tmp.size());
    const size_t half = (tmp.size() + 1) / 2;
    for (size_t i = 0, a = 0, c = half; i < tmp.size();) {
        out[i++] = tmp[a++];
        if (i < tmp.size())
            out[i++] = tmp[c++];
    }
}

void deinterleave_predict(const Bytes &raw, Bytes &tmp)
{
    tmp.resize(raw.size());
    const size_t half = (raw.size() + 1) / 2;
    for (size_t i = 0, a = 0, c = half; i < raw.size();) {
        tmp[a++] = raw[i++];
        if (i < raw.size())
            tmp[c++] = raw[i++];
    }
    unsigned char prev = tmp.empty() ? 0 : tmp[0];
    for (size_t i = 1; i < tmp.size(); i++) {
        const unsigned char cur = tmp[i];
        tmp[i] = (unsigned char)(cur - prev + 128);
        prev = cur;
    }
}

void rle_decode(const unsigned char *in, size_t n, Bytes &out, size_t expect)
{
    out.clear();
    size_t i = 0;
    while (i < n) {
        const signed char c = (signed char)in[i++];
        if (c < 0) {
            const size_t cnt = (size_t)(-(int)c);
            if (i + cnt > n)
                throw LumaException("EXR: corrupt RLE data");
            out.insert(out.end(), in + i, in + i + cnt);
            i += cnt;
        } else {
            if (i >= n)
                throw LumaException("EXR: corrupt RLE data");
            out.insert(out.end(), (size_t)c + 1, in[i++]);
        }
    }
    if (out.size() != expect)
        throw LumaException("EXR: RLE block has the wrong size");
}

}  // namespace

uint16_t ExrInterface::floatToHalf(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (e == 255)  // inf / nan (keep a nan a nan)
        return (uint16_t)(sign | 0x7c00u | (m ? (0x200u | (m >> 13)) : 0u));
    const int32_t he = e - 127 + 15;
    if (he >= 31)
        return (uint16_t)(sign | 0x7c00u);  // overflow -> infinity
    if (he <= 0) {
        if (he < -10)
            return (uint16_t)sign;  // underflows to zero
        m |= 0x800000u;             // denormal half: shift with round-to-nearest-even
        const int shift = 14 - he;
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        uint32_t r = q;
        if (rem > halfway || (rem == halfway && (q & 1)))
            r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)he << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1)))
        r++;  // may carry into the exponent, up to infinity: correct
    return (uint16_t)(sign | r);
}

float ExrInterface::halfToFloat(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            int s = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                s++;
            }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e - 15 + 127) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

bool ExrInterface::testFrame(LumaFrame &frame, unsigned int w, unsigned int h) { return lumaTestFrame(frame, w, h); }

static bool readFrameImpl(const char *inputFile, LumaFrame &frame);

bool ExrInterface::readFrame(const char *inputFile, LumaFrame &frame)
{
    try {
        return readFrameImpl(inputFile, frame);
    } catch (const LumaException &) {
        throw;
    } catch (const std::exception &e) {  // bad_alloc, length_error ...: same conversion the reference applies
        throw LumaException(e.what());
    }
}

static bool readFrameImpl(const char *inputFile, LumaFrame &frame)
{
    const int NO_COMPRESSION = ExrInterface::NO_COMPRESSION, RLE_COMPRESSION = ExrInterface::RLE_COMPRESSION,
              ZIP_COMPRESSION = ExrInterface::ZIP_COMPRESSION, PIZ_COMPRESSION = ExrInterface::PIZ_COMPRESSION,
              PXR24_COMPRESSION = ExrInterface::PXR24_COMPRESSION;
    const Bytes data = slurp(inputFile);
    Reader rd(data);
    if (rd.i32() != 20000630)
        throw LumaException((std::string("File \"") + inputFile + "\" is not an OpenEXR image file.").c_str());
    const int32_t version = rd.i32();
    if ((version & 0xff) != 2 || (version & (0x200 | 0x800 | 0x1000)))
        throw LumaException("EXR: only single-part scan-line files are supported (no tiles, deep data or multi-part)");
    std::vector<Channel> chans;
    int compression = -1, lineOrder = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    bool haveDW = false;
    for (;;) {
        const std::string name = rd.str();
        if (name.empty())
            break;
        const std::string type = rd.str();
        const int32_t size = rd.i32();
        if (size < 0)
            throw LumaException("EXR: malformed attribute");
        rd.need((size_t)size);
        const size_t end = rd.p + (size_t)size;
        if (name == "channels" && type == "chlist") {
            for (;;) {
                Channel c;
                c.name = rd.str();
                if (c.name.empty())
                    break;
                c.type = rd.i32();
                rd.need(4);
                rd.p += 4;  // pLinear + 3 reserved
                c.xs = rd.i32();
                c.ys = rd.i32();
                chans.push_back(c);
            }
        } else if (name == "compression") {
            if (size < 1)
                throw LumaException("EXR: malformed compression attribute");
            compression = data[rd.p];
        } else if (name == "dataWindow") {
            for (int i = 0; i < 4; i++)
                dw[i] = rd.i32();
            haveDW = true;
        } else if (name == "lineOrder") {
            if (size < 1)
                throw LumaException("EXR: malformed lineOrder attribute");
            lineOrder = data[rd.p];
        }
        rd.p = end;
    }
    if (chans.empty() || !haveDW || compression < 0)
        throw LumaException("EXR: header lacks channels / dataWindow / compression");
    if (compression > PXR24_COMPRESSION)
        throw LumaException("EXR: unsupported compression (NONE, RLE, ZIPS, ZIP, PIZ and PXR24 are implemented; B44 / DWA are not)");
    const long W = (long)dw[2] - dw[0] + 1, H = (long)dw[3] - dw[1] + 1;
    if (W <= 0 || H <= 0 || W > 65536 || H > 65536)
        throw LumaException("EXR: bad data window");
    size_t lineBytes = 0;
    for (size_t i = 0; i < chans.size(); i++) {
        if (chans[i].xs != 1 || chans[i].ys != 1)
            throw LumaException("EXR: sub-sampled channels are not supported");
        if (chans[i].type < 0 || chans[i].type > 2)
            throw LumaException("EXR: unknown pixel type");
        lineBytes += (size_t)W * chans[i].size();
    }
    // which of R,G,B,A are present (the RgbaChannels mask of Imf::RgbaInputFile::channels())
    int mask = 0;
    long off[4] = {-1, -1, -1, -1};
    int typ[4] = {0, 0, 0, 0};
    {
        size_t o = 0;
        for (size_t i = 0; i < chans.size(); i++) {
            const char *nm[4] = {"R", "G", "B", "A"};
            for (int k = 0; k < 4; k++)
                if (chans[i].name == nm[k]) {
                    mask |= 1 << k;
                    off[k] = (long)o;
                    typ[k] = chans[i].type;
                }
            o += (size_t)W * chans[i].size();
        }
    }
    int src[3];
    switch (mask) {
    case 7: case 15: src[0] = 0; src[1] = 1; src[2] = 2; break;  // WRITE_RGB, WRITE_RGBA
    case 1: src[0] = src[1] = src[2] = 0; break;                  // WRITE_R
    case 2: src[0] = src[1] = src[2] = 1; break;                  // WRITE_G
    case 4: src[0] = src[1] = src[2] = 2; break;                  // WRITE_B
    default: throw LumaException("Reading of luminance only frames not yet supported");
    }
    (void)lineOrder;  // every chunk carries its own y, so the order of chunks in the file does not matter
    // A corrupt header must not make us allocate gigabytes: every scan line needs a chunk in the file, and deflate
    // expands at most ~1032:1, so the decoded pixel bytes are bounded by the file size.
    if ((double)lineBytes * (double)H > 1040.0 * (double)data.size() + 65536.0)
        throw LumaException("EXR: data window is inconsistent with the file size");

    frame.width = (unsigned int)W;
    frame.height = (unsigned int)H;
    frame.channels = 3;
    if (!frame.init())
        throw LumaException("Cannot allocate memory for input frame");

    const int linesPerBlock = (compression == PIZ_COMPRESSION) ? 32 : (compression == ZIP_COMPRESSION || compression == PXR24_COMPRESSION) ? 16 : 1;
    std::vector<lumaexr::ChannelLayout> layout;
    for (size_t i = 0; i < chans.size(); i++) {
        lumaexr::ChannelLayout cl;
        cl.type = chans[i].type;
        cl.bytes = chans[i].size();
        layout.push_back(cl);
    }
    const long nblocks = (H + linesPerBlock - 1) / linesPerBlock;
    std::vector<uint64_t> offsets((size_t)nblocks);
    for (long i = 0; i < nblocks; i++)
        offsets[(size_t)i] = rd.u64();
    Bytes raw, tmp;
    for (long blk = 0; blk < nblocks; blk++) {
        Reader c(data);
        if (offsets[(size_t)blk] >= (uint64_t)data.size())
            throw LumaException("EXR: chunk offset outside the file");
        c.p = (size_t)offsets[(size_t)blk];
        const int32_t y0 = c.i32();
        const int32_t dsz = c.i32();
        if (dsz < 0 || y0 < dw[1] || y0 > dw[3])
            throw LumaException("EXR: corrupt chunk header");
        c.need((size_t)dsz);
        const long lines = std::min<long>(linesPerBlock, (long)dw[3] - y0 + 1);
        const size_t expect = lineBytes * (size_t)lines;
        const unsigned char *payload = &data[c.p];
        if ((size_t)dsz == expect || compression == NO_COMPRESSION) {
            if ((size_t)dsz != expect)
                throw LumaException("EXR: chunk has the wrong size");
            raw.assign(payload, payload + dsz);  // stored uncompressed
        } else if (compression == RLE_COMPRESSION) {
            rle_decode(payload, (size_t)dsz, tmp, expect);
            unpredict_interleave(tmp, raw);
        } else if (compression == PIZ_COMPRESSION) {
            lumaexr::piz_decode_block(payload, (size_t)dsz, layout, (int)W, (int)lines, raw);
            if (raw.size() != expect)
                throw LumaException("EXR: PIZ block has the wrong size");
        } else if (compression == PXR24_COMPRESSION) {
            lumaexr::pxr24_decode_block(payload, (size_t)dsz, layout, (int)W, (int)lines, raw);
            if (raw.size() != expect)
                throw LumaException("EXR: PXR24 block has the wrong size");
        } else {
            tmp.resize(expect);
            uLongf got = (uLongf)expect;
            if (uncompress(tmp.data(), &got, payload, (uLong)dsz) != Z_OK || got != expect)
                throw LumaException("EXR: zlib decompression failed");
            unpredict_interleave(tmp, raw);
        }
        for (long l = 0; l < lines; l++) {
            const size_t y = (size_t)(y0 - dw[1] + l);
            const unsigned char *line = &raw[(size_t)l * lineBytes];
            for (int ch = 0; ch < 3; ch++) {
                const int k = src[ch];
                float *dst = frame.getChannel((unsigned)ch) + y * (size_t)W;
                const unsigned char *p = line + off[k];
                for (long x = 0; x < W; x++) {
                    uint16_t hv;
                    if (typ[k] == 1) {
                        memcpy(&hv, p + 2 * x, 2);
                    } else if (typ[k] == 2) {
                        float f;
                        memcpy(&f, p + 4 * x, 4);
                        hv = ExrInterface::floatToHalf(f);  // Imf::Rgba holds halfs: FLOAT channels are narrowed on read
                    } else {
                        uint32_t u;
                        memcpy(&u, p + 4 * x, 4);
                        hv = ExrInterface::floatToHalf((float)u);
                    }
                    dst[x] = ExrInterface::halfToFloat(hv);
                }
            }
        }
    }
    return true;
}

bool ExrInterface::writeFrame(const char *outputFile, LumaFrame &frame) { return writeFrame(outputFile, frame, ZIP_COMPRESSION, false); }

bool ExrInterface::writeFrame(const char *outputFile, LumaFrame &frame, Compression comp, bool asFloat)
{
    if (frame.buffer == NULL)
        throw LumaException("Frame does not contain any data");
    if (comp != NO_COMPRESSION && comp != RLE_COMPRESSION && comp != ZIPS_COMPRESSION && comp != ZIP_COMPRESSION)
        throw LumaException("EXR: writeFrame produces NONE, RLE, ZIPS or ZIP files");
    const unsigned int W = frame.width, H = frame.height;
    Bytes out;
    auto put = [&](const void *p, size_t n) { out.insert(out.end(), (const unsigned char *)p, (const unsigned char *)p + n); };
    auto puti = [&](int32_t v) { put(&v, 4); };
    auto putf = [&](float v) { put(&v, 4); };
    auto puts_ = [&](const char *s) { put(s, strlen(s) + 1); };
    auto attr = [&](const char *name, const char *type, int32_t size) { puts_(name); puts_(type); puti(size); };
    puti(20000630);
    puti(2);
    attr("channels", "chlist", 3 * 18 + 1);
    const char *names[3] = {"B", "G", "R"};  // alphabetical
    for (int i = 0; i < 3; i++) {
        puts_(names[i]);
        puti(asFloat ? 2 : 1);
        const unsigned char z[4] = {0, 0, 0, 0};
        put(z, 4);
        puti(1);
        puti(1);
    }
    out.push_back(0);
    attr("compression", "compression", 1);
    out.push_back((unsigned char)comp);
    attr("dataWindow", "box2i", 16);
    puti(0); puti(0); puti((int32_t)W - 1); puti((int32_t)H - 1);
    attr("displayWindow", "box2i", 16);
    puti(0); puti(0); puti((int32_t)W - 1); puti((int32_t)H - 1);
    attr("lineOrder", "lineOrder", 1);
    out.push_back(0);
    attr("pixelAspectRatio", "float", 4);
    putf(1.0f);
    attr("screenWindowCenter", "v2f", 8);
    putf(0.0f); putf(0.0f);
    attr("screenWindowWidth", "float", 4);
    putf(1.0f);
    out.push_back(0);

    const int linesPerBlock = (comp == ZIP_COMPRESSION) ? 16 : 1;
    const size_t nblocks = ((size_t)H + linesPerBlock - 1) / linesPerBlock;
    const size_t tablePos = out.size();
    out.resize(out.size() + 8 * nblocks);
    const size_t px = asFloat ? 4 : 2, lineBytes = 3 * (size_t)W * px;
    const int chOrder[3] = {2, 1, 0};  // B, G, R
    Bytes raw, tmp, packed;
    for (size_t blk = 0; blk < nblocks; blk++) {
        const size_t y0 = blk * linesPerBlock, lines = std::min<size_t>(linesPerBlock, H - y0);
        raw.resize(lineBytes * lines);
        for (size_t l = 0; l < lines; l++)
            for (int c = 0; c < 3; c++) {
                const float *srcp = frame.getChannel((unsigned)chOrder[c]) + (y0 + l) * W;
                unsigned char *d = &raw[l * lineBytes + (size_t)c * W * px];
                for (unsigned int x = 0; x < W; x++) {
                    if (asFloat) {
                        memcpy(d + 4 * x, &srcp[x], 4);
                    } else {
                        const uint16_t hv = floatToHalf(srcp[x]);
                        memcpy(d + 2 * x, &hv, 2);
                    }
                }
            }
        const Bytes *payload = &raw;
        if (comp == ZIP_COMPRESSION || comp == ZIPS_COMPRESSION) {
            deinterleave_predict(raw, tmp);
            uLongf cap = compressBound((uLong)tmp.size());
            packed.resize(cap);
            if (compress(packed.data(), &cap, tmp.data(), (uLong)tmp.size()) == Z_OK && cap < raw.size()) {
                packed.resize(cap);
                payload = &packed;
            }
        } else if (comp == RLE_COMPRESSION) {
            // runs are optional in the format; literal packets only when RLE would not shrink is allowed, but a
            // writer may always fall back to the uncompressed form, which is what we do for simplicity
            payload = &raw;
        }
        const uint64_t pos = out.size();
        memcpy(&out[tablePos + 8 * blk], &pos, 8);
        puti((int32_t)y0);
        puti((int32_t)payload->size());
        put(payload->data(), payload->size());
    }
    FILE *f = fopen(outputFile, "wb");
    if (!f)
        throw LumaException((std::string("Cannot open image file \"") + outputFile + "\" for writing.").c_str());
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok)
        throw LumaException("EXR: short write");
    return true;
}
