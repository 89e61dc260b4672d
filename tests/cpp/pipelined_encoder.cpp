// tests/cpp/pipelined_encoder.cpp -- LumaEncoder::setPipelined(true) against the default (synchronous) mode: N distinct frames
// through both into raw plane streams; the streams must be byte-identical (same planes, same order, same attachments), the
// caller's frame buffer is overwritten right after every encode() (it may be reused as soon as the call returns), the sink
// sees frame i during encode(i+1) or finish(), and the C ABI's stream entry points refuse what they document as refused.
//   pipelined_encoder <dir> <w> <h> <frames> [colorspace] [ptfBits] [profile]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "luma/luma_decoder.h"
#include "luma/luma_encoder.h"
#include "luma/luma_test_pattern.h"
#include "lumahip.h"

static std::vector<char> slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct CountingSink : LumaRawStreamWriter {
    int frames = 0;
    bool addFrame(const LumaPlanes &im)
    {
        frames++;
        return LumaRawStreamWriter::addFrame(im);
    }
};

static void fill(LumaFrame &fr, unsigned w, unsigned h, int i)
{
    lumaTestFrame(fr, w, h);
    const size_t n = fr.pixelCount();
    for (size_t j = (size_t)i; j < n; j += 13)
        fr.buffer[j] = fr.buffer[j] * (1.0f + 0.03f * (float)i) + 0.001f * (float)((j + i) % 97);
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <dir> <w> <h> <frames> [colorspace] [ptfBits] [profile]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const unsigned w = atoi(argv[2]), h = atoi(argv[3]);
    const int frames = atoi(argv[4]);
    try {
        std::string path[2] = {dir + "/sync.lhs", dir + "/pipelined.lhs"};
        float lastMean[2] = {0, 0};
        for (int mode = 0; mode < 2; mode++) {
            LumaEncoder enc;
            LumaEncoderParams p = enc.getParams();
            p.profile = argc > 7 ? atoi(argv[7]) : 2;
            p.bitDepth = p.profile > 1 ? 12 : 8;
            p.ptfBitDepth = argc > 6 ? atoi(argv[6]) : 11;
            p.colorBitDepth = 8;
            p.ptf = LumaQuantizer::PTF_PQ;
            p.colorSpace = argc > 5 ? (LumaQuantizer::colorSpace_t)atoi(argv[5]) : LumaQuantizer::CS_LUV;
            enc.setParams(p);
            CountingSink sink;
            enc.setSink(&sink);
            enc.setPipelined(mode == 1);
            enc.initialize(path[mode].c_str(), w, h);
            LumaFrame fr;
            for (int i = 0; i < frames; i++) {
                fill(fr, w, h, i);
                enc.encode(&fr);
                memset(fr.buffer, 0x7f, fr.pixelCount() * sizeof(float));   // the caller may reuse its frame at once
                const int expect = mode == 0 ? i + 1 : i;                       // one frame of latency when pipelined
                if (sink.frames != expect) {
                    printf("FAIL mode %d: sink has %d frames after encode(%d), expected %d\n", mode, sink.frames, i, expect);
                    return 1;
                }
            }
            enc.finish();
            if (sink.frames != frames) {
                printf("FAIL mode %d: sink has %d frames after finish(), expected %d\n", mode, sink.frames, frames);
                return 1;
            }
            lastMean[mode] = enc.lastMeanLuminance();
        }
        const std::vector<char> a = slurp(path[0]), b = slurp(path[1]);
        if (a.empty() || a != b) {
            printf("FAIL streams differ (%zu / %zu bytes)\n", a.size(), b.size());
            return 1;
        }
        // (the statistic is summed with float atomics, per row band in the synchronous mode: equal to ~1e-6, not to the last bit)
        const float dm = lastMean[0] - lastMean[1];
        if (!((dm < 0 ? -dm : dm) <= 2e-6f * lastMean[0])) {
            printf("FAIL mean luminance of the last frame differs: %.9g / %.9g\n", lastMean[0], lastMean[1]);
            return 1;
        }
        printf("OK streams identical: %d frames, %zu bytes\n", frames, a.size());

        // ---- LumaDecoder::setPipelined(true) against the synchronous decoder: every frame, bit for bit, in order
        std::vector<std::vector<float>> ref;
        {
            LumaDecoder dec(path[0].c_str());
            LumaFrame *out;
            while ((out = dec.decode()) != NULL)
                ref.emplace_back(out->buffer, out->buffer + (size_t)3 * out->width * out->height);
        }
        {
            LumaDecoder dec;
            dec.setPipelined(true);
            dec.initialize(path[0].c_str());
            LumaFrame *out, *prev = NULL;
            size_t n = 0;
            while ((out = dec.decode()) != NULL) {
                if (n >= ref.size() || out->width != w || out->height != h ||
                    memcmp(out->buffer, ref[n].data(), ref[n].size() * sizeof(float)) != 0) {
                    printf("FAIL pipelined decode: frame %zu differs from the synchronous decoder's\n", n);
                    return 1;
                }
                if (out == prev) {
                    printf("FAIL pipelined decode: frame %zu came back in the buffer that is being written\n", n);
                    return 1;
                }
                memset(out->buffer, 0x55, ref[n].size() * sizeof(float));   // the caller may do what it likes with the frame
                prev = out;
                n++;
            }
            if (n != ref.size() || (int)n != frames) {
                printf("FAIL pipelined decode: %zu frames, expected %d\n", n, frames);
                return 1;
            }
            // a seek drops what was read ahead and restarts cleanly
            dec.seekToTime(0.0f, true);
            out = dec.decode();
            if (!out || memcmp(out->buffer, ref[0].data(), ref[0].size() * sizeof(float)) != 0) {
                printf("FAIL pipelined decode after seekToTime(0)\n");
                return 1;
            }
        }
        printf("OK pipelined decode: %d frames identical\n", frames);

        // ---- the C ABI's rules
        LumaEncoder enc;
        enc.setPipelined(true);
        CountingSink sink;
        enc.setSink(&sink);
        enc.initialize((dir + "/rules.lhs").c_str(), w, h);
        lumahip_ctx *ctx = enc.getQuantizer()->context();
        LumaFrame fr;
        fill(fr, w, h, 1);
        LumaPlaneBuffer pb[3];
        for (auto &x : pb)
            x.allocate(w, h, 2);
        float m = 0;
        bool ok = lumahip_encode_stream_pop(ctx, &m) == LUMAHIP_ERR_STATE;                       // nothing in flight
        ok = ok && lumahip_encode_stream_push(ctx, fr.buffer, w, h, 1.0f, 2, pb[0].image().planes, pb[0].image().stride) == LUMAHIP_OK;
        ok = ok && lumahip_encode_stream_push(ctx, fr.buffer, w, h, 1.0f, 2, pb[1].image().planes, pb[1].image().stride) == LUMAHIP_OK;
        ok = ok && lumahip_encode_stream_pending(ctx) == 2;
        ok = ok && lumahip_encode_stream_push(ctx, fr.buffer, w, h, 1.0f, 2, pb[2].image().planes, pb[2].image().stride) == LUMAHIP_ERR_STATE;   // a third
        const float *one[1] = {fr.buffer};
        unsigned char *pl3[3] = {pb[2].image().planes[0], pb[2].image().planes[1], pb[2].image().planes[2]};
        ok = ok && lumahip_encode_frames_host(ctx, one, 1, w, h, 1.0f, 2, pl3, pb[2].image().stride, NULL) == LUMAHIP_ERR_STATE;   // batched form
        {   // the quantizer cannot be replaced under frames in flight
            std::vector<float> lut(2048);
            ok = ok && lumahip_build_lut(LUMAHIP_PTF_PQ, 11, 1e4f, 0.005f, lut.data(), lut.size()) == LUMAHIP_OK;
            ok = ok && lumahip_set_quantizer(ctx, LUMAHIP_PTF_PQ, 11, LUMAHIP_CS_LUV, 8, 1e4f, 0.005f, lut.data(), lut.size()) == LUMAHIP_ERR_STATE;
        }
        // a synchronous single-frame call in between is allowed and must not disturb the frames in flight
        ok = ok && lumahip_encode_frame_host(ctx, fr.buffer, w, h, 1.0f, 2, pb[2].image().planes, pb[2].image().stride, &m, NULL) == LUMAHIP_OK;
        ok = ok && lumahip_encode_stream_pop(ctx, &m) == LUMAHIP_OK && lumahip_encode_stream_pop(ctx, &m) == LUMAHIP_OK;
        ok = ok && lumahip_encode_stream_pending(ctx) == 0;
        for (int p = 0; p < 3 && ok; p++) {
            const LumaPlanes &x = pb[0].image(), &y = pb[1].image(), &z = pb[2].image();
            const size_t bytes = (size_t)x.planeHeight(p) * x.stride[p];
            ok = memcmp(x.planes[p], y.planes[p], bytes) == 0 && memcmp(x.planes[p], z.planes[p], bytes) == 0;
        }
        printf(ok ? "OK stream rules\n" : "FAIL stream rules\n");
        enc.finish();
        return ok ? 0 : 1;
    } catch (LumaException &e) {
        fprintf(stderr, "LumaException: %s\n", e.what());
        return 1;
    }
}
