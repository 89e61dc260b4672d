// tests/cpp/facade_roundtrip.cpp -- exercises the C++ facade the way the reference's test_simple_enc /
// test_simple_dec exercise the reference (test/test_simple_enc.cpp:27-69, test/test_simple_dec.cpp:6-38):
// LumaEncoder with the test_simple_enc parameters encodes N test frames into a stream, LumaDecoder reads
// it back.  Prints FNV-1a-64 digests (standard offset basis) that tests/test_gpu_facade.py compares with the
// oracle's.  usage: facade_roundtrip <stream> <w> <h> <frames> [cs] [ptfBits] [profile]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "luma/luma_decoder.h"
#include "luma/luma_encoder.h"
#include "luma/luma_test_pattern.h"

static uint64_t fnv(const void *d, size_t n, uint64_t h = 0xcbf29ce484222325ull)
{
    const unsigned char *p = (const unsigned char *)d;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

static uint64_t fnv_plane(const LumaPlanes &im, int p)
{
    uint64_t h = 0xcbf29ce484222325ull;
    const size_t rb = (size_t)im.planeWidth(p) * im.bytesPerSample();
    for (unsigned y = 0; y < im.planeHeight(p); y++)
        h = fnv(im.planes[p] + (size_t)y * im.stride[p], rb, h);
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <stream> <w> <h> <frames> [colorspace 0..3] [ptfBits] [profile]\n", argv[0]);
        return 2;
    }
    const char *path = argv[1];
    const unsigned w = atoi(argv[2]), h = atoi(argv[3]);
    const int frames = atoi(argv[4]);
    try {
        {
            LumaEncoder encoder;
            LumaEncoderParams params = encoder.getParams();
            params.profile = argc > 7 ? atoi(argv[7]) : 2;
            params.bitrate = 1000;
            params.keyframeInterval = 0;
            params.bitDepth = params.profile > 1 ? 12 : 8;
            params.ptfBitDepth = argc > 6 ? atoi(argv[6]) : 11;
            params.colorBitDepth = 8;
            params.lossLess = 0;
            params.quantizerScale = 4;
            params.ptf = LumaQuantizer::PTF_PQ;
            params.colorSpace = argc > 5 ? (LumaQuantizer::colorSpace_t)atoi(argv[5]) : LumaQuantizer::CS_LUV;
            encoder.setParams(params);
            for (int f = 1; f <= frames; f++) {
                LumaFrame frame;
                lumaTestFrame(frame, w, h);
                if (f == 1)
                    printf("input %016llx\n", (unsigned long long)fnv(frame.buffer, (size_t)3 * w * h * 4));
                if (!encoder.initialized())
                    encoder.initialize(path, frame.width, frame.height);
                if (f == 2)
                    encoder.setInPlaceCompat(true);  // reference behaviour: the frame holds the transformed floats
                encoder.encode(&frame);
                if (f == 1) {
                    const LumaPlanes &im = encoder.getRawFrame();
                    printf("Y %016llx\nU %016llx\nV %016llx\n", (unsigned long long)fnv_plane(im, 0),
                           (unsigned long long)fnv_plane(im, 1), (unsigned long long)fnv_plane(im, 2));
                    printf("mean %.3f\n", encoder.lastMeanLuminance());
                }
                if (f == 2)
                    printf("transformed %016llx\n", (unsigned long long)fnv(frame.buffer, (size_t)3 * w * h * 4));
            }
            encoder.finish();
            printf("Encoding finished. %d frames encoded.\n", frames);
        }
        LumaDecoder decoder(path);
        int n = 0;
        LumaFrame *out;
        while ((out = decoder.decode()) != NULL) {
            if (n == 0)
                printf("decoded %016llx %ux%u\n", (unsigned long long)fnv(out->buffer, (size_t)3 * out->width * out->height * 4),
                       out->width, out->height);
            n++;
        }
        printf("Decoding finished. %d frames decoded. size %u\n", n, decoder.getQuantizer()->getSize());
        // what lumaplay asks the reader (lumaplay.cpp:200,443)
        printf("reader %.6f %.6f\n", decoder.getReader()->getDuration(), decoder.getReader()->getFrameDuration());
        // per-value API (GPU one-element launches): PQ-11 pins of SURVEY.md 8(c) when the table is PQ-11
        LumaQuantizer *q = decoder.getQuantizer();
        printf("scalar %.9g %.9g %.9g %.9g\n", q->quantize(1.0f, 0), q->quantize(100.0f, 0), q->quantize(0.3f, 1),
               q->dequantize(307.0f, 0));
        // LumaEncoderBase::initialize(file, w, h, ma, mi) opens the container with a luminance range and does nothing else
        // (reference luma_encoder.h:83-91): the encoder is not initialised by it and its parameters are untouched
        {
            LumaEncoder ranged;
            LumaEncoderBase &base = ranged;
            base.initialize((std::string(path) + ".ranged").c_str(), 64, 32, 4000.0f, 0.02f);
            const bool initialisedByBase = ranged.initialized();
            ranged.initialize((std::string(path) + ".ranged").c_str(), 64, 32);
            LumaFrame tf;
            lumaTestFrame(tf, 64, 32);
            ranged.encode(&tf);
            ranged.finish();
            LumaDecoder rd((std::string(path) + ".ranged").c_str());
            printf("ranged %d %.6g %.6g\n", (int)initialisedByBase, rd.getParams().maxLum, rd.getParams().minLum);
        }
        // error conventions
        try {
            LumaEncoder bad;
            bad.initialize("/tmp/should_not_exist.lhs", 5, 4);
            printf("odd-size: no exception\n");
        } catch (LumaException &e) {
            printf("odd-size: %s\n", e.what());
        }
    } catch (LumaException &e) {
        fprintf(stderr, "LumaException: %s\n", e.what());
        return 1;
    }
    return 0;
}
