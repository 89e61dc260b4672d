// tests/cpp/multi_batch.cpp -- the many-GPU layer of the C ABI (lumahip_multi_*) and LumaBatchEncoder, driven from C++ the
// way a caller replacing the reference's frame loop (lumaenc.cpp:205-243 there) would drive them.
//   multi_batch <w> <h> <frames> <stream.lhs>
// 1. `frames` test-pattern frames (each scaled differently so that no two are alike) through ONE context, frame by frame
//    (the reference's loop); 2. the same frames through lumahip_multi_encode_frames_host over N = all visible devices and
//    over N = 2 and 3 logical shards on device 0: planes must equal the single-context planes byte for byte, frame by
//    frame; mean luminances equal; 3. decode likewise; 4. the device-resident form; 5. LumaBatchEncoder writes a stream
//    whose frames LumaDecoder reads back equal to (3).  Prints "OK ..." lines; any mismatch exits non-zero.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "luma/luma_batch_encoder.h"
#include "luma/luma_decoder.h"
#include "luma/luma_test_pattern.h"
#include "lumahip.h"

#define CHECK(cond, ...)                      \
    do {                                      \
        if (!(cond)) {                        \
            std::fprintf(stderr, "FAIL: ");   \
            std::fprintf(stderr, __VA_ARGS__); \
            std::fprintf(stderr, "\n");       \
            return 1;                         \
        }                                     \
    } while (0)

struct Planes {
    std::vector<unsigned char> y, u, v;
};

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s <w> <h> <frames> <stream.lhs>\n", argv[0]);
        return 2;
    }
    const unsigned w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    const unsigned n = std::atoi(argv[3]);
    const char *path = argv[4];
    const int profile = 2;
    const float sc = 1.0f;
    const int st[3] = {(int)((w + 31) / 32 * 32 * 2), (int)((w + 31) / 32 * 32), (int)((w + 31) / 32 * 32)};
    const size_t psz[3] = {(size_t)h * st[0], (size_t)(h / 2) * st[1], (size_t)(h / 2) * st[2]};
    const size_t n3 = (size_t)3 * w * h;

    // frames: the reference's test pattern (src/exr_interface.cpp:50-70 there) scaled by 0.5^(i mod 7) * (1 + i/64)
    std::vector<std::unique_ptr<LumaFrame>> frames;
    for (unsigned i = 0; i < n; i++) {
        std::unique_ptr<LumaFrame> f(new LumaFrame());
        lumaTestFrame(*f, w, h);
        const float k = (1.0f + (float)i / 64.0f) / (float)(1 << (i % 7));
        for (size_t j = 0; j < n3; j++)
            f->buffer[j] *= k;
        frames.push_back(std::move(f));
    }
    std::vector<float> lut(2048);
    CHECK(lumahip_build_lut(LUMAHIP_PTF_PQ, 11, 1e4f, 0.005f, lut.data(), lut.size()) == LUMAHIP_OK, "build_lut");

    // 1. one context, one frame per call
    lumahip_ctx *one = nullptr;
    CHECK(lumahip_create(&one, 0) == LUMAHIP_OK, "no HIP device (there is no CPU fallback)");
    CHECK(lumahip_set_quantizer(one, LUMAHIP_PTF_PQ, 11, LUMAHIP_CS_LUV, 8, 1e4f, 0.005f, lut.data(), lut.size()) == LUMAHIP_OK,
          "%s", lumahip_last_error(one));
    std::vector<Planes> ref(n);
    std::vector<float> refMean(n);
    std::vector<std::vector<float>> refDec(n);
    for (unsigned i = 0; i < n; i++) {
        ref[i].y.assign(psz[0], 0);
        ref[i].u.assign(psz[1], 0);
        ref[i].v.assign(psz[2], 0);
        unsigned char *pl[3] = {ref[i].y.data(), ref[i].u.data(), ref[i].v.data()};
        CHECK(lumahip_encode_frame_host(one, frames[i]->buffer, w, h, sc, profile, pl, st, &refMean[i], nullptr) == LUMAHIP_OK, "%s",
              lumahip_last_error(one));
        refDec[i].assign(n3, 0.0f);
        CHECK(lumahip_decode_frame_host(one, pl, st, w, h, profile, sc, refDec[i].data()) == LUMAHIP_OK, "%s", lumahip_last_error(one));
    }
    std::printf("OK single-context reference: %u frames\n", n);

    int ndev = 0;
    CHECK(lumahip_device_count(&ndev) == LUMAHIP_OK && ndev > 0, "device count");
    struct Case {
        std::vector<int> devices;
        const char *name;
    };
    std::vector<Case> cases;
    {
        Case all;
        for (int d = 0; d < ndev; d++)
            all.devices.push_back(d);
        all.name = "all visible devices";
        cases.push_back(all);
        cases.push_back(Case{{0, 0}, "2 shards on device 0"});
        cases.push_back(Case{{0, 0, 0}, "3 shards on device 0"});
        if (ndev >= 2)
            cases.push_back(Case{{0, 1, 0, 1, 1}, "5 shards over devices 0 and 1"});
    }
    for (const Case &cs : cases) {
        lumahip_multi *m = nullptr;
        CHECK(lumahip_multi_create(&m, cs.devices.data(), (int)cs.devices.size()) == LUMAHIP_OK, "multi_create (%s)", cs.name);
        // the default transport: RCCL exactly when the shards span several devices (host copies on one device) ...
        CHECK(lumahip_multi_set_quantizer(m, LUMAHIP_PTF_PQ, 11, LUMAHIP_CS_LUV, 8, 1e4f, 0.005f, lut.data(), lut.size()) == LUMAHIP_OK,
              "%s", lumahip_multi_last_error(m));
        {
            bool several = false;
            for (int d : cs.devices)
                several = several || d != cs.devices[0];
            CHECK(lumahip_multi_used_rccl(m) == (several ? 1 : 0), "default transport (%s): %s", cs.name, lumahip_multi_transport_note(m));
        }
        // ... and the same table through RCCL on request (a one-rank communicator when there is one device): what follows runs on it
        CHECK(lumahip_multi_set_transport(m, 1) == LUMAHIP_OK && lumahip_multi_set_transport(m, 7) != LUMAHIP_OK, "set_transport");
        CHECK(lumahip_multi_set_quantizer(m, LUMAHIP_PTF_PQ, 11, LUMAHIP_CS_LUV, 8, 1e4f, 0.005f, lut.data(), lut.size()) == LUMAHIP_OK,
              "%s", lumahip_multi_last_error(m));
        CHECK(lumahip_multi_used_rccl(m) == 1, "the table did not travel over RCCL (%s)", cs.name);
        const int ns = lumahip_multi_shards(m);
        // block sharding: contiguous, complete, the first n % ns shards one frame more
        unsigned covered = 0;
        for (int s = 0; s < ns; s++) {
            unsigned first = 0, count = 0;
            CHECK(lumahip_shard_range(n, s, ns, &first, &count) == LUMAHIP_OK && first == covered, "shard_range");
            CHECK(count == n / ns + ((unsigned)s < n % ns ? 1u : 0u), "shard sizes");
            covered += count;
        }
        CHECK(covered == n, "shards do not cover the stream");
        // 2. host batch
        std::vector<Planes> got(n);
        std::vector<const float *> rgb(n);
        std::vector<unsigned char *> pl(3 * (size_t)n);
        for (unsigned i = 0; i < n; i++) {
            got[i].y.assign(psz[0], 0xee);
            got[i].u.assign(psz[1], 0xee);
            got[i].v.assign(psz[2], 0xee);
            rgb[i] = frames[i]->buffer;
            pl[3 * i] = got[i].y.data();
            pl[3 * i + 1] = got[i].u.data();
            pl[3 * i + 2] = got[i].v.data();
        }
        std::vector<float> mean(n, -1.0f);
        CHECK(lumahip_multi_encode_frames_host(m, rgb.data(), n, w, h, sc, profile, pl.data(), st, mean.data()) == LUMAHIP_OK, "%s",
              lumahip_multi_last_error(m));
        for (unsigned i = 0; i < n; i++) {
            // rows only: the padding behind each row is the caller's
            for (unsigned y = 0; y < h; y++)
                CHECK(std::memcmp(got[i].y.data() + (size_t)y * st[0], ref[i].y.data() + (size_t)y * st[0], 2 * w) == 0,
                      "%s: frame %u Y row %u differs", cs.name, i, y);
            for (unsigned y = 0; y < h / 2; y++) {
                CHECK(std::memcmp(got[i].u.data() + (size_t)y * st[1], ref[i].u.data() + (size_t)y * st[1], w) == 0,
                      "%s: frame %u U row %u differs", cs.name, i, y);
                CHECK(std::memcmp(got[i].v.data() + (size_t)y * st[2], ref[i].v.data() + (size_t)y * st[2], w) == 0,
                      "%s: frame %u V row %u differs", cs.name, i, y);
            }
            CHECK(mean[i] == refMean[i] || std::fabs(mean[i] - refMean[i]) <= 2e-5f * std::fabs(refMean[i]), "%s: frame %u mean %g vs %g",
                  cs.name, i, mean[i], refMean[i]);
        }
        // 3. decode
        std::vector<std::vector<float>> dec(n, std::vector<float>(n3, -1.0f));
        std::vector<float *> outp(n);
        std::vector<const unsigned char *> cpl(pl.begin(), pl.end());
        for (unsigned i = 0; i < n; i++)
            outp[i] = dec[i].data();
        CHECK(lumahip_multi_decode_frames_host(m, cpl.data(), st, n, w, h, profile, sc, outp.data()) == LUMAHIP_OK, "%s",
              lumahip_multi_last_error(m));
        for (unsigned i = 0; i < n; i++)
            CHECK(std::memcmp(dec[i].data(), refDec[i].data(), n3 * sizeof(float)) == 0, "%s: decoded frame %u differs", cs.name, i);
        // 4. device-resident: each shard's block uploaded to its GPU, one launch per shard
        std::vector<float *> dIn(ns, nullptr);
        std::vector<unsigned char *> dPl(3 * (size_t)ns, nullptr);
        std::vector<unsigned> cnt(ns, 0), firsts(ns, 0);
        for (int s = 0; s < ns; s++) {
            lumahip_ctx *c = lumahip_multi_ctx(m, s);
            (void)lumahip_shard_range(n, s, ns, &firsts[s], &cnt[s]);
            if (!cnt[s])
                continue;
            CHECK(lumahip_malloc(c, (void **)&dIn[s], cnt[s] * n3 * sizeof(float)) == LUMAHIP_OK, "malloc");
            for (int p = 0; p < 3; p++)
                CHECK(lumahip_malloc(c, (void **)&dPl[3 * s + p], cnt[s] * psz[p]) == LUMAHIP_OK, "malloc");
            for (unsigned k = 0; k < cnt[s]; k++)
                CHECK(lumahip_memcpy_h2d(c, dIn[s] + k * n3, frames[firsts[s] + k]->buffer, n3 * sizeof(float)) == LUMAHIP_OK, "h2d");
        }
        std::vector<const float *> cIn(dIn.begin(), dIn.end());
        CHECK(lumahip_multi_encode_frames_device(m, cIn.data(), n3, cnt.data(), w, h, sc, profile, dPl.data(), st, psz) == LUMAHIP_OK, "%s",
              lumahip_multi_last_error(m));
        CHECK(lumahip_multi_sync(m) == LUMAHIP_OK, "sync");
        std::vector<unsigned char> back(psz[0]);
        for (int s = 0; s < ns; s++) {
            lumahip_ctx *c = lumahip_multi_ctx(m, s);
            for (unsigned k = 0; k < cnt[s]; k++) {
                const unsigned i = firsts[s] + k;
                CHECK(lumahip_memcpy_d2h(c, back.data(), dPl[3 * s] + k * psz[0], psz[0]) == LUMAHIP_OK, "d2h");
                for (unsigned y = 0; y < h; y++)
                    CHECK(std::memcmp(back.data() + (size_t)y * st[0], ref[i].y.data() + (size_t)y * st[0], 2 * w) == 0,
                          "%s: device-resident frame %u Y row %u differs", cs.name, i, y);
                CHECK(lumahip_memcpy_d2h(c, back.data(), dPl[3 * s + 1] + k * psz[1], psz[1]) == LUMAHIP_OK, "d2h");
                for (unsigned y = 0; y < h / 2; y++)
                    CHECK(std::memcmp(back.data() + (size_t)y * st[1], ref[i].u.data() + (size_t)y * st[1], w) == 0,
                          "%s: device-resident frame %u U row %u differs", cs.name, i, y);
            }
            (void)lumahip_free(c, dIn[s]);
            for (int p = 0; p < 3; p++)
                (void)lumahip_free(c, dPl[3 * s + p]);
        }
        lumahip_multi_destroy(m);
        std::printf("OK %s: %d shard(s), %u frames encode + decode + device-resident encode equal the single context\n", cs.name, ns, n);
    }

    // 5. the facade: LumaBatchEncoder -> stream -> LumaDecoder
    {
        LumaBatchEncoder enc;
        LumaEncoderParams p = enc.getParams();
        p.profile = 2;
        p.bitDepth = 12;
        p.ptfBitDepth = 11;
        p.colorBitDepth = 8;
        enc.setParams(p);
        const int dv[3] = {0, 0, 0};
        enc.initialize(path, w, h, false, ndev > 1 ? NULL : dv, ndev > 1 ? 0 : 3);
        // several devices: RCCL broadcast; three shards on the one device of this box: nothing to broadcast, host copies
        CHECK(enc.quantizerCameOverRccl() == (ndev > 1), "LumaBatchEncoder: table transport (RCCL exactly when several devices are in use)");
        std::vector<LumaFrame *> ptrs;
        for (auto &f : frames)
            ptrs.push_back(f.get());
        // two calls of unequal size: the second exercises a batch smaller than the shard count as well
        const unsigned first = n > 2 ? n - 2 : n;
        enc.encode(ptrs.data(), first);
        if (n > first)
            enc.encode(ptrs.data() + first, n - first);
        enc.finish();
        CHECK(enc.framesEncoded() == n, "frame count");
        LumaDecoder dec(path);
        for (unsigned i = 0; i < n; i++) {
            LumaFrame *f = dec.decode();
            CHECK(f != NULL, "stream ended at frame %u", i);
            CHECK(std::memcmp(f->buffer, refDec[i].data(), n3 * sizeof(float)) == 0, "LumaBatchEncoder stream: frame %u differs", i);
        }
        CHECK(dec.decode() == NULL, "stream holds more frames than were encoded");
        std::printf("OK LumaBatchEncoder: %u shard(s), %u frames in stream order\n", enc.shards(), n);
    }
    lumahip_destroy(one);
    std::printf("OK all\n");
    return 0;
}
