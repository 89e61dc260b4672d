// tests/cpp/multi_stress.cpp -- randomized stress of the many-shard host path (lumahip_multi_{encode,decode}_frames_host:
// one host thread per shard, each with its own context, copy-thread pool and staging rings) against ONE context.
//   multi_stress <iterations> <shards> [seed]
// Every iteration draws a frame size (even, 2..322 x 2..182), a frame count (1..24, often fewer than shards), a VP9 profile
// (0..3), pageable or registered (pinned) buffers and plane strides with random extra padding; every 16th iteration also
// replaces the quantizer (Lu'v' PQ-11 / HDR10 YCbCr PQ-10 / XYZ LINEAR-12).  The planes and the decoded floats of the sharded
// calls must equal, byte for byte, those of a single context looping over the frames.  All shards sit on device 0, so the
// threads contend for one GPU -- the arrangement in which a missing wait shows.  Also the program tests/test_gpu_tsan.py runs
// under ThreadSanitizer.  Prints "OK multi_stress ..." or "FAIL ..." (exit 1).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lumahip.h"

#define CHECK(cond, ...)                       \
    do {                                       \
        if (!(cond)) {                         \
            std::fprintf(stderr, "FAIL: ");    \
            std::fprintf(stderr, __VA_ARGS__); \
            std::fprintf(stderr, "\n");        \
            return 1;                          \
        }                                      \
    } while (0)

static uint64_t g_state;
static uint32_t rnd()
{
    g_state = g_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (uint32_t)(g_state >> 33);
}

struct Cfg {
    int ptf, cs;
    unsigned bits, bitsC;
    float maxLum, minLum, sc;
};

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
    const int shards = argc > 2 ? std::atoi(argv[2]) : 8;
    g_state = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 20260929ULL;
    const Cfg cfgs[3] = {{LUMAHIP_PTF_PQ, LUMAHIP_CS_LUV, 11, 8, 1e4f, 0.005f, 1.0f},
                         {LUMAHIP_PTF_PQ, LUMAHIP_CS_YCBCR, 10, 10, 1000.0f, 0.01f, 20.0f},
                         {LUMAHIP_PTF_LINEAR, LUMAHIP_CS_XYZ, 12, 8, 1e4f, 0.005f, 1.0f}};
    std::vector<int> devs(shards, 0);
    lumahip_multi *m = nullptr;
    CHECK(lumahip_multi_create(&m, devs.data(), shards) == LUMAHIP_OK, "multi_create: no HIP device (there is no CPU fallback)");
    lumahip_ctx *one = nullptr;
    CHECK(lumahip_create(&one, 0) == LUMAHIP_OK, "create");
    const Cfg *cur = nullptr;
    unsigned long frames_total = 0;
    for (int it = 0; it < iters; it++) {
        if (it % 16 == 0) {
            cur = &cfgs[(it / 16) % 3];
            std::vector<float> lut((size_t)1 << cur->bits);
            CHECK(lumahip_build_lut(cur->ptf, cur->bits, cur->maxLum, cur->minLum, lut.data(), lut.size()) == LUMAHIP_OK, "build_lut");
            CHECK(lumahip_multi_set_quantizer(m, cur->ptf, cur->bits, cur->cs, cur->bitsC, cur->maxLum, cur->minLum, lut.data(), lut.size()) == LUMAHIP_OK,
                  "%s", lumahip_multi_last_error(m));
            CHECK(lumahip_set_quantizer(one, cur->ptf, cur->bits, cur->cs, cur->bitsC, cur->maxLum, cur->minLum, lut.data(), lut.size()) == LUMAHIP_OK,
                  "%s", lumahip_last_error(one));
        }
        const unsigned w = 2 + 2 * (rnd() % 161), h = 2 + 2 * (rnd() % 91);
        const unsigned n = 1 + rnd() % 24;
        const int profile = (int)(rnd() % 4);
        const bool sub = profile == 0 || profile == 2;
        const int bps = profile > 1 ? 2 : 1;
        const bool pin = (rnd() % 3) == 0;
        const unsigned cw = sub ? w / 2 : w, ch = sub ? h / 2 : h;
        const int st[3] = {(int)(w * bps + 2 * (rnd() % 9)), (int)(cw * bps + 2 * (rnd() % 9)), (int)(cw * bps + 2 * (rnd() % 9))};
        const size_t psz[3] = {(size_t)h * st[0], (size_t)ch * st[1], (size_t)ch * st[2]};
        const size_t n3 = (size_t)3 * w * h;
        // frames: log-uniform positive values, a few specials
        std::vector<std::vector<float>> fr(n, std::vector<float>(n3));
        for (auto &f : fr) {
            for (auto &v : f) {
                const uint32_t bits = ((117u + rnd() % 24u) << 23) | (rnd() & 0x7fe000u);
                std::memcpy(&v, &bits, 4);
            }
            f[rnd() % n3] = 0.0f;
            f[rnd() % n3] = -1.5f;
        }
        std::vector<std::vector<unsigned char>> pa(3 * (size_t)n), pb(3 * (size_t)n);
        std::vector<const float *> rgb(n);
        std::vector<unsigned char *> pla(3 * (size_t)n), plb(3 * (size_t)n);
        for (unsigned i = 0; i < n; i++) {
            rgb[i] = fr[i].data();
            for (int p = 0; p < 3; p++) {
                pa[3 * i + p].assign(psz[p], 0xA5);
                pb[3 * i + p].assign(psz[p], 0xA5);
                pla[3 * i + p] = pa[3 * i + p].data();
                plb[3 * i + p] = pb[3 * i + p].data();
            }
            if (pin)
                (void)lumahip_host_register(one, fr[i].data(), n3 * sizeof(float));
        }
        std::vector<float> ma(n), mb(n);
        CHECK(lumahip_multi_encode_frames_host(m, rgb.data(), n, w, h, cur->sc, profile, pla.data(), st, ma.data()) == LUMAHIP_OK,
              "iteration %d: %s", it, lumahip_multi_last_error(m));
        CHECK(lumahip_encode_frames_host(one, rgb.data(), n, w, h, cur->sc, profile, plb.data(), st, mb.data()) == LUMAHIP_OK,
              "iteration %d: %s", it, lumahip_last_error(one));
        for (unsigned i = 0; i < n; i++)
            for (int p = 0; p < 3; p++)
                CHECK(pa[3 * i + p] == pb[3 * i + p], "iteration %d (%ux%u x%u, profile %d, cs %d): frame %u plane %d differs", it, w, h, n, profile,
                      cur->cs, i, p);
        // decode: sharded against single
        std::vector<std::vector<float>> da(n, std::vector<float>(n3, -7.0f)), db(n, std::vector<float>(n3, -7.0f));
        std::vector<float *> oa(n), ob(n);
        std::vector<const unsigned char *> cpl(3 * (size_t)n);
        for (unsigned i = 0; i < n; i++) {
            oa[i] = da[i].data();
            ob[i] = db[i].data();
            for (int p = 0; p < 3; p++)
                cpl[3 * i + p] = pa[3 * i + p].data();
        }
        CHECK(lumahip_multi_decode_frames_host(m, cpl.data(), st, n, w, h, profile, cur->sc, oa.data()) == LUMAHIP_OK, "iteration %d: %s", it,
              lumahip_multi_last_error(m));
        CHECK(lumahip_decode_frames_host(one, cpl.data(), st, n, w, h, profile, cur->sc, ob.data()) == LUMAHIP_OK, "iteration %d: %s", it,
              lumahip_last_error(one));
        for (unsigned i = 0; i < n; i++)
            CHECK(std::memcmp(da[i].data(), db[i].data(), n3 * sizeof(float)) == 0, "iteration %d: decoded frame %u differs", it, i);
        if (pin)
            for (unsigned i = 0; i < n; i++)
                (void)lumahip_host_unregister(one, fr[i].data());
        frames_total += n;
    }
    lumahip_multi_destroy(m);
    lumahip_destroy(one);
    std::printf("OK multi_stress: %d iterations, %d shards on device 0, %lu frames, planes and decoded floats identical to one context\n", iters,
                shards, frames_total);
    return 0;
}
