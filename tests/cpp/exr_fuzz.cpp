// tests/cpp/exr_fuzz.cpp -- mutation fuzzing of ExrInterface::readFrame under ASan + UBSan.
// Writes a valid file with ExrInterface::writeFrame, then corrupts it (byte flips, truncation, size-field
// stomping) N times; every attempt must either decode or throw LumaException -- never crash, hang or overrun.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "exr_interface.h"

static unsigned rng_state = 12345u;
static unsigned rnd()
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const int iters = argc > 2 ? atoi(argv[2]) : 300;
    int decoded = 0, rejected = 0;
    for (int comp = 0; comp <= 3; comp++) {
        LumaFrame f(37, 21, 3);
        for (size_t i = 0; i < f.pixelCount(); i++)
            f.buffer[i] = (float)((i * 2654435761u) % 100000u) / 7.0f;
        const std::string good = dir + "/good.exr", bad = dir + "/bad.exr";
        ExrInterface::writeFrame(good.c_str(), f, (ExrInterface::Compression)comp, comp == 1);
        std::vector<unsigned char> data;
        {
            FILE *fp = fopen(good.c_str(), "rb");
            unsigned char buf[4096];
            size_t n;
            while ((n = fread(buf, 1, sizeof buf, fp)) > 0)
                data.insert(data.end(), buf, buf + n);
            fclose(fp);
        }
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> m = data;
            const unsigned kind = rnd() % 4;
            if (kind == 0) {
                m.resize(rnd() % m.size());  // truncation
            } else if (kind == 1) {
                for (unsigned k = 0, nflip = 1 + rnd() % 8; k < nflip; k++)
                    m[rnd() % m.size()] ^= (unsigned char)(1u << (rnd() % 8));
            } else if (kind == 2) {
                const size_t p = rnd() % (m.size() - 4);  // stomp a 32-bit field with an extreme value
                const unsigned vals[4] = {0xffffffffu, 0x7fffffffu, 0x80000000u, 0u};
                memcpy(&m[p], &vals[rnd() % 4], 4);
            } else {
                const size_t p = rnd() % m.size(), n = rnd() % 64;
                for (size_t k = p; k < m.size() && k < p + n; k++)
                    m[k] = (unsigned char)rnd();
            }
            FILE *fp = fopen(bad.c_str(), "wb");
            if (!m.empty())
                fwrite(m.data(), 1, m.size(), fp);
            fclose(fp);
            try {
                LumaFrame g;
                ExrInterface::readFrame(bad.c_str(), g);
                decoded++;
            } catch (const LumaException &) {
                rejected++;
            }
        }
    }
    printf("ok decoded=%d rejected=%d\n", decoded, rejected);
    return 0;
}
