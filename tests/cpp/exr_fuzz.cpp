// tests/cpp/exr_fuzz.cpp -- mutation fuzzing of ExrInterface::readFrame under ASan + UBSan.
// Writes a valid file with ExrInterface::writeFrame, then corrupts it (byte flips, truncation, 32-bit size-field
// stomping, 64-bit chunk-offset stomping incl. values that wrap p + n) N times; every attempt must either decode or throw LumaException -- never crash, hang or overrun.
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
    // seeds: one file per compression the writer produces, plus any files named on the command line (the Python tests
    // pass PIZ and PXR24 files, which only the reader knows)
    const int nseeds = 4 + (argc > 3 ? argc - 3 : 0);
    for (int comp = 0; comp < nseeds; comp++) {
        const std::string bad = dir + "/bad.exr";
        std::string good = dir + "/good.exr";
        if (comp <= 3) {
            LumaFrame f(37, 21, 3);
            for (size_t i = 0; i < f.pixelCount(); i++)
                f.buffer[i] = (float)((i * 2654435761u) % 100000u) / 7.0f;
            ExrInterface::writeFrame(good.c_str(), f, (ExrInterface::Compression)comp, comp == 1);
        } else {
            good = argv[3 + comp - 4];
        }
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
            const unsigned kind = rnd() % 5;
            if (kind == 0) {
                m.resize(rnd() % m.size());  // truncation
            } else if (kind == 1) {
                for (unsigned k = 0, nflip = 1 + rnd() % 8; k < nflip; k++)
                    m[rnd() % m.size()] ^= (unsigned char)(1u << (rnd() % 8));
            } else if (kind == 2) {
                const size_t p = rnd() % (m.size() - 4);  // stomp a 32-bit field with an extreme value
                const unsigned vals[4] = {0xffffffffu, 0x7fffffffu, 0x80000000u, 0u};
                memcpy(&m[p], &vals[rnd() % 4], 4);
            } else if (kind == 3) {
                // stomp an aligned-or-not 64-bit field (the chunk offset table lives right after the header) with values
                // that wrap `offset + size` arithmetic or point just past / far past the end of the file
                const size_t p = rnd() % (m.size() - 8);
                const unsigned long long vals[6] = {0xfffffffffffffffeull, 0xfffffffffffffff8ull, 0x8000000000000000ull,
                                                    (unsigned long long)m.size(), (unsigned long long)m.size() - 1, 0x100000000ull};
                memcpy(&m[p], &vals[rnd() % 6], 8);
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
