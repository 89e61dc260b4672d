// tools/experiments/half_stage_lab.cpp -- would uploading binary16 instead of binary32 pay for host frames that hold halves?
// (The reference's LumaFrame is float, but its EXR reader fills it with widened halves: src/exr_interface.cpp:77-146.)
// Measures, for one 3840x2160x3 float frame in pageable memory: (a) plain memcpy into a pinned chunk, (b) float -> half
// conversion with a round-trip exactness check into a pinned chunk of half the size, each with 1..16 threads; (c) the DMA of
// 99.5 MB against 49.8 MB.   g++ -O2 -mavx2 -mf16c -pthread half_stage_lab.cpp -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool conv(const float *src, unsigned short *dst, size_t n)
{
    __m256 bad = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(src + i);
        const __m128i h = _mm256_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const __m256 b = _mm256_cvtph_ps(h);
        bad = _mm256_or_ps(bad, _mm256_xor_ps(b, v));
        _mm_storeu_si128((__m128i *)(dst + i), h);
    }
    return _mm256_testz_si256(_mm256_castps_si256(bad), _mm256_castps_si256(bad)) != 0;
}

int main()
{
    const size_t n = (size_t)3 * 3840 * 2160;
    std::vector<float> frame(n);
    for (size_t i = 0; i < n; i++)
        frame[i] = (float)((i * 2654435761u) & 0x3ff) * 0.25f;   // halves
    void *pin = nullptr;
    if (hipHostMalloc(&pin, n * 4, hipHostMallocDefault) != hipSuccess)
        return 1;
    memset(pin, 0, n * 4);
    void *dev = nullptr;
    (void)hipMalloc(&dev, n * 4);
    for (int nt : {1, 2, 3, 4, 6, 8, 12, 16}) {
        double best[2] = {1e9, 1e9};
        bool ok = true;
        for (int rep = 0; rep < 5; rep++) {
            for (int mode = 0; mode < 2; mode++) {
                const double t0 = now();
                std::vector<std::thread> th;
                std::vector<char> oks(nt, 1);
                for (int t = 0; t < nt; t++)
                    th.emplace_back([&, t]() {
                        const size_t a = n / nt * t / 8 * 8, b = (t == nt - 1) ? n : n / nt * (t + 1) / 8 * 8;
                        if (mode == 0)
                            memcpy((float *)pin + a, frame.data() + a, (b - a) * 4);
                        else
                            oks[t] = conv(frame.data() + a, (unsigned short *)pin + a, b - a);
                    });
                for (auto &x : th)
                    x.join();
                const double dt = now() - t0;
                if (dt < best[mode])
                    best[mode] = dt;
                for (char c : oks)
                    ok = ok && c;
            }
        }
        printf("%2d threads: memcpy %.2f ms (%.1f GB/s read)   float->half + check %.2f ms (%.1f GB/s read) exact=%d\n", nt, best[0] * 1e3,
               n * 4 / best[0] / 1e9, best[1] * 1e3, n * 4 / best[1] / 1e9, (int)ok);
    }
    for (size_t bytes : {n * 4, n * 2}) {
        double best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            const double t0 = now();
            (void)hipMemcpy(dev, pin, bytes, hipMemcpyHostToDevice);
            const double dt = now() - t0;
            if (dt < best)
                best = dt;
        }
        printf("DMA of %.1f MB: %.2f ms (%.1f GB/s)\n", bytes / 1e6, best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
