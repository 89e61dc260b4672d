// tools/experiments/hostfed_lab.hip -- where the time of a host-fed frame goes (-> profiles/r03_hostfed_lab.txt): CPU copy rate of a
// 4K float frame (99.5 MB) from pageable memory into a pinned buffer with 1..16 threads, H2D / D2H DMA rates from
// hipHostMalloc'd and hipHostRegister'd memory, and the two overlapped (copy of chunk k+1 while chunk k is in flight).
//   hipcc -O2 --offload-arch=gfx950 tools/experiments/hostfed_lab.hip -o /tmp/hostfed_lab -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(unsigned char *dst, const unsigned char *src, size_t n, int threads)
{
    if (threads <= 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    size_t per = ((n + threads - 1) / threads + 4095) & ~(size_t)4095;
    for (int t = 1; t < threads; t++) {
        size_t a = std::min(n, t * per), b = std::min(n, (t + 1) * per);
        if (a < b)
            th.emplace_back([=]() { memcpy(dst + a, src + a, b - a); });
    }
    memcpy(dst, src, std::min(n, per));
    for (auto &t : th)
        t.join();
}

int main()
{
    const size_t N = (size_t)3 * 3840 * 2160 * 4;
    const int F = 6;
    std::vector<unsigned char *> page(F);
    for (auto &p : page) {
        p = (unsigned char *)malloc(N);
        memset(p, 1, N);
    }
    unsigned char *pin = nullptr, *pin2 = nullptr, *dev = nullptr;
    hipHostMalloc((void **)&pin, N, hipHostMallocDefault);
    hipHostMalloc((void **)&pin2, N, hipHostMallocDefault);
    hipMalloc((void **)&dev, N);
    memset(pin, 2, N);
    memset(pin2, 2, N);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("frame = %.1f MB; hardware threads %u\n", N / 1e6, std::thread::hardware_concurrency());
    for (int T : {1, 2, 3, 4, 6, 8, 12, 16}) {
        par_copy(pin, page[0], N, T);
        double t0 = now();
        for (int i = 0; i < 2 * F; i++)
            par_copy(pin, page[i % F], N, T);
        double dt = (now() - t0) / (2 * F);
        printf("pageable -> pinned, %2d threads (spawned per call): %.2f ms  %.1f GB/s\n", T, dt * 1e3, N / dt / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < 8; i++)
            hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double dt = (now() - t0) / 8;
        printf("H2D from hipHostMalloc memory: %.2f ms  %.1f GB/s\n", dt * 1e3, N / dt / 1e9);
    }
    {
        double t0 = now();
        for (int i = 0; i < 8; i++)
            hipMemcpyAsync(pin2, dev, N / 4, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double dt = (now() - t0) / 8;
        printf("D2H (24.9 MB) to hipHostMalloc memory: %.2f ms  %.1f GB/s\n", dt * 1e3, N / 4 / dt / 1e9);
    }
    {
        hipHostRegister(page[0], N, hipHostRegisterDefault);
        hipMemcpyAsync(dev, page[0], N, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < 8; i++)
            hipMemcpyAsync(dev, page[0], N, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double dt = (now() - t0) / 8;
        printf("H2D from hipHostRegister'd malloc memory: %.2f ms  %.1f GB/s\n", dt * 1e3, N / dt / 1e9);
        hipHostUnregister(page[0]);
    }
    // chunked pipeline: CPU copy of chunk k+1 into one half while chunk k is DMA'd from the other
    for (size_t chunk : {(size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20})
        for (int T : {1, 4, 8}) {
            hipEvent_t ev[2];
            hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
            hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
            bool pend[2] = {false, false};
            double t0 = now();
            const int reps = 6;
            for (int f = 0; f < reps; f++) {
                int k = 0;
                for (size_t done = 0; done < N; done += chunk, k++) {
                    unsigned char *h = (k & 1) ? pin2 : pin;
                    if (pend[k & 1])
                        hipEventSynchronize(ev[k & 1]);
                    size_t n = std::min(chunk, N - done);
                    par_copy(h, page[f % F] + done, n, T);
                    hipMemcpyAsync(dev + done, h, n, hipMemcpyHostToDevice, s);
                    hipEventRecord(ev[k & 1], s);
                    pend[k & 1] = true;
                }
                hipStreamSynchronize(s);
            }
            double dt = (now() - t0) / reps;
            printf("staged H2D, chunk %2zu MiB, %d copy threads: %.2f ms per frame  %.1f GB/s  (%.0f Mpixel/s)\n", chunk >> 20, T, dt * 1e3,
                   N / dt / 1e9, 3840.0 * 2160 / dt / 1e6);
        }
    return 0;
}
