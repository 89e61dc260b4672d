// tools/experiments/membench3.hip -- second access-pattern exploration (memory system only): units per thread in flight,
// XCD-aware tile order, workgroup size.  Same traffic as lh::k_encode (15 B/px), nt accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// UNITS: units (4 px x 2 rows) a thread has in flight per iteration (consecutive tiles); XCD: remap tiles so that
// workgroups of one XCD (blockIdx % 8) walk one contiguous eighth of the tile range
template <int UNITS, bool XCD, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_encshape(const float *src, unsigned char *y, unsigned char *u, unsigned char *v,
                                                    int w, int h, int nframes)
{
    constexpr int NW = BLOCK / 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int unitsX = w / 4, unitsY = h / 2, tilesX = (unitsX + 63) / 64, tilesY = (unitsY + NW - 1) / NW;
    const int tpf = tilesX * tilesY, total = tpf * nframes;
    const size_t cs = (size_t)w * h;
    const int G = gridDim.x;
    int b = blockIdx.x;
    for (int it = 0;; it++) {
        int tbase;
        if (XCD) {
            // logical order: xcd-major; each XCD owns [xcd*per, (xcd+1)*per)
            const int xcd = b & 7, slot = b >> 3, per = (total + 7) / 8, gx = (G + 7) / 8;
            tbase = xcd * per + (slot + it * gx) * UNITS;
            if (slot + it * gx >= (per + UNITS - 1) / UNITS) break;
            if (tbase >= total) break;
        } else {
            tbase = (b + it * G) * UNITS;
            if (tbase >= total) break;
        }
        v4f a[UNITS][6];
        bool ok[UNITS];
        int f[UNITS], ux[UNITS], uy[UNITS];
#pragma unroll
        for (int k = 0; k < UNITS; k++) {
            const int t = tbase + k;
            ok[k] = t < total;
            if (!ok[k]) continue;
            f[k] = t / tpf;
            const int r = t - f[k] * tpf, by = r / tilesX, bx = r - by * tilesX;
            ux[k] = bx * 64 + tx; uy[k] = by * NW + ty;
            ok[k] = ux[k] < unitsX && uy[k] < unitsY;
            if (!ok[k]) continue;
            const float *p = src + (size_t)f[k] * 3 * cs + (size_t)(2 * uy[k]) * w + (size_t)ux[k] * 4;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                a[k][2 * c] = __builtin_nontemporal_load((const v4f *)(p + c * cs));
                a[k][2 * c + 1] = __builtin_nontemporal_load((const v4f *)(p + c * cs + w));
            }
        }
#pragma unroll
        for (int k = 0; k < UNITS; k++) {
            if (!ok[k]) continue;
            unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) { q0 ^= __float_as_uint(a[k][j].x); q1 ^= __float_as_uint(a[k][j].y); q2 ^= __float_as_uint(a[k][j].z); q3 ^= __float_as_uint(a[k][j].w); }
            unsigned char *dy = y + (size_t)f[k] * (2 * cs) + (size_t)(2 * uy[k]) * (2 * w) + (size_t)ux[k] * 8;
            v2u s0 = {q0, q1}, s1 = {q2, q3};
            __builtin_nontemporal_store(s0, (v2u *)dy);
            __builtin_nontemporal_store(s1, (v2u *)(dy + 2 * w));
            __builtin_nontemporal_store(q0 ^ q2, (unsigned *)(u + (size_t)f[k] * (cs / 2) + (size_t)uy[k] * w + (size_t)ux[k] * 4));
            __builtin_nontemporal_store(q1 ^ q3, (unsigned *)(v + (size_t)f[k] * (cs / 2) + (size_t)uy[k] * w + (size_t)ux[k] * 4));
        }
    }
}

__global__ __launch_bounds__(256) void k_read(const v4f *in, float *out, size_t n)
{
    v4f acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += __builtin_nontemporal_load(in + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main()
{
    const int w = 3840, h = 2160, B = 20, NB = 4;
    const size_t cs = (size_t)w * h, n3 = 3 * cs;
    float *src; unsigned char *y, *u, *v;
    CK(hipMalloc(&src, NB * B * n3 * 4));
    CK(hipMalloc(&y, NB * B * cs * 2)); CK(hipMalloc(&u, NB * B * cs / 2)); CK(hipMalloc(&v, NB * B * cs / 2));
    CK(hipMemset(src, 1, NB * B * n3 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char *name, double bytes) {
        float best = 1e9, sum = 0; int n = 0;
        for (int rep = 0; rep < 3; rep++) for (int b = 0; b < NB; b++) {
            (void)hipEventRecord(e0); launch(b); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep) { sum += ms; n++; if (ms < best) best = ms; }
        }
        printf("%-44s avg %.4f ms best %.4f -> %.0f GB/s (%.3f)\n", name, sum / n, best, bytes / (sum / n) / 1e6, bytes / (sum / n) / 1e6 / 8000);
    };
#define RUN(UN, XC, BL, GRID) { char nm[96]; snprintf(nm, sizeof nm, "units=%d xcd=%d block=%d grid=%d", UN, XC, BL, GRID); \
    timeit([&](int b) { hipLaunchKernelGGL((k_encshape<UN, XC, BL>), dim3(GRID), dim3(BL), 0, 0, src + b * B * n3, y + b * B * cs * 2, u + b * B * cs / 2, v + b * B * cs / 2, w, h, B); }, nm, 15.0 * B * cs); }
    for (int g : {1024, 2048}) { char nm[64]; snprintf(nm, sizeof nm, "read-only float4 nt, grid %d", g);
        timeit([&](int b) { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, (const v4f *)(src + b * B * n3), (float *)y, (size_t)B * n3 / 4); }, nm, 4.0 * B * n3); }
    RUN(1, false, 256, 1024) RUN(1, false, 256, 2048) RUN(2, false, 256, 1024) RUN(2, false, 256, 2048) RUN(4, false, 256, 1024)
    RUN(1, true, 256, 1024) RUN(1, true, 256, 2048) RUN(2, true, 256, 2048)
    RUN(1, false, 512, 512) RUN(1, false, 512, 1024) RUN(1, false, 1024, 256) RUN(1, false, 1024, 512)
    return 0;
}
