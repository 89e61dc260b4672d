// tools/experiments/membench2.hip -- access-pattern exploration for the encode kernel (memory system only, trivial math).
// Variants: waves-along-x per workgroup (WX: 1 = 256x8 px tiles, 2 = 512x4, 4 = 1024x2), non-temporal
// loads/stores (NT), pixels per thread per row (VW 4 or 8), workgroup size 256.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ float4 ld16(const float *p)
{
    v4f t = NT ? __builtin_nontemporal_load((const v4f *)p) : *(const v4f *)p;
    return make_float4(t.x, t.y, t.z, t.w);
}
template <bool NT> __device__ __forceinline__ void st(uint4 *p, uint4 v)
{
    v4u t = {v.x, v.y, v.z, v.w};
    if (NT) __builtin_nontemporal_store(t, (v4u *)p); else *(v4u *)p = t;
}
template <bool NT> __device__ __forceinline__ void st(uint2 *p, uint2 v)
{
    v2u t = {v.x, v.y};
    if (NT) __builtin_nontemporal_store(t, (v2u *)p); else *(v2u *)p = t;
}
template <bool NT> __device__ __forceinline__ void st(unsigned *p, unsigned v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int WX, bool NT, int VW>
__global__ __launch_bounds__(256) void k_encshape(const float *src, unsigned char *y, unsigned char *u, unsigned char *v,
                                                  int w, int h, int nframes)
{
    constexpr int WY = 4 / WX;               // waves along y
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wx = wave % WX, wy = wave / WX;
    const int unitsX = w / VW, unitsY = h / 2;
    const int tilesX = (unitsX + 64 * WX - 1) / (64 * WX), tilesY = (unitsY + WY - 1) / WY;
    const int tpf = tilesX * tilesY, total = tpf * nframes;
    const size_t cs = (size_t)w * h;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int f = t / tpf, r = t - f * tpf, by = r / tilesX, bx = r - by * tilesX;
        const int ux = (bx * WX + wx) * 64 + lane, uy = by * WY + wy;
        if (ux >= unitsX || uy >= unitsY) continue;
        const float *p = src + (size_t)f * 3 * cs + (size_t)(2 * uy) * w + (size_t)ux * VW;
        unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int q = 0; q < VW / 4; q++) {
                    float4 a = ld16<NT>(p + c * cs + rr * w + 4 * q);
                    acc0 ^= __float_as_uint(a.x); acc1 ^= __float_as_uint(a.y); acc2 ^= __float_as_uint(a.z); acc3 ^= __float_as_uint(a.w);
                }
        unsigned char *dy = y + (size_t)f * (2 * cs) + (size_t)(2 * uy) * (2 * w) + (size_t)ux * VW * 2;
        unsigned char *du = u + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * VW;
        unsigned char *dv = v + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * VW;
        if (VW == 4) {
            st<NT>((uint2 *)dy, make_uint2(acc0, acc1)); st<NT>((uint2 *)(dy + 2 * w), make_uint2(acc2, acc3));
            st<NT>((unsigned *)du, acc0 ^ acc2); st<NT>((unsigned *)dv, acc1 ^ acc3);
        } else {
            st<NT>((uint4 *)dy, make_uint4(acc0, acc1, acc2, acc3)); st<NT>((uint4 *)(dy + 2 * w), make_uint4(acc2, acc3, acc0, acc1));
            st<NT>((uint2 *)du, make_uint2(acc0 ^ acc2, acc1)); st<NT>((uint2 *)dv, make_uint2(acc1 ^ acc3, acc0));
        }
    }
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
#define RUN(WX, NT, VW, GRID) { char nm[96]; snprintf(nm, sizeof nm, "WX=%d NT=%d VW=%d grid=%d", WX, NT, VW, GRID); \
    timeit([&](int b) { hipLaunchKernelGGL((k_encshape<WX, NT, VW>), dim3(GRID), dim3(256), 0, 0, src + b * B * n3, y + b * B * cs * 2, u + b * B * cs / 2, v + b * B * cs / 2, w, h, B); }, nm, 15.0 * B * cs); }
    RUN(1, false, 4, 1024) RUN(2, false, 4, 1024) RUN(4, false, 4, 1024)
    RUN(1, true, 4, 1024) RUN(2, true, 4, 1024) RUN(4, true, 4, 1024)
    RUN(1, false, 8, 1024) RUN(2, false, 8, 1024) RUN(4, false, 8, 1024)
    RUN(1, true, 8, 1024) RUN(2, true, 8, 1024) RUN(4, true, 8, 1024)
    RUN(4, true, 4, 512) RUN(4, true, 4, 2048) RUN(4, true, 8, 512) RUN(4, true, 8, 2048) RUN(1, true, 8, 2048) RUN(1, true, 4, 2048)
    return 0;
}
