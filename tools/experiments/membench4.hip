// tools/experiments/membench4.hip -- decode-shaped traffic (3 B read + 12 B written per pixel), memory system only:
// plain vs non-temporal accesses, 4 vs 8 pixels per thread per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT, typename T> __device__ __forceinline__ T ld(const T *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT, typename T> __device__ __forceinline__ void st(T *p, T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NTL, bool NTS, int VW>
__global__ __launch_bounds__(256) void k_decshape(float *dst, const unsigned char *y, const unsigned char *u, const unsigned char *v,
                                                  int w, int h, int nframes)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int unitsX = w / VW, unitsY = h / 2, tilesX = (unitsX + 63) / 64, tilesY = (unitsY + 3) / 4;
    const int tpf = tilesX * tilesY, total = tpf * nframes;
    const size_t cs = (size_t)w * h;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int f = t / tpf, r = t - f * tpf, by = r / tilesX, bx = r - by * tilesX;
        const int ux = bx * 64 + tx, uy = by * 4 + ty;
        if (ux >= unitsX || uy >= unitsY) continue;
        const unsigned char *sy = y + (size_t)f * (2 * cs) + (size_t)(2 * uy) * (2 * w) + (size_t)ux * VW * 2;
        const unsigned char *su = u + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * VW;
        const unsigned char *sv = v + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * VW;
        unsigned a0, a1, a2, a3;
        if (VW == 4) {
            v2u y0 = ld<NTL>((const v2u *)sy), y1 = ld<NTL>((const v2u *)(sy + 2 * w));
            a0 = y0.x ^ ld<NTL>((const unsigned *)su); a1 = y0.y; a2 = y1.x ^ ld<NTL>((const unsigned *)sv); a3 = y1.y;
        } else {
            v4u y0 = ld<NTL>((const v4u *)sy), y1 = ld<NTL>((const v4u *)(sy + 2 * w));
            v2u uu = ld<NTL>((const v2u *)su), vv = ld<NTL>((const v2u *)sv);
            a0 = y0.x ^ y0.z ^ uu.x; a1 = y0.y ^ y0.w ^ uu.y; a2 = y1.x ^ y1.z ^ vv.x; a3 = y1.y ^ y1.w ^ vv.y;
        }
        float *p = dst + (size_t)f * 3 * cs + (size_t)(2 * uy) * w + (size_t)ux * VW;
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 2; rr++)
                for (int q = 0; q < VW / 4; q++) {
                    v4f o = {__uint_as_float(a0) + c, __uint_as_float(a1) + rr, __uint_as_float(a2) + q, __uint_as_float(a3)};
                    st<NTS>((v4f *)(p + c * cs + rr * w + 4 * q), o);
                }
    }
}

int main()
{
    const int w = 3840, h = 2160, B = 20, NB = 4;
    const size_t cs = (size_t)w * h, n3 = 3 * cs;
    float *dst; unsigned char *y, *u, *v;
    CK(hipMalloc(&dst, NB * B * n3 * 4));
    CK(hipMalloc(&y, NB * B * cs * 2)); CK(hipMalloc(&u, NB * B * cs / 2)); CK(hipMalloc(&v, NB * B * cs / 2));
    CK(hipMemset(y, 1, NB * B * cs * 2)); CK(hipMemset(u, 1, NB * B * cs / 2)); CK(hipMemset(v, 1, NB * B * cs / 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char *name, double bytes) {
        float best = 1e9, sum = 0; int n = 0;
        for (int rep = 0; rep < 3; rep++) for (int b = 0; b < NB; b++) {
            (void)hipEventRecord(e0); launch(b); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep) { sum += ms; n++; if (ms < best) best = ms; }
        }
        printf("%-44s avg %.4f ms best %.4f -> %.0f GB/s (%.3f)\n", name, sum / n, best, bytes / (sum / n) / 1e6, bytes / (sum / n) / 1e6 / 8000);
    };
#define RUN(NTL, NTS, VW, GRID) { char nm[96]; snprintf(nm, sizeof nm, "ntload=%d ntstore=%d VW=%d grid=%d", NTL, NTS, VW, GRID); \
    timeit([&](int b) { hipLaunchKernelGGL((k_decshape<NTL, NTS, VW>), dim3(GRID), dim3(256), 0, 0, dst + b * B * n3, y + b * B * cs * 2, u + b * B * cs / 2, v + b * B * cs / 2, w, h, B); }, nm, 15.0 * B * cs); }
    RUN(false, false, 4, 2048) RUN(true, false, 4, 2048) RUN(false, true, 4, 2048) RUN(true, true, 4, 2048)
    RUN(false, false, 8, 2048) RUN(true, true, 8, 2048) RUN(true, true, 4, 1024) RUN(true, true, 8, 1024) RUN(false, false, 4, 1024)
    return 0;
}
