// tools/experiments/membench.hip -- what the MEMORY SYSTEM alone delivers for the encode kernel's access pattern.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/experiments/membench.hip && /tmp/membench
// (a) float4 copy (read N, write N)                  -> the chip's achievable streaming bandwidth
// (b) encode-shaped traffic: per thread 6 x 16 B loads (3 planes x 2 rows), 2 x 8 B + 2 x 4 B stores,
//     persistent workgroups, same tile order as lh::k_encode, trivial arithmetic -> ceiling for k_encode
// (c) decode-shaped traffic: 2 x 8 B + 2 x 4 B loads, 6 x 16 B stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4 *in, float4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

template <bool PREFETCH>
__global__ __launch_bounds__(256) void k_encshape(const float *src, unsigned char *y, unsigned char *u, unsigned char *v,
                                                  int w, int h, int nframes)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int unitsX = w / 4, unitsY = h / 2, tilesX = (unitsX + 63) / 64, tilesY = (unitsY + 3) / 4;
    const int tpf = tilesX * tilesY, total = tpf * nframes;
    const size_t cs = (size_t)w * h;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int f = t / tpf, r = t - f * tpf, by = r / tilesX, bx = r - by * tilesX;
        const int ux = bx * 64 + tx, uy = by * 4 + ty;
        if (ux >= unitsX || uy >= unitsY) continue;
        const float *p = src + (size_t)f * 3 * cs + (size_t)(2 * uy) * w + (size_t)ux * 4;
        float4 a[6];
        for (int c = 0; c < 3; c++) { a[2 * c] = *(const float4 *)(p + c * cs); a[2 * c + 1] = *(const float4 *)(p + c * cs + w); }
        uint2 y0, y1; unsigned uu, vv;
        y0.x = __float_as_uint(a[0].x) ^ __float_as_uint(a[2].y); y0.y = __float_as_uint(a[4].z) ^ __float_as_uint(a[0].w);
        y1.x = __float_as_uint(a[1].x) ^ __float_as_uint(a[3].y); y1.y = __float_as_uint(a[5].z) ^ __float_as_uint(a[1].w);
        uu = __float_as_uint(a[2].x) ^ __float_as_uint(a[3].w); vv = __float_as_uint(a[4].x) ^ __float_as_uint(a[5].w);
        unsigned char *dy = y + (size_t)f * (2 * cs) + (size_t)(2 * uy) * (2 * w) + (size_t)ux * 8;
        *(uint2 *)dy = y0; *(uint2 *)(dy + 2 * w) = y1;
        *(unsigned *)(u + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * 4) = uu;
        *(unsigned *)(v + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * 4) = vv;
    }
}

__global__ __launch_bounds__(256) void k_decshape(float *dst, const unsigned char *y, const unsigned char *u, const unsigned char *v,
                                                  int w, int h, int nframes)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int unitsX = w / 4, unitsY = h / 2, tilesX = (unitsX + 63) / 64, tilesY = (unitsY + 3) / 4;
    const int tpf = tilesX * tilesY, total = tpf * nframes;
    const size_t cs = (size_t)w * h;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int f = t / tpf, r = t - f * tpf, by = r / tilesX, bx = r - by * tilesX;
        const int ux = bx * 64 + tx, uy = by * 4 + ty;
        if (ux >= unitsX || uy >= unitsY) continue;
        const unsigned char *sy = y + (size_t)f * (2 * cs) + (size_t)(2 * uy) * (2 * w) + (size_t)ux * 8;
        const uint2 y0 = *(const uint2 *)sy, y1 = *(const uint2 *)(sy + 2 * w);
        const unsigned uu = *(const unsigned *)(u + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * 4);
        const unsigned vv = *(const unsigned *)(v + (size_t)f * (cs / 2) + (size_t)uy * w + (size_t)ux * 4);
        float *p = dst + (size_t)f * 3 * cs + (size_t)(2 * uy) * w + (size_t)ux * 4;
        for (int c = 0; c < 3; c++) {
            const float s = (float)c;
            *(float4 *)(p + c * cs) = make_float4(__uint_as_float(y0.x) + s, __uint_as_float(y0.y), __uint_as_float(uu), __uint_as_float(vv));
            *(float4 *)(p + c * cs + w) = make_float4(__uint_as_float(y1.x) + s, __uint_as_float(y1.y), __uint_as_float(uu), __uint_as_float(vv));
        }
    }
}

int main()
{
    const int w = 3840, h = 2160, B = 20, NB = 4;
    const size_t cs = (size_t)w * h, n3 = 3 * cs;
    float *src, *dst; unsigned char *y, *u, *v;
    CK(hipMalloc(&src, NB * B * n3 * 4)); CK(hipMalloc(&dst, NB * B * n3 * 4));
    CK(hipMalloc(&y, NB * B * cs * 2)); CK(hipMalloc(&u, NB * B * cs / 2)); CK(hipMalloc(&v, NB * B * cs / 2));
    CK(hipMemset(src, 1, NB * B * n3 * 4)); CK(hipMemset(y, 1, NB * B * cs * 2)); CK(hipMemset(u, 1, NB * B * cs / 2)); CK(hipMemset(v, 1, NB * B * cs / 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char *name, double bytes) {
        float best = 1e9, sum = 0; int n = 0;
        for (int rep = 0; rep < 3; rep++) for (int b = 0; b < NB; b++) {
            (void)hipEventRecord(e0); launch(b); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep) { sum += ms; n++; if (ms < best) best = ms; }
        }
        printf("%-46s avg %.4f ms  best %.4f ms  -> %.0f GB/s avg (%.3f of 8 TB/s)\n", name, sum / n, best, bytes / (sum / n) / 1e6, bytes / (sum / n) / 1e6 / 8000);
    };
    const size_t ncopy = (size_t)B * n3 / 4;  // float4 count of one batch
    for (int grid : {1024, 2048, 4096}) {
        char nm[64]; snprintf(nm, sizeof nm, "float4 copy, grid %d", grid);
        timeit([&](int b) { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const float4 *)(src + b * B * n3), (float4 *)(dst + b * B * n3), ncopy); }, nm, 2.0 * B * n3 * 4);
    }
    for (int grid : {1024, 1536, 2048}) {
        char nm[64]; snprintf(nm, sizeof nm, "encode-shaped traffic (15 B/px), grid %d", grid);
        timeit([&](int b) { hipLaunchKernelGGL(k_encshape<false>, dim3(grid), dim3(256), 0, 0, src + b * B * n3, y + b * B * cs * 2, u + b * B * cs / 2, v + b * B * cs / 2, w, h, B); }, nm, 15.0 * B * cs);
        snprintf(nm, sizeof nm, "decode-shaped traffic (15 B/px), grid %d", grid);
        timeit([&](int b) { hipLaunchKernelGGL(k_decshape, dim3(grid), dim3(256), 0, 0, dst + b * B * n3, y + b * B * cs * 2, u + b * B * cs / 2, v + b * B * cs / 2, w, h, B); }, nm, 15.0 * B * cs);
    }
    return 0;
}
