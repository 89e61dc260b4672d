// membench6.hip -- encode traffic mix (no arithmetic) with a general workgroup tile: WX waves side by side in x, WY waves
// stacked in y (lh::k_encode uses WX=1, WY=4).  Does a wider contiguous footprint per workgroup help the HBM mix?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/membench6.hip -o membench6
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

struct Args {
    const float *src; unsigned char *y, *u, *v;
    int w, h, unitsX, unitsY, tilesX, tilesY, tpf, total, wx, wy;
};

__global__ __launch_bounds__(1024) void k(const Args a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wxi = wave % a.wx, wyi = wave / a.wx;
    const size_t cs = (size_t)a.w * a.h;
    for (int t = blockIdx.x; t < a.total; t += gridDim.x) {
        const int f = t / a.tpf, r = t - f * a.tpf, by = r / a.tilesX, bx = r - by * a.tilesX;
        const int ux = (bx * a.wx + wxi) * 64 + lane, uy = by * a.wy + wyi;
        if (ux >= a.unitsX || uy >= a.unitsY) continue;
        unsigned q0 = t, q1 = lane, q2 = wave, q3 = f;
        const float *p = a.src + (size_t)f * 3 * cs + (size_t)(2 * uy) * a.w + (size_t)ux * 4;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p + c * cs + (size_t)rr * a.w));
                q0 ^= __float_as_uint(v.x); q1 ^= __float_as_uint(v.y); q2 ^= __float_as_uint(v.z); q3 ^= __float_as_uint(v.w);
            }
        const size_t ys = (size_t)a.w * 2, us = (size_t)a.w;
        unsigned char *d0 = a.y + (size_t)f * ys * a.h + (size_t)(2 * uy) * ys + (size_t)ux * 8;
        unsigned char *d1 = a.u + (size_t)f * us * (a.h / 2) + (size_t)uy * us + (size_t)ux * 4;
        unsigned char *d2 = a.v + (size_t)f * us * (a.h / 2) + (size_t)uy * us + (size_t)ux * 4;
        v2u s0 = {q0, q1}, s1 = {q2, q3};
        __builtin_nontemporal_store(s0, reinterpret_cast<v2u *>(d0));
        __builtin_nontemporal_store(s1, reinterpret_cast<v2u *>(d0 + ys));
        __builtin_nontemporal_store(q0 ^ q2, reinterpret_cast<unsigned *>(d1));
        __builtin_nontemporal_store(q1 ^ q3, reinterpret_cast<unsigned *>(d2));
    }
}

int main()
{
    const int W = 3840, H = 2160, B = 20;
    const size_t cs = (size_t)W * H;
    Args a{};
    float *src; hipMalloc(&src, cs * 3 * B * 4); hipMemset(src, 1, cs * 3 * B * 4);
    hipMalloc(&a.y, cs * 2 * B); hipMalloc(&a.u, cs / 2 * B); hipMalloc(&a.v, cs / 2 * B);
    a.src = src; a.w = W; a.h = H; a.unitsX = W / 4; a.unitsY = H / 2;
    const double bytes = 15.0 * cs * B;
    struct G { int wx, wy, percu; };
    const G geoms[] = {{1, 4, 8}, {1, 4, 5}, {4, 1, 8}, {4, 1, 5}, {5, 1, 6}, {5, 1, 4}, {3, 1, 10}, {3, 1, 7}, {2, 2, 8}, {15, 1, 2}, {15, 1, 1},
                       {5, 2, 3}, {1, 8, 4}, {1, 2, 16}, {1, 1, 32}, {1, 1, 20}};
    for (int round = 0; round < 3; round++)
        for (const G &g : geoms) {
            a.wx = g.wx; a.wy = g.wy;
            a.tilesX = (a.unitsX + 64 * g.wx - 1) / (64 * g.wx); a.tilesY = (a.unitsY + g.wy - 1) / g.wy;
            a.tpf = a.tilesX * a.tilesY; a.total = a.tpf * B;
            const int threads = 64 * g.wx * g.wy, grid = std::min(256 * g.percu, a.total);
            std::vector<float> ms;
            for (int rep = 0; rep < 8; rep++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                k<<<grid, threads>>>(a);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (rep >= 2) ms.push_back(t);
                hipEventDestroy(e0); hipEventDestroy(e1);
            }
            std::sort(ms.begin(), ms.end());
            printf("round %d WX=%2d WY=%d threads %4d wg/CU %2d : med %.4f min %.4f ms -> %.0f GB/s\n", round, g.wx, g.wy, threads, g.percu,
                   ms[ms.size() / 2], ms[0], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
        }
    return 0;
}
