// membench7.hip -- cache-policy bits on the encode traffic mix: loads {plain, nt, sc1, sc1 nt} x stores {plain, nt, sc1, sc0 sc1,
// sc1 nt, sc0 sc1 nt} (gfx950: sc0 / sc1 / nt select the scope / temporal hint of global_load / global_store).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/membench7.hip -o membench7
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

struct Args {
    const float *src; unsigned char *y, *u, *v;
    int w, h, tilesX, tilesY, tpf, total;
};

#define ST2(MOD, P, V) asm volatile("global_store_dwordx2 %0, %1, off " MOD : : "v"(P), "v"(V) : "memory")
#define ST1(MOD, P, V) asm volatile("global_store_dword %0, %1, off " MOD : : "v"(P), "v"(V) : "memory")

template <int LM, int SM>
__global__ __launch_bounds__(256) void k(const Args a)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const size_t cs = (size_t)a.w * a.h;
    for (int t = blockIdx.x; t < a.total; t += gridDim.x) {
        const int f = t / a.tpf, r = t - f * a.tpf, by = r / a.tilesX, bx = r - by * a.tilesX;
        const int ux = bx * 64 + tx, uy = by * NW + ty;
        if (ux * 4 >= a.w || uy * 2 >= a.h) continue;
        const float *p = a.src + (size_t)f * 3 * cs + (size_t)(2 * uy) * a.w + (size_t)ux * 4;
        v4f vv[6];
        const float *p0 = p, *p1 = p + a.w, *p2 = p + cs, *p3 = p + cs + a.w, *p4 = p + 2 * cs, *p5 = p + 2 * cs + a.w;
        // all six loads and their wait in ONE asm statement: the compiler must not touch the destination registers before
        // the data has landed
#define LOADS(MOD)                                                                                                       \
    asm volatile("global_load_dwordx4 %0, %6, off " MOD "\n global_load_dwordx4 %1, %7, off " MOD "\n"                   \
                 "global_load_dwordx4 %2, %8, off " MOD "\n global_load_dwordx4 %3, %9, off " MOD "\n"                   \
                 "global_load_dwordx4 %4, %10, off " MOD "\n global_load_dwordx4 %5, %11, off " MOD "\n s_waitcnt vmcnt(0)" \
                 : "=&v"(vv[0]), "=&v"(vv[1]), "=&v"(vv[2]), "=&v"(vv[3]), "=&v"(vv[4]), "=&v"(vv[5])                     \
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5)                                                 \
                 : "memory")
        if (LM == 0) { LOADS(""); }
        else if (LM == 1) { LOADS("nt"); }
        else if (LM == 2) { LOADS("sc1"); }
        else { LOADS("sc1 nt"); }
        unsigned q0 = t, q1 = tx, q2 = ty, q3 = f;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            q0 ^= __float_as_uint(vv[i].x); q1 ^= __float_as_uint(vv[i].y); q2 ^= __float_as_uint(vv[i].z); q3 ^= __float_as_uint(vv[i].w);
        }
        const size_t ys = (size_t)a.w * 2, us = (size_t)a.w;
        unsigned char *d0 = a.y + (size_t)f * ys * a.h + (size_t)(2 * uy) * ys + (size_t)ux * 8;
        unsigned char *d0b = d0 + ys;
        unsigned char *d1 = a.u + (size_t)f * us * (a.h / 2) + (size_t)uy * us + (size_t)ux * 4;
        unsigned char *d2 = a.v + (size_t)f * us * (a.h / 2) + (size_t)uy * us + (size_t)ux * 4;
        v2u s0 = {q0, q1}, s1 = {q2, q3};
        unsigned c1 = q0 ^ q2, c2 = q1 ^ q3;
#define STORES(MOD) ST2(MOD, d0, s0); ST2(MOD, d0b, s1); ST1(MOD, d1, c1); ST1(MOD, d2, c2)
        if (SM == 0) { STORES(""); }
        else if (SM == 1) { STORES("nt"); }
        else if (SM == 2) { STORES("sc1"); }
        else if (SM == 3) { STORES("sc0 sc1"); }
        else if (SM == 4) { STORES("sc1 nt"); }
        else { STORES("sc0 sc1 nt"); }
    }
}

template <int LM, int SM>
static void run(const char *name, const Args &a, double bytes, int round)
{
    std::vector<float> ms;
    for (int rep = 0; rep < 8; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<LM, SM><<<2048, 256>>>(a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep >= 2) ms.push_back(t);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    std::sort(ms.begin(), ms.end());
    printf("round %d %-34s : med %.4f min %.4f ms -> %.0f GB/s\n", round, name, ms[ms.size() / 2], ms[0], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
}

int main()
{
    const int W = 3840, H = 2160, B = 20;
    const size_t cs = (size_t)W * H;
    Args a{};
    float *src; hipMalloc(&src, cs * 3 * B * 4); hipMemset(src, 1, cs * 3 * B * 4);
    hipMalloc(&a.y, cs * 2 * B); hipMalloc(&a.u, cs / 2 * B); hipMalloc(&a.v, cs / 2 * B);
    a.src = src; a.w = W; a.h = H;
    a.tilesX = (W / 4 + 63) / 64; a.tilesY = (H / 2 + 3) / 4; a.tpf = a.tilesX * a.tilesY; a.total = a.tpf * B;
    const double bytes = 15.0 * cs * B;
    for (int round = 0; round < 3; round++) {
        run<1, 1>("load nt      | store nt", a, bytes, round);
        run<0, 0>("load plain   | store plain", a, bytes, round);
        run<1, 0>("load nt      | store plain", a, bytes, round);
        run<0, 1>("load plain   | store nt", a, bytes, round);
        run<1, 2>("load nt      | store sc1", a, bytes, round);
        run<1, 3>("load nt      | store sc0 sc1", a, bytes, round);
        run<1, 4>("load nt      | store sc1 nt", a, bytes, round);
        run<1, 5>("load nt      | store sc0 sc1 nt", a, bytes, round);
        run<2, 1>("load sc1     | store nt", a, bytes, round);
        run<3, 1>("load sc1 nt  | store nt", a, bytes, round);
        run<3, 5>("load sc1 nt  | store sc0 sc1 nt", a, bytes, round);
        run<3, 4>("load sc1 nt  | store sc1 nt", a, bytes, round);
    }
    return 0;
}
