// membench5.hip -- where does the encode traffic mix lose bandwidth?  Same tile walk as lh::k_encode (64 units x NW
// rows per workgroup, persistent), 20 x 3840x2160 frames: (R) the six 16-byte loads only, (W) the four stores only,
// (RW) both (= lh::k_encode_traffic_probe), plus store-pattern variants.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/membench5.hip -o membench5
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct Args {
    const float *src; unsigned char *y, *u, *v;
    int w, h, nframes, tilesX, tilesY, tpf, total;
};

// MODE bit0: loads, bit1: stores.  SP: store pattern 0 = as the encode kernel (Y 8 B x2 rows, U 4 B, V 4 B per thread),
// 1 = lane pairs exchange so that even lanes store 16 B of Y row 0 / odd lanes 16 B of Y row 1 and U/V 8 B each from
// half the lanes (same bytes, half the store instructions per lane, 16-byte stores)
template <int MODE, int SP, bool NT>
__global__ __launch_bounds__(256) void k(const Args a)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const size_t cs = (size_t)a.w * a.h;
    unsigned acc = 0;
    for (int t = blockIdx.x; t < a.total; t += gridDim.x) {
        const int f = t / a.tpf, r = t - f * a.tpf, by = r / a.tilesX, bx = r - by * a.tilesX;
        const int ux = bx * 64 + tx, uy = by * NW + ty;
        if (ux * 4 >= a.w || uy * 2 >= a.h) continue;
        unsigned q0 = t, q1 = tx, q2 = ty, q3 = f;
        if (MODE & 1) {
            const float *p = a.src + (size_t)f * 3 * cs + (size_t)(2 * uy) * a.w + (size_t)ux * 4;
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const v4f *pp = reinterpret_cast<const v4f *>(p + c * cs + (size_t)rr * a.w);
                    v4f v = NT ? __builtin_nontemporal_load(pp) : *pp;
                    q0 ^= __float_as_uint(v.x); q1 ^= __float_as_uint(v.y); q2 ^= __float_as_uint(v.z); q3 ^= __float_as_uint(v.w);
                }
        }
        if (MODE & 2) {
            const size_t ys = (size_t)a.w * 2, us = (size_t)a.w;
            unsigned char *d0 = a.y + (size_t)f * ys * a.h + (size_t)(2 * uy) * ys + (size_t)ux * 8;
            unsigned char *d1 = a.u + (size_t)f * us * (a.h / 2) + (size_t)uy * us + (size_t)ux * 4;
            unsigned char *d2 = a.v + (size_t)f * us * (a.h / 2) + (size_t)uy * us + (size_t)ux * 4;
            if (SP == 0) {
                v2u s0 = {q0, q1}, s1 = {q2, q3};
                if (NT) {
                    __builtin_nontemporal_store(s0, reinterpret_cast<v2u *>(d0));
                    __builtin_nontemporal_store(s1, reinterpret_cast<v2u *>(d0 + ys));
                    __builtin_nontemporal_store(q0 ^ q2, reinterpret_cast<unsigned *>(d1));
                    __builtin_nontemporal_store(q1 ^ q3, reinterpret_cast<unsigned *>(d2));
                } else {
                    *reinterpret_cast<v2u *>(d0) = s0; *reinterpret_cast<v2u *>(d0 + ys) = s1;
                    *reinterpret_cast<unsigned *>(d1) = q0 ^ q2; *reinterpret_cast<unsigned *>(d2) = q1 ^ q3;
                }
            } else {
                // exchange with the lane partner (tx ^ 1): even lane keeps row 0 of both, odd lane row 1 of both
                const bool odd = tx & 1;
                const unsigned g0 = odd ? q0 : q2, g1 = odd ? q1 : q3;       // what I give away
                const unsigned r0 = __shfl_xor(g0, 1, 64), r1 = __shfl_xor(g1, 1, 64);
                v4u s = odd ? v4u{r0, r1, q2, q3} : v4u{q0, q1, r0, r1};
                unsigned char *dy = d0 + (odd ? ys : 0) - (odd ? 8 : 0);
                const unsigned cu = q0 ^ q2, cv = q1 ^ q3;
                const unsigned pu = __shfl_xor(odd ? cu : cv, 1, 64);
                v2u sc = odd ? v2u{pu, cv} : v2u{cu, pu};
                unsigned char *dc = odd ? (d2 - 4) : d1;
                if (NT) {
                    __builtin_nontemporal_store(s, reinterpret_cast<v4u *>(dy));
                    __builtin_nontemporal_store(sc, reinterpret_cast<v2u *>(dc));
                } else {
                    *reinterpret_cast<v4u *>(dy) = s; *reinterpret_cast<v2u *>(dc) = sc;
                }
            }
        } else {
            acc ^= q0 ^ q1 ^ q2 ^ q3;
        }
    }
    if (!(MODE & 2) && acc == 0x12345678u) a.y[threadIdx.x] = 1;
}

template <int MODE, int SP, bool NT>
static void run(const char *name, const Args &a, int threads, int grid, double bytes)
{
    std::vector<float> ms;
    for (int rep = 0; rep < 10; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<MODE, SP, NT><<<grid, threads>>>(a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep >= 2) ms.push_back(t);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-44s block %4d grid %5d : med %.4f min %.4f ms -> %.0f GB/s\n", name, threads, grid, ms[ms.size() / 2], ms[0],
           bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
}

int main()
{
    const int W = 3840, H = 2160, B = 20;
    const size_t cs = (size_t)W * H;
    Args a{};
    float *src; hipMalloc(&src, cs * 3 * B * 4); hipMemset(src, 1, cs * 3 * B * 4);
    hipMalloc(&a.y, cs * 2 * B); hipMalloc(&a.u, cs / 2 * B); hipMalloc(&a.v, cs / 2 * B);
    a.src = src; a.w = W; a.h = H; a.nframes = B;
    const double px = (double)cs * B;
    for (int threads : {256, 1024}) {
        const int NW = threads / 64;
        a.tilesX = (W / 4 + 63) / 64; a.tilesY = (H / 2 + NW - 1) / NW; a.tpf = a.tilesX * a.tilesY; a.total = a.tpf * B;
        for (int grid : {256 * 2048 / threads, 256 * 1024 / threads}) {
            run<1, 0, true>("R  loads only (12 B/px), nt", a, threads, grid, 12 * px);
            run<2, 0, true>("W  stores only (3 B/px), nt, kernel pattern", a, threads, grid, 3 * px);
            run<2, 1, true>("W  stores only, nt, 16B-Y/8B-C pattern", a, threads, grid, 3 * px);
            run<2, 0, false>("W  stores only, plain, kernel pattern", a, threads, grid, 3 * px);
            run<3, 0, true>("RW both (15 B/px), nt, kernel pattern", a, threads, grid, 15 * px);
            run<3, 1, true>("RW both, nt, 16B-Y/8B-C pattern", a, threads, grid, 15 * px);
            run<3, 0, false>("RW both, plain", a, threads, grid, 15 * px);
        }
    }
    return 0;
}
