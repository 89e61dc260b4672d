// exp_encode.hip -- experiment harness: ONE instantiation of the headline kernel (k_encode<CS_LUV,4:2:0,VW=4,records>)
// built in seconds, timed L2-fed (512x512 frames aliasing one frame), Infinity-Cache-fed (4K frames aliasing one
// frame) and HBM-fed (distinct 4K frames), same pixel count each.  Kernel variants are selected with -D switches at
// compile time (see the LH_EXP_* uses in luma_device.hpp / luma_kernels.hpp while an experiment is in flight).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Ilumahdrv_amd/csrc \
//         tools/experiments/exp_encode.hip lumahdrv_amd/csrc/lut_index.cpp lumahdrv_amd/csrc/host_lut.cpp -ldl -o exp_encode
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/lumahip.h"
#include "luma_kernels.hpp"
#include "lut_index.hpp"

using namespace lh;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

#ifndef EXP_LM
#define EXP_LM 3
#endif
#ifndef EXP_CS
#define EXP_CS 0  // 0 = CS_LUV (PQ-11, the headline), 2 = CS_YCBCR (the HDR10 recipe: PQ-10, 1000 cd/m2, preScaling 20)
#endif

static bool geom(FrameGeom &g, int w, int h, int vw, int nw, int nframes)
{
    g.w = w; g.h = h; g.unitsX = w / vw; g.unitsY = h / 2;
    g.tilesX = (g.unitsX + 63) / 64; g.tilesY = (g.unitsY + nw - 1) / nw;
    g.tilesPerFrame = g.tilesX * g.tilesY; g.totalTiles = g.tilesPerFrame * nframes;
    return true;
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 256;
    const int percu = argc > 2 ? atoi(argv[2]) : 2048 / threads;
    const int ptf = argc > 3 ? atoi(argv[3]) : 1, bits = argc > 4 ? atoi(argv[4]) : (EXP_CS == 2 ? 10 : 11);
    const float maxLum = EXP_CS == 2 ? 1000.0f : 1e4f;
    const size_t n = (size_t)1 << bits;
    std::vector<float> lut(n);
    if (lumahip_build_lut(ptf, bits, maxLum, 0.005f, lut.data(), n)) return 2;
    ThreshIndex ix = build_thresh_index(lut.data(), (int)n, 1 << 19);
    if (!ix.ok) return 3;
    QuantDev q{};
    std::vector<float> padded((n + 4) & ~(size_t)3, __builtin_nanf(""));
    memcpy(padded.data(), lut.data(), n * 4);
    float *d_lut; uint32_t *d_rec;
    CK(hipMalloc(&d_lut, padded.size() * 4));
    CK(hipMemcpy(d_lut, padded.data(), padded.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> rec((ix.rec.size() + 3) & ~(size_t)3, 0u);
    memcpy(rec.data(), ix.rec.data(), ix.rec.size() * 4);
    CK(hipMalloc(&d_rec, rec.size() * 4));
    CK(hipMemcpy(d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    q.lut = d_lut; q.rec = d_rec; q.lut_len = (int)n; q.pad = (int)(padded.size() - n);
    q.maxVal = (int)n - 1; q.mode = LUT_THRESH_LDS; q.shift = ix.shift; q.kmin = ix.kmin; q.nbuckets = ix.nbuckets;
    q.maxC = EXP_CS == 2 ? 1023.0f : 255.0f; q.cs = EXP_CS; q.Lmax = maxLum;
    const size_t lds = (EXP_LM == 3 ? (((size_t)ix.nbuckets * 4 + 15) & ~(size_t)15) : 0) + (EXP_CS == 2 ? sizeof(PowfTablesWide) : 0);

    const int W = 3840, H = 2160, B = 20;
    const size_t n3 = (size_t)3 * W * H;
    float *src;
    CK(hipMalloc(&src, (size_t)B * n3 * 4 * 2));  // two distinct batches
    k_synth<<<4096, 256>>>(src, n3, 2 * B, n3, 20250929ull, 0ull);
    const int st[3] = {W * 2, W, W};
    const size_t psz[3] = {(size_t)H * st[0], (size_t)(H / 2) * st[1], (size_t)(H / 2) * st[2]};
    unsigned char *pl[3];
    for (int p = 0; p < 3; p++) CK(hipMalloc(&pl[p], psz[p] * B * 2));
    CK(hipDeviceSynchronize());
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;

    auto kern = k_encode<EXP_CS, true, 4, EXP_LM>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    struct Case { const char *name; int w, h, nf; size_t fs; bool alias; };
    const Case cases[] = {{"L2-fed  512x512 ", 512, 512, B * W * H / (512 * 512), 0, true},
                          {"MALL-fed 4K alias", W, H, B, 0, true},
                          {"HBM-fed 4K       ", W, H, B, n3, false}};
    for (const Case &cs : cases) {
        EncArgs a{};
        a.q = q;
        geom(a.g, cs.w, cs.h, 4, threads / 64, cs.nf);
        a.frame_stride = cs.fs; a.sc = EXP_CS == 2 ? 20.0f : 1.0f; a.bps = 2; a.aligned = 1; a.stats = nullptr;
        const int cst[3] = {cs.w * 2, cs.w, cs.w};
        long grid = std::min<long>((long)ncu * percu, a.g.totalTiles);
        std::vector<float> ms;
        for (int rep = 0; rep < 12; rep++) {
            const int b = rep & 1;
            a.src = src + (cs.alias ? 0 : (size_t)b * B * n3);
            for (int p = 0; p < 3; p++) {
                a.dst[p] = pl[p] + (cs.alias ? 0 : (size_t)b * B * psz[p]);
                a.stride[p] = cst[p];
                a.dst_frame_stride[p] = cs.alias ? 0 : psz[p];
            }
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
#ifdef EXP_PROBE
            hipLaunchKernelGGL(k_encode_traffic_probe, dim3((unsigned)grid), dim3(threads), 0, 0, a);
#else
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, 0, a);
#endif
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep >= 2) ms.push_back(t);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        std::sort(ms.begin(), ms.end());
        printf("%s block %d wg/CU %d lds %zu : med %.4f min %.4f ms\n", cs.name, threads, percu, lds, ms[ms.size() / 2], ms[0]);
    }
    return 0;
}
