// lumahip_encode.hip -- dispatch of the fused encode kernels (lh::k_encode, luma_kernels.hpp) and of the other encode-side
// instantiations: array quantize, the search probe, the traffic-only probe.
#include "lumahip_internal.hpp"

using namespace lh;
using namespace lhost;

namespace lh {
// Traffic probe: the loads and stores of k_encode<.,4:2:0,VW=4> for 16-bit planes with NO arithmetic (an xor
// keeps every loaded word live).  Its run time is what the memory system alone needs for the encode traffic mix
// (12 B read + 3 B written per pixel, same tile order, same non-temporal accesses); bench.py reports the encode
// kernel's time as a fraction of it next to the fraction of the 8 TB/s peak.
__global__ __launch_bounds__(256) void k_encode_traffic_probe(const EncArgs a)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    for (int t = blockIdx.x; t < a.g.totalTiles; t += gridDim.x) {
        int f, bx, by;
        tile_coords(t, a.g, f, bx, by);
        const int ux = bx * 64 + tx, uy = by * NW + ty;
        if (ux >= a.g.unitsX || uy >= a.g.unitsY)
            continue;
        const size_t off = (size_t)f * a.frame_stride + (size_t)(2 * uy) * a.g.w + (size_t)ux * 4;
        uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                float v[4];
                load_px<4>(a.src[c] + off + (size_t)r * a.g.w, v);
                q0 ^= __float_as_uint(v[0]); q1 ^= __float_as_uint(v[1]); q2 ^= __float_as_uint(v[2]); q3 ^= __float_as_uint(v[3]);
            }
        unsigned char *d0 = a.dst[0] + (size_t)f * a.dst_frame_stride[0] + (size_t)(2 * uy) * a.stride[0] + (size_t)ux * 8;
        nt_store_u32x2(d0, q0, q1);
        nt_store_u32x2(d0 + a.stride[0], q2, q3);
        nt_store_u32(a.dst[1] + (size_t)f * a.dst_frame_stride[1] + (size_t)uy * a.stride[1] + (size_t)ux * 4, q0 ^ q2);
        nt_store_u32(a.dst[2] + (size_t)f * a.dst_frame_stride[2] + (size_t)uy * a.stride[2] + (size_t)ux * 4, q1 ^ q3);
    }
}

}  // namespace lh

typedef void (*enc_kernel_t)(const EncArgs);

template <int CS, bool SUB>
static enc_kernel_t pick_enc2(int vw, int mode)
{
    if constexpr (CS == CS_YCBCR) {
        if (mode == 5)   // composite luma -> code records (vw == 4 or 2 as for the records)
            return vw == 4 ? k_encode<CS, SUB, 4, 5> : k_encode<CS, SUB, 2, 5>;
        if (mode == 6)   // the same + the half-input table
            return vw == 4 ? k_encode<CS, SUB, 4, 6> : k_encode<CS, SUB, 2, 6>;
    }
    if (mode == LUT_THRESH_LDS)
        return vw == 4 ? k_encode<CS, SUB, 4, 3> : k_encode<CS, SUB, 2, 3>;
    if (mode == LUT_THRESH_GLOBAL)
        return vw == 4 ? k_encode<CS, SUB, 4, 4> : k_encode<CS, SUB, 2, 4>;
    if (mode == LUT_LINKEY_LDS)
        return vw == 4 ? k_encode<CS, SUB, 4, 7> : k_encode<CS, SUB, 2, 7>;
    if (mode == LUT_LITERAL_LDS)
        return k_encode<CS, SUB, 2, 0>;
    return k_encode<CS, SUB, 2, 2>;
}

// binary16 frames: the records-in-LDS kernels at four pixels per thread only (what the host entry points' half upload needs)
static enc_kernel_t pick_enc_in16(int cs, bool sub)
{
    switch (cs) {
    case CS_LUV: return sub ? k_encode<CS_LUV, true, 4, 3, true> : k_encode<CS_LUV, false, 4, 3, true>;
    case CS_RGB: return sub ? k_encode<CS_RGB, true, 4, 3, true> : k_encode<CS_RGB, false, 4, 3, true>;
    case CS_YCBCR: return sub ? k_encode<CS_YCBCR, true, 4, 3, true> : k_encode<CS_YCBCR, false, 4, 3, true>;
    case CS_XYZ: return sub ? k_encode<CS_XYZ, true, 4, 3, true> : k_encode<CS_XYZ, false, 4, 3, true>;
    }
    return nullptr;   // (CS_PACK: frames that are already colour-transformed do not hold halves; never asked for)
}

static enc_kernel_t pick_enc(int cs, bool sub, int vw, int mode)
{
    switch (cs) {
    case CS_LUV: return sub ? pick_enc2<CS_LUV, true>(vw, mode) : pick_enc2<CS_LUV, false>(vw, mode);
    case CS_RGB: return sub ? pick_enc2<CS_RGB, true>(vw, mode) : pick_enc2<CS_RGB, false>(vw, mode);
    case CS_YCBCR: return sub ? pick_enc2<CS_YCBCR, true>(vw, mode) : pick_enc2<CS_YCBCR, false>(vw, mode);
    case CS_XYZ: return sub ? pick_enc2<CS_XYZ, true>(vw, mode) : pick_enc2<CS_XYZ, false>(vw, mode);
    case CS_PACK: return sub ? pick_enc2<CS_PACK, true>(vw, mode) : pick_enc2<CS_PACK, false>(vw, mode);
    }
    return nullptr;
}

// partial triples {0, +inf, -inf}: nframes * STATS_SLOTS of them
__global__ void k_init_stats(float *s, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        s[3 * i + 0] = 0.0f;
        s[3 * i + 1] = __builtin_inff();
        s[3 * i + 2] = -__builtin_inff();
    }
}

// fold the STATS_SLOTS partial triples of every frame into the caller's {sum, min, max}
__global__ void k_fold_stats(const float *part, float *out, int nframes)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < nframes) {
        float s = 0.0f, mn = __builtin_inff(), mx = -__builtin_inff();
        for (int k = 0; k < STATS_SLOTS; k++) {
            const float *p = part + 3 * ((size_t)f * STATS_SLOTS + k);
            s += p[0];
            mn = fminf(mn, p[1]);
            mx = fmaxf(mx, p[2]);
        }
        out[3 * f + 0] = s;
        out[3 * f + 1] = mn;
        out[3 * f + 2] = mx;
    }
}

namespace lhost {

int encode_frames_device_impl(lumahip_ctx *c, const float *const rgb[3], size_t frame_stride, unsigned nframes,
                              unsigned w, unsigned h, float sc, int profile, unsigned char *const planes[3],
                              const int stride[3], const size_t pfs[3], float *stats, int cs_eff, bool lanes, bool in16)
{
    if (!c || !rgb || !rgb[0] || !rgb[1] || !rgb[2] || !planes || !stride || !pfs || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    // (binary16 frames come from this library's own staging code: the overlap test of the float planes, which measures in floats, is skipped)
    if ((rc = check_layout(c, w, h, profile, nframes, in16 ? nullptr : rgb, frame_stride, stride, pfs)))
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure_search_index(c)))
        return rc;
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    const int mode = c->q.mode;
    const bool fast_search = (mode == LUT_THRESH_LDS || mode == LUT_THRESH_GLOBAL || mode == LUT_LINKEY_LDS);
    const bool al16 = is_aligned(rgb[0], 16) && is_aligned(rgb[1], 16) && is_aligned(rgb[2], 16);
    int vw = (fast_search && (w % 4) == 0 && al16 && (frame_stride % 4) == 0) ? 4 : 2;
    if (!is_aligned(rgb[0], 8) || !is_aligned(rgb[1], 8) || !is_aligned(rgb[2], 8) || (frame_stride % 2) != 0)
        return fail(c, LUMAHIP_ERR_ARG, "colour planes must be 8-byte aligned and the frame stride even");
    if (in16) {   // halves: 8-byte loads of four pixels
        if (mode != LUT_THRESH_LDS || (w % 4) != 0 || (frame_stride % 4) != 0)
            return fail(c, LUMAHIP_ERR_UNSUPPORTED, "binary16 frames need the luminance records in LDS and rows of a multiple of 4 pixels");
        vw = 4;
    }
    // YCbCr without per-frame statistics (they need the luminance itself): the luminance code comes straight from the luma
    const bool ycode = !in16 && cs_eff == CS_YCBCR && !stats && ycbcr_composite_ready(c);
    // ... and R', G', B' of binary16 inputs from the half-input table of this call's (sc, Lmax), when table + records fit the LDS
    const float *half = nullptr;
    uint32_t *half_flag = nullptr;   // this launch's feedback word (LagPolicy)
    if (ycode && c->half_mode != 0 && lds_bytes(c, true, cs_eff, true, true) <= LUMAHIP_LDS_PER_WORKGROUP) {
        if ((rc = half_table_for(c, sc, &half)))
            return rc;
        if (half && c->half_mode == 1 && !lag_policy_next(c->half_pol, &half_flag))   // (mode 2: always, no feedback)
            half = nullptr;
    }
    LagLaunchGuard half_guard{c->half_pol, half_flag};   // (a return before the launch takes the word back)
    EncArgs a{};
    a.q = ycode ? c->q_y : c->q;
    a.half = half;
    if (half) {
        c->half_launches++;
        a.half_flag = half_flag;
    }
    const size_t lds = lds_bytes(c, true, cs_eff, ycode, half != nullptr);
    const bool long_launch = (unsigned long long)w * h * nframes >= 60000000ull;   // >= 7 4K frames
    const int threads = block_threads_for(c, lds, long_launch && cs_eff != CS_YCBCR, cs_eff == CS_YCBCR && !half);
    if (!make_geom(a.g, w, h, vw, threads / 64, nframes))
        return fail(c, LUMAHIP_ERR_ARG, "batch too large: more than 2^31 tiles in one launch");
    for (int k = 0; k < 3; k++)
        a.src[k] = rgb[k];
    a.frame_stride = frame_stride;
    a.sc = sc;
    a.bps = bps;
    a.stats = nullptr;
    a.aligned = 1;
    for (int p = 0; p < 3; p++) {
        if (!planes[p])
            return fail(c, LUMAHIP_ERR_ARG, "null plane %d", p);
        a.dst[p] = planes[p];
        a.stride[p] = stride[p];
        a.dst_frame_stride[p] = pfs[p];
        const size_t ub = (size_t)((p && sub) ? vw / 2 : vw) * bps;
        if (!is_aligned(planes[p], ub) || (stride[p] % (int)ub) != 0 || (pfs[p] % ub) != 0)
            a.aligned = 0;
    }
    a.q.cs = cs_eff;
    enc_kernel_t kern = in16 ? pick_enc_in16(cs_eff, sub) : pick_enc(cs_eff, sub, vw, half ? 6 : ycode ? 5 : mode);
    if (!kern)
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "no encode kernel for colour space %d%s", cs_eff, in16 ? " with binary16 frames" : "");
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = grid_for(c, threads, a.g.totalTiles, 0, 0, half ? 2 : cs_eff == CS_YCBCR ? 1 : 0);
    hipStream_t s = launch_stream(c, lanes);
    if (stats) {
        // partial triples live in a context-owned scratch buffer; launches with statistics of one context share it, which is
        // safe on one stream (in order) and is why an unordered section with statistics keeps to its first lane
        const size_t need = (size_t)nframes * STATS_SLOTS * 3 * sizeof(float);
        if (c->d_stats_part_cap < need) {
            HIPCHK(c, hipDeviceSynchronize());
            (void)hipFree(c->d_stats_part);
            c->d_stats_part = nullptr;
            c->d_stats_part_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_stats_part, need));
            c->d_stats_part_cap = need;
        }
        if (lanes && c->lanes_active)
            s = c->lane_stream[0];
        a.stats = c->d_stats_part;
        const int np = (int)nframes * STATS_SLOTS;
        hipLaunchKernelGGL(k_init_stats, dim3((np + 255) / 256), dim3(256), 0, s, c->d_stats_part, np);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, a);
    half_guard.launched = true;
    if (half_flag && (rc = lag_policy_launched(c, c->half_pol, s)))
        return rc;
    if (stats)
        hipLaunchKernelGGL(k_fold_stats, dim3((nframes + 63) / 64), dim3(64), 0, s, c->d_stats_part, stats, (int)nframes);
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}

bool encode_supports_in16(lumahip_ctx *c, unsigned w)
{
    return ensure_search_index(c) == LUMAHIP_OK && c->q.mode == LUT_THRESH_LDS && (w % 4) == 0;
}

}  // namespace lhost

extern "C" int lumahip_encode_frames_device(lumahip_ctx *c, const float *rgb, size_t frame_stride, unsigned nframes,
                                            unsigned w, unsigned h, float sc, int profile,
                                            unsigned char *const planes[3], const int stride[3],
                                            const size_t pfs[3], float *stats)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!rgb)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    const size_t n = (size_t)w * h;
    const float *const pl[3] = {rgb, rgb + n, rgb + 2 * n};
    return encode_frames_device_impl(c, pl, frame_stride, nframes, w, h, sc, profile, planes, stride, pfs, stats, c->q.cs, true);
}

extern "C" int lumahip_encode_frames_device_planar(lumahip_ctx *c, const float *const rgb_planes[3], size_t frame_stride,
                                                   unsigned nframes, unsigned w, unsigned h, float sc, int profile,
                                                   unsigned char *const planes[3], const int stride[3], const size_t pfs[3],
                                                   float *stats)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    return encode_frames_device_impl(c, rgb_planes, frame_stride, nframes, w, h, sc, profile, planes, stride, pfs, stats, c->q.cs, true);
}

extern "C" int lumahip_probe_encode_traffic_device(lumahip_ctx *c, const float *rgb, size_t frame_stride, unsigned nframes,
                                                   unsigned w, unsigned h, unsigned char *const planes[3],
                                                   const int stride[3], const size_t pfs[3], int iters, float *avg_ms)
{
    if (!c || !rgb || !planes || !stride || !pfs || nframes == 0 || iters <= 0 || !avg_ms)
        return fail(c, LUMAHIP_ERR_ARG, "bad argument");
    if (w == 0 || h == 0 || (w % 4) || (h & 1) || !is_aligned(rgb, 16) || (frame_stride % 4))
        return fail(c, LUMAHIP_ERR_ARG, "the traffic probe needs w %% 4 == 0, even h and 16-byte aligned frames");
    for (int p = 0; p < 3; p++)
        if (!planes[p] || !is_aligned(planes[p], 8) || (stride[p] % (p ? 4 : 8)) || (pfs[p] % 8))
            return fail(c, LUMAHIP_ERR_ARG, "the traffic probe needs 8-byte aligned 16-bit 4:2:0 planes");
    HIPCHK(c, hipSetDevice(c->device));
    EncArgs a{};
    const int threads = 256;
    if (!make_geom(a.g, w, h, 4, threads / 64, nframes))
        return fail(c, LUMAHIP_ERR_ARG, "batch too large");
    for (int k = 0; k < 3; k++)
        a.src[k] = rgb + (size_t)k * w * h;
    a.frame_stride = frame_stride;
    a.bps = 2;
    a.aligned = 1;
    for (int p = 0; p < 3; p++) {
        a.dst[p] = planes[p];
        a.stride[p] = stride[p];
        a.dst_frame_stride[p] = pfs[p];
    }
    const int grid = grid_for(c, threads, a.g.totalTiles, 0);
    EventPair ev;
    HIPCHK(c, ev.create());
    HIPCHK(c, hipEventRecord(ev.e0, c->stream));
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(k_encode_traffic_probe, dim3(grid), dim3(threads), 0, c->stream, a);
    HIPCHK(c, hipEventRecord(ev.e1, c->stream));
    HIPCHK(c, hipEventSynchronize(ev.e1));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    HIPCHK(c, hipGetLastError());
    *avg_ms = ms / iters;
    return LUMAHIP_OK;
}

extern "C" int lumahip_quantize_probe_device(lumahip_ctx *c, uint16_t *out_dev, uint32_t first_bits, size_t n, int nonneg)
{
    if (!c || !out_dev || n == 0 || (n % 4) != 0 || !is_aligned(out_dev, 8))
        return fail(c, LUMAHIP_ERR_ARG, "bad argument (n must be a multiple of 4, out 8-byte aligned)");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_search_index(c))
        return rc;
    const size_t lds = lds_bytes(c, true, CS_PACK);
    void (*kern)(const QuantDev, uint16_t *, uint32_t, size_t) = nullptr;
    switch (c->q.mode) {
    case LUT_LITERAL_LDS: kern = nonneg ? k_quantize_probe<0, true> : k_quantize_probe<0, false>; break;
    case LUT_LITERAL_GLOBAL: kern = nonneg ? k_quantize_probe<2, true> : k_quantize_probe<2, false>; break;
    case LUT_THRESH_LDS: kern = nonneg ? k_quantize_probe<3, true> : k_quantize_probe<3, false>; break;
    case LUT_THRESH_GLOBAL: kern = nonneg ? k_quantize_probe<4, true> : k_quantize_probe<4, false>; break;
    case LUT_LINKEY_LDS: kern = nonneg ? k_quantize_probe<7, true> : k_quantize_probe<7, false>; break;
    }
    if (!kern)
        return fail(c, LUMAHIP_ERR_STATE, "unknown search mode %d", c->q.mode);
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long grid = (long)((n / 4 + 255) / 256);
    if (grid > (long)c->num_cu * 8)
        grid = (long)c->num_cu * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, c->stream, c->q, out_dev, first_bits, n / 4);
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}


// ---- test probe: the luminance code of a YCbCr pixel as a function of t = 219 y + 16 (y = its luma), over consecutive fp32 bit patterns ----
// direct = 0: through the composite records exactly as k_encode<CS_YCBCR, ., ., 5> reads them (quantize_thresh<4, NONNEG>);
// direct = 1: the reference's arithmetic on the device -- PQdec((219 y + 16) / 255) with the complete powf and IEEE division,
// then the literal table search.  tests/test_gpu_exhaustive.py compares the two for every t >= +0 and every NaN.
namespace lh {
__global__ __launch_bounds__(256) void k_ycbcr_luma_probe(const QuantDev q, const QuantDev qy, float Lmax, uint16_t *out, uint32_t first_bits,
                                                          size_t n4, int direct)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PowfTablesWide *pw = reinterpret_cast<PowfTablesWide *>(smem);
    stage_powf_tables(pw);
    uint32_t *s_rec = reinterpret_cast<uint32_t *>(smem + sizeof(PowfTablesWide));
    for (int i = threadIdx.x; i < qy.nbuckets; i += blockDim.x)
        s_rec[i] = qy.rec[i];
    __syncthreads();
    const XformConst k = make_xform_const<CS_YCBCR>(1.0f, Lmax, pw);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float v[4];
        int c[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            v[j] = __uint_as_float(first_bits + (uint32_t)(4 * i + j));
        if (direct) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                c[j] = quantize_lut_literal(pq_decode(div_ieee(v[j], 255.0f), k), q.lut, q.maxVal);
        } else {
            quantize_lut<5, 4>(v, c, q.lut, s_rec, qy);
        }
        store_samples<4>(reinterpret_cast<unsigned char *>(out + 4 * i), c, 2, 1);
    }
}
}  // namespace lh

extern "C" int lumahip_ycbcr_luma_probe_device(lumahip_ctx *c, uint16_t *out_dev, uint32_t first_bits, size_t n, int direct)
{
    if (!c || !out_dev || n == 0 || (n % 4) != 0 || !is_aligned(out_dev, 8))
        return fail(c, LUMAHIP_ERR_ARG, "bad argument (n must be a multiple of 4, out 8-byte aligned)");
    if (!c->have_quant || c->q.cs != CS_YCBCR)
        return fail(c, LUMAHIP_ERR_STATE, "the probe needs a YCbCr quantizer");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_search_index(c))
        return rc;
    if (!ycbcr_composite_ready(c))
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "no composite luma -> code records for this table");
    const size_t lds = sizeof(PowfTablesWide) + (((size_t)c->q_y.nbuckets * 4 + 15) & ~(size_t)15);
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void *)k_ycbcr_luma_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long grid = (long)((n / 4 + 255) / 256);
    if (grid > (long)c->num_cu * 8)
        grid = (long)c->num_cu * 8;
    hipLaunchKernelGGL(k_ycbcr_luma_probe, dim3((unsigned)grid), dim3(256), lds, c->stream, c->q, c->q_y, c->q.Lmax, out_dev, first_bits,
                       n / 4, direct);
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}
