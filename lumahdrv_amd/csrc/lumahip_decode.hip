// lumahip_decode.hip -- dispatch of the fused decode kernels (lh::k_decode, luma_kernels.hpp), with and without the display
// epilogue, and the array forms of quantize / dequantize.
#include "lumahip_internal.hpp"

using namespace lh;
using namespace lhost;

namespace lh {
// ---- array quantize / dequantize (LumaQuantizer::quantize / dequantize over arrays) ----------------
struct QArrArgs {
    QuantDev q;
    const float *in;
    float *out;
    size_t n;
    int lut_channel;  // 1: LUT path, 0: colour path
};

// MODE: the table's search mode (lut_index.hpp LutMode), a template parameter so that each instantiation stages
// exactly what it probes
template <int MODE>
__global__ __launch_bounds__(256) void k_quantize_array(const QArrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    stage_tables<(MODE == 0 ? STAGE_LUT : 0) | ((MODE == 3 || MODE == 7) ? STAGE_REC : 0)>(smem, a.q);
    const float *s_lut = reinterpret_cast<const float *>(smem);
    const uint32_t *s_rec = reinterpret_cast<const uint32_t *>(smem);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        const float v[1] = {a.in[i]};
        int c[1];
        if (!a.lut_channel)
            c[0] = quantize_color(v[0], a.q.maxC);
        else if constexpr (MODE == 3 || MODE == 7)
            quantize_lut<MODE, 1>(v, c, s_lut, s_rec, a.q);  // any NaN sign
        else if constexpr (MODE == 4)
            quantize_lut<4, 1>(v, c, a.q.lut, a.q.rec, a.q);
        else if constexpr (MODE == 0)
            quantize_lut<0, 1>(v, c, s_lut, s_rec, a.q);
        else
            quantize_lut<2, 1>(v, c, a.q.lut, s_rec, a.q);
        a.out[i] = (float)c[0];
    }
}

__global__ __launch_bounds__(256) void k_dequantize_array(const QArrArgs a)
{
    // src/luma_quantizer.cpp:247-264 with a float argument (may be negative, fractional or NaN)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        const float val = a.in[i];
        float res;
        if (a.lut_channel) {
            if (val < 0)
                res = a.q.lut[0];
            else if (val >= (float)a.q.maxVal)
                res = a.q.lut[a.q.maxVal];
            else
                res = a.q.lut[(val != val) ? a.q.maxVal : (int)val];
        } else {
            res = std_max(div_ieee(val, a.q.maxC), 1e-10f);
        }
        a.out[i] = res;
    }
}

// The per-stream red / blue tables of the YCbCr decode kernels (DecArgs::rb): out[cb * n + y] = blue of (luminance code y,
// colour code cb), out[plane + cr * n + y] = red, with the complete functions on the device powf (== the host libm's, pow_glibc.hpp)
struct RbArgs {
    const float *ytab;   // n entries (+ padding)
    float *out;
    int n, nc;           // luminance codes, colour codes
    float maxC, sc, Lmax;
};

__global__ __launch_bounds__(256) void k_build_rb(const RbArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[sizeof(PowfTablesWide)];
    PowfTablesWide &s_pw = *reinterpret_cast<PowfTablesWide *>(s_raw);
    stage_powf_tables(&s_pw);
    __syncthreads();
    const XformConst k = make_xform_const<CS_YCBCR>(a.sc, a.Lmax, &s_pw);
    const size_t plane = (size_t)a.n * a.nc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * plane; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i < plane ? i : i - plane;
        const int cc = (int)(j / a.n), yc = (int)(j - (size_t)cc * a.n);
        a.out[i] = ycbcr_rb_entry(a.ytab[yc], cc, a.maxC, i < plane ? 1.8814f : 1.4746f, k);
    }
}

}  // namespace lh

typedef void (*dec_kernel_t)(const DecArgs);

template <int CS, bool SUB>
static dec_kernel_t pick_dec2(int vw, bool gl, bool disp, bool yt, bool rb)
{
    if constexpr (CS == CS_YCBCR) {
        if (yt && rb && !gl && !disp)   // + red and blue from the per-stream tables in global memory
            return vw == 4 ? k_decode<CS, SUB, 4, false, false, true, true> : k_decode<CS, SUB, 2, false, false, true, true>;
        if (yt && !gl && !disp)   // per-stream y table in LDS
            return vw == 4 ? k_decode<CS, SUB, 4, false, false, true> : k_decode<CS, SUB, 2, false, false, true>;
    }
    if (disp) {
        if (gl)
            return k_decode<CS, SUB, 2, true, true>;
        return vw == 4 ? k_decode<CS, SUB, 4, false, true> : k_decode<CS, SUB, 2, false, true>;
    }
    if (gl)
        return k_decode<CS, SUB, 2, true>;
    return vw == 4 ? k_decode<CS, SUB, 4, false> : k_decode<CS, SUB, 2, false>;
}

static dec_kernel_t pick_dec(int cs, bool sub, int vw, bool gl, bool disp, bool yt, bool rb)
{
    switch (cs) {
    case CS_LUV: return sub ? pick_dec2<CS_LUV, true>(vw, gl, disp, yt, rb) : pick_dec2<CS_LUV, false>(vw, gl, disp, yt, rb);
    case CS_RGB: return sub ? pick_dec2<CS_RGB, true>(vw, gl, disp, yt, rb) : pick_dec2<CS_RGB, false>(vw, gl, disp, yt, rb);
    case CS_YCBCR: return sub ? pick_dec2<CS_YCBCR, true>(vw, gl, disp, yt, rb) : pick_dec2<CS_YCBCR, false>(vw, gl, disp, yt, rb);
    case CS_XYZ: return sub ? pick_dec2<CS_XYZ, true>(vw, gl, disp, yt, rb) : pick_dec2<CS_XYZ, false>(vw, gl, disp, yt, rb);
    case CS_PACK: return sub ? pick_dec2<CS_PACK, true>(vw, gl, disp, yt, rb) : pick_dec2<CS_PACK, false>(vw, gl, disp, yt, rb);
    }
    return nullptr;
}

namespace lhost {

// true when no colour plane of the batch starts inside another one's extent: R, G and B are three buffers, not three
// sections of packed LumaFrames
static bool planes_are_separate_buffers(float *const rgb[3], size_t frame_stride, unsigned nframes, unsigned w, unsigned h)
{
    const size_t extent = ((size_t)(nframes - 1) * frame_stride + (size_t)w * h) * sizeof(float);
    auto apart = [&](const float *p, const float *q) {
        const uintptr_t x = (uintptr_t)p, y = (uintptr_t)q;
        return (x > y ? x - y : y - x) >= extent;
    };
    return apart(rgb[0], rgb[1]) && apart(rgb[1], rgb[2]) && apart(rgb[0], rgb[2]);
}

// The device red / blue tables of this call's preScaling (nullptr: not for this stream).  Built by one launch of k_build_rb the
// first time a preScaling is seen (two tables of 2^(bitdepth + bitdepthC) floats: 8 MiB for the HDR10 recipe, ~20 us), kept per
// context -- up to two preScalings; launches that read an older copy may still be queued anywhere, so making room waits for the
// device first.  What a gather costs is the L1 hit rate of the lines a picture touches, not the size of the table: PQ-12 with
// 12-bit colour (2 x 64 MiB) is read as profitably as the HDR10 recipe's 8 MiB; beyond RB_MAX_BYTES the tables are not built.
static constexpr size_t RB_MAX_BYTES = (size_t)256 << 20;

int rb_table_for(lumahip_ctx *c, float sc, const float **tab)
{
    *tab = nullptr;
    const size_t n = (size_t)c->q.lut_len, nc = (size_t)c->q.maxC + 1;
    // The tables are a speed-up, never a requirement: when they cannot be had (no memory for them on a nearly full GPU, or the
    // build launch fails) the call goes on with the plain kernels, and the stream does not ask again (lumahip_set_quantizer
    // clears `rb_unavailable` with the tables).
    if (c->rb_mode == 0 || c->rb_unavailable || !c->q.ytab || 2 * n * nc * sizeof(float) > RB_MAX_BYTES || !(sc == sc))
        return LUMAHIP_OK;
    for (auto &t : c->rb_tabs)
        if (memcmp(&t.sc, &sc, 4) == 0) {
            t.last_use = ++c->rb_clock;
            *tab = t.d;
            return LUMAHIP_OK;
        }
    if (c->rb_tabs.size() >= 2) {
        HIPCHK(c, hipDeviceSynchronize());
        const size_t old = c->rb_tabs[0].last_use < c->rb_tabs[1].last_use ? 0 : 1;
        (void)hipFree(c->rb_tabs[old].d);
        c->rb_tabs.erase(c->rb_tabs.begin() + (long)old);
    }
    lumahip_ctx::RbTab t;
    t.sc = sc;
    t.last_use = ++c->rb_clock;
    if (c->test_fail_rb_alloc || hipMalloc(&t.d, 2 * n * nc * sizeof(float)) != hipSuccess) {
        c->test_fail_rb_alloc = false;
        (void)hipGetLastError();
        c->rb_unavailable = true;
        return LUMAHIP_OK;
    }
    RbArgs a{};
    a.ytab = c->q.ytab;
    a.out = t.d;
    a.n = (int)n;
    a.nc = (int)nc;
    a.maxC = c->q.maxC;
    a.sc = sc;
    a.Lmax = c->q.Lmax;
    hipLaunchKernelGGL(k_build_rb, dim3((unsigned)c->num_cu * 8), dim3(256), 0, c->stream, a);
    // done when this returns, whatever stream or lane the decode launch goes to
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(t.d);
        c->rb_unavailable = true;
        return LUMAHIP_OK;
    }
    c->rb_tabs.push_back(t);
    *tab = t.d;
    return LUMAHIP_OK;
}

int decode_impl(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3], const size_t pfs[3],
                unsigned nframes, unsigned w, unsigned h, int profile, float sc, float *const rgb_in[3], size_t frame_stride,
                const DisplayParams &dp, int cs_eff, bool lanes, float *const rot[3])
{
    // rot: packed frames rotating over three buffers (DecArgs::rot); the checks below then look at buffer 0's first frame
    float *rot_planes[3] = {nullptr, nullptr, nullptr};
    if (rot) {
        if (!rot[0] || !rot[1] || !rot[2] || dp.rgba)
            return fail(c, LUMAHIP_ERR_ARG, "null argument");
        if (frame_stride < (size_t)3 * w * h)
            return fail(c, LUMAHIP_ERR_ARG, "frame stride %zu < 3*w*h = %zu floats (packed frames)", frame_stride, (size_t)3 * w * h);
        // every buffer is written with the vector stores the launch picks from buffer 0's alignment: hold all three to it
        // (16 bytes where four pixels per thread are possible, 8 always)
        for (int k = 0; k < 3; k++) {
            if (!is_aligned(rot[k], ((w % 4) == 0 && (frame_stride % 4) == 0) ? 16 : 8))
                return fail(c, LUMAHIP_ERR_ARG, "the three frame buffers must be %d-byte aligned", ((w % 4) == 0 && (frame_stride % 4) == 0) ? 16 : 8);
            rot_planes[k] = rot[0] + (size_t)k * w * h;
        }
        if (rot[0] == rot[1] || rot[1] == rot[2] || rot[0] == rot[2])
            return fail(c, LUMAHIP_ERR_ARG, "the three frame buffers must be distinct");
        // buffer k holds frames k, k + 3, ...: ceil((nframes - k) / 3) frames, the last one 3*w*h floats long
        for (int i = 0; i < 3; i++)
            for (int j = i + 1; j < 3; j++) {
                const size_t ni = (nframes + 2 - (unsigned)i) / 3, nj = (nframes + 2 - (unsigned)j) / 3;
                if (ni == 0 || nj == 0)
                    continue;
                const uintptr_t bi = (uintptr_t)rot[i], bj = (uintptr_t)rot[j];
                const size_t ei = ((ni - 1) * frame_stride + (size_t)3 * w * h) * sizeof(float);
                const size_t ej = ((nj - 1) * frame_stride + (size_t)3 * w * h) * sizeof(float);
                if (bi < bj + ej && bj < bi + ei)
                    return fail(c, LUMAHIP_ERR_ARG, "the three frame buffers must not overlap (buffers %d and %d do over this batch)", i, j);
            }
    }
    float *const *rgb = rot ? rot_planes : rgb_in;
    const bool have_rgb = rgb && rgb[0];
    if (!c || (!have_rgb && !dp.rgba) || (have_rgb && (!rgb[1] || !rgb[2])) || !planes || !stride || !pfs || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    if ((rc = check_layout(c, w, h, profile, nframes, (have_rgb && !rot) ? rgb : nullptr, frame_stride, stride, pfs)))
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    const bool gl = !c->lut_in_lds;
    float *const none[3] = {nullptr, nullptr, nullptr};
    float *const *out = have_rgb ? rgb : none;
    const bool al16 = is_aligned(out[0], 16) && is_aligned(out[1], 16) && is_aligned(out[2], 16);
    int vw = (!gl && (w % 4) == 0 && al16 && (frame_stride % 4) == 0) ? 4 : 2;
    if (c->dec_vw == 2)   // lumahip_tune("dec_vw", 2): two pixels per thread and row even where four are possible (measurements)
        vw = 2;
    if (!is_aligned(out[0], 8) || !is_aligned(out[1], 8) || !is_aligned(out[2], 8) || (frame_stride % 2) != 0)
        return fail(c, LUMAHIP_ERR_ARG, "colour planes must be 8-byte aligned and the frame stride even");
    if (dp.rgba && (!is_aligned(dp.rgba, 4) || (dp.stride % 4) != 0 || (dp.frame_stride % 4) != 0 || dp.stride < (int)(4 * w)))
        return fail(c, LUMAHIP_ERR_ARG, "display buffer must be 4-byte aligned with stride >= 4*w");
    DecArgs a{};
    a.q = c->q;
    const bool yt = cs_eff == CS_YCBCR && c->q.ytab && !gl && dp.rgba == nullptr;
    if (!yt)
        a.q.ytab = nullptr;
    size_t lds = lds_bytes(c, false, cs_eff);
    if (cs_eff == CS_YCBCR && c->q.ytab && !yt)
        lds -= ((size_t)(c->q.lut_len + c->q.pad) * 4 + 15) & ~(size_t)15;   // (display variant: no y table)
    const int threads = block_threads_for(c, lds, false, cs_eff == CS_YCBCR);
    if (!make_geom(a.g, w, h, vw, threads / 64, nframes))
        return fail(c, LUMAHIP_ERR_ARG, "batch too large: more than 2^31 tiles in one launch");
    for (int k = 0; k < 3; k++)
        a.dst[k] = out[k];
    if (rot) {
        for (int k = 0; k < 3; k++)
            a.rot[k] = rot[k];
        a.rot_on = 1;
        a.g.interleave = 1;
    }
    a.frame_stride = frame_stride;
    a.sc = sc;
    a.bps = bps;
    a.aligned = 1;
    a.disp = dp.rgba;
    a.disp_stride = dp.stride;
    a.disp_frame_stride = dp.frame_stride;
    a.exposure = dp.exposure;
    a.inv_gamma = 1.0f / dp.gamma;
    a.do_tmo = dp.do_tmo;
    a.ldr_sim = dp.ldr_sim;
    for (int p = 0; p < 3; p++) {
        if (!planes[p])
            return fail(c, LUMAHIP_ERR_ARG, "null plane %d", p);
        a.src[p] = planes[p];
        a.stride[p] = stride[p];
        a.src_frame_stride[p] = pfs[p];
        const size_t ub = (size_t)((p && sub) ? vw / 2 : vw) * bps;
        if (!is_aligned(planes[p], ub) || (stride[p] % (int)ub) != 0 || (pfs[p] % ub) != 0)
            a.aligned = 0;
    }
    a.q.cs = cs_eff;
    const float *rb = nullptr;
    uint32_t *rb_flag = nullptr;   // this launch's feedback word (LagPolicy)
    if (yt && (rc = rb_table_for(c, sc, &rb)))
        return rc;
    // mode 1: the kernels with the tables test every wave's codes for locality first (rb_wave_local); on a stream none of whose
    // waves ever passes, that test and the larger kernel cost 3-4 % for nothing, so launches that report no gathers send the
    // following ones to the plain kernels for a while
    if (rb && c->rb_mode == 1 && !lag_policy_next(c->rb_pol, &rb_flag))
        rb = nullptr;
    a.rb = rb;
    a.rb_flag = rb_flag;
    a.rb_plane = (size_t)c->q.lut_len * ((size_t)c->q.maxC + 1);
    a.rb_near_y = c->rb_mode == 2 ? -1 : c->rb_near_y;
    a.rb_near_c = c->rb_near_c;
    if (rb)
        c->rb_launches++;
    dec_kernel_t kern = pick_dec(cs_eff, sub, vw, gl, dp.rgba != nullptr, yt, rb != nullptr);
    LagLaunchGuard rb_guard{c->rb_pol, rb_flag};   // (a return before the launch takes the word back: the policy must not wait for it)
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // few_writers: the 4:2:0 16-bit kernels of the HBM-bound colour spaces (12 of 15 bytes per pixel are writes); 2 when the three
    // colour planes of the batch are separate buffers (no plane starts inside another plane's extent over the batch) -- the layout
    // a caller uses to spread the three write streams over the HBM region groups (lumahip_decode_frames_device_planar)
    int few_writers = (sub && bps == 2 && cs_eff != CS_YCBCR && dp.rgba == nullptr) ? 1 : 0;
    if (few_writers && have_rgb && (rot || planes_are_separate_buffers(rgb, frame_stride, nframes, w, h)))
        few_writers = 2;
    const int grid = grid_for(c, threads, a.g.totalTiles, 1, few_writers, cs_eff == CS_YCBCR);
    hipStream_t s = launch_stream(c, lanes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, a);
    rb_guard.launched = true;
    if (rb_flag && (rc = lag_policy_launched(c, c->rb_pol, s)))
        return rc;
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}


int array_launch(lumahip_ctx *c, const float *d_in, float *d_out, size_t n, unsigned ch, bool quant)
{
    HIPCHK(c, hipSetDevice(c->device));
    if (quant)
        if (int rc = ensure_search_index(c))
            return rc;
    QArrArgs a{};
    a.q = c->q;
    a.in = d_in;
    a.out = d_out;
    a.n = n;
    // src/luma_quantizer.cpp:219,251: LUT path for ch 0 and for every channel of RGB / XYZ
    a.lut_channel = (ch == 0 || c->q.cs == CS_RGB || c->q.cs == CS_XYZ) ? 1 : 0;
    long grid = (long)((n + 255) / 256);
    if (grid > (long)c->num_cu * 8)
        grid = (long)c->num_cu * 8;
    if (quant) {
        const size_t lds = lds_bytes(c, true, CS_PACK);  // no powf tables for the array kernels
        void (*kern)(const QArrArgs) = k_quantize_array<2>;
        switch (c->q.mode) {
        case LUT_LITERAL_LDS: kern = k_quantize_array<0>; break;
        case LUT_THRESH_LDS: kern = k_quantize_array<3>; break;
        case LUT_THRESH_GLOBAL: kern = k_quantize_array<4>; break;
        case LUT_LINKEY_LDS: kern = k_quantize_array<7>; break;
        default: break;
        }
        if (lds > 64 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, c->stream, a);
    } else {
        hipLaunchKernelGGL(k_dequantize_array, dim3((unsigned)grid), dim3(256), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}

}  // namespace lhost

extern "C" int lumahip_rb_table_info(lumahip_ctx *c, float sc, int info[4])
{
    if (!c || !info)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    HIPCHK(c, hipSetDevice(c->device));
    const float *t = nullptr;
    if (c->q.cs == CS_YCBCR && c->q.ytab && c->lut_in_lds)
        if (int rc = lhost::rb_table_for(c, sc, &t))
            return rc;
    info[0] = t != nullptr;
    info[1] = t ? (int)(2 * (size_t)c->q.lut_len * ((size_t)c->q.maxC + 1) * sizeof(float)) : 0;
    info[2] = (int)std::min<unsigned long>(c->rb_launches, 0x7fffffffUL);
    info[3] = (int)std::min<unsigned long>(c->rb_pol.backoff_launches, 0x7fffffffUL);
    return LUMAHIP_OK;
}

extern "C" int lumahip_decode_frames_device(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                            const size_t pfs[3], unsigned nframes, unsigned w, unsigned h, int profile,
                                            float sc, float *rgb, size_t frame_stride)
{
    if (!rgb)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c)
        return LUMAHIP_ERR_ARG;
    const size_t n = (size_t)w * h;
    float *const pl[3] = {rgb, rgb + n, rgb + 2 * n};
    return decode_impl(c, planes, stride, pfs, nframes, w, h, profile, sc, pl, frame_stride, DisplayParams(), c->q.cs, true);
}

extern "C" int lumahip_decode_frames_device_rotating(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                                     const size_t pfs[3], unsigned nframes, unsigned w, unsigned h, int profile,
                                                     float sc, float *const bases[3], size_t frame_stride)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!bases)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    return decode_impl(c, planes, stride, pfs, nframes, w, h, profile, sc, nullptr, frame_stride, DisplayParams(), c->q.cs, true, bases);
}

extern "C" int lumahip_decode_frames_device_planar(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                                   const size_t pfs[3], unsigned nframes, unsigned w, unsigned h, int profile,
                                                   float sc, float *const rgb_planes[3], size_t frame_stride)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!rgb_planes || !rgb_planes[0])
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    return decode_impl(c, planes, stride, pfs, nframes, w, h, profile, sc, rgb_planes, frame_stride, DisplayParams(), c->q.cs, true);
}

// Traffic probe: the loads and stores of k_decode<., 4:2:0, VW=4> for 16-bit planes with NO arithmetic (an xor keeps every
// loaded word live): 3 B read + 12 B written per pixel, same tile order, same software pipeline (the next unit's sample
// loads go out after the current unit's stores), same non-temporal accesses.  Its run time is what the memory system alone
// needs for the decode traffic mix; bench.py reports the decode kernel's time as a fraction of it.
namespace lh {
__global__ __launch_bounds__(256) void k_decode_traffic_probe(const DecArgs a)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int G = gridDim.x;
    DecUnit<true, 4> cur, nxt;
    dec_load<true, 4>(cur, a, blockIdx.x, tx, ty, NW);
    for (int t = blockIdx.x; t < a.g.totalTiles; t += G) {
        if (cur.valid) {
            const size_t off = (size_t)cur.f * a.frame_stride + (size_t)(2 * cur.uy) * a.g.w + (size_t)cur.ux * 4;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int k = c == 0 ? 0 : (c == 1 ? cur.c1[0] ^ cur.c1[1] : cur.c2[0] ^ cur.c2[1]);
                float r0[4], r1[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    r0[i] = __int_as_float(cur.y[0][i] ^ k);
                    r1[i] = __int_as_float(cur.y[1][i] ^ k);
                }
                store_px<4>(a.dst[c] + off, r0);
                store_px<4>(a.dst[c] + off + a.g.w, r1);
            }
        }
        dec_load<true, 4>(nxt, a, t + G, tx, ty, NW);
        cur = nxt;
    }
}
}  // namespace lh

extern "C" int lumahip_probe_decode_traffic_device(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                                   const size_t pfs[3], unsigned nframes, unsigned w, unsigned h,
                                                   float *const rgb_planes[3], size_t frame_stride, int iters, float *avg_ms)
{
    if (!c || !rgb_planes || !planes || !stride || !pfs || nframes == 0 || iters <= 0 || !avg_ms)
        return fail(c, LUMAHIP_ERR_ARG, "bad argument");
    if (w == 0 || h == 0 || (w % 4) || (h & 1) || (frame_stride % 4))
        return fail(c, LUMAHIP_ERR_ARG, "the traffic probe needs w %% 4 == 0, even h and 16-byte aligned frames");
    for (int k = 0; k < 3; k++)
        if (!rgb_planes[k] || !is_aligned(rgb_planes[k], 16))
            return fail(c, LUMAHIP_ERR_ARG, "the traffic probe needs 16-byte aligned colour planes");
    for (int p = 0; p < 3; p++)
        if (!planes[p] || !is_aligned(planes[p], 8) || (stride[p] % (p ? 4 : 8)) || (pfs[p] % 8))
            return fail(c, LUMAHIP_ERR_ARG, "the traffic probe needs 8-byte aligned 16-bit 4:2:0 planes");
    HIPCHK(c, hipSetDevice(c->device));
    DecArgs a{};
    const int threads = 256;
    if (!make_geom(a.g, w, h, 4, threads / 64, nframes))
        return fail(c, LUMAHIP_ERR_ARG, "batch too large");
    for (int k = 0; k < 3; k++)
        a.dst[k] = rgb_planes[k];
    a.frame_stride = frame_stride;
    a.bps = 2;
    a.aligned = 1;
    for (int p = 0; p < 3; p++) {
        a.src[p] = planes[p];
        a.stride[p] = stride[p];
        a.src_frame_stride[p] = pfs[p];
    }
    const int grid = grid_for(c, threads, a.g.totalTiles, 1, planes_are_separate_buffers(rgb_planes, frame_stride, nframes, w, h) ? 2 : 1, false);
    EventPair ev;
    HIPCHK(c, ev.create());
    HIPCHK(c, hipEventRecord(ev.e0, c->stream));
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(k_decode_traffic_probe, dim3(grid), dim3(threads), 0, c->stream, a);
    HIPCHK(c, hipEventRecord(ev.e1, c->stream));
    HIPCHK(c, hipEventSynchronize(ev.e1));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    HIPCHK(c, hipGetLastError());
    *avg_ms = ms / iters;
    return LUMAHIP_OK;
}

extern "C" int lumahip_decode_display_frames_device(lumahip_ctx *c, const unsigned char *const planes[3],
                                                    const int stride[3], const size_t pfs[3], unsigned nframes, unsigned w,
                                                    unsigned h, int profile, float sc, float *rgb_or_null,
                                                    size_t frame_stride, unsigned char *rgba, int rgba_stride,
                                                    size_t rgba_frame_stride, float exposure, float gamma, int do_tmo,
                                                    int ldr_sim)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!rgba || !(gamma > 0.0f))
        return fail(c, LUMAHIP_ERR_ARG, "display output needs a buffer and gamma > 0");
    DisplayParams dp;
    dp.rgba = rgba;
    dp.stride = rgba_stride;
    dp.frame_stride = rgba_frame_stride;
    dp.exposure = exposure;
    dp.gamma = gamma;
    dp.do_tmo = do_tmo;
    dp.ldr_sim = ldr_sim;
    const size_t n = (size_t)w * h;
    float *const pl[3] = {rgb_or_null, rgb_or_null ? rgb_or_null + n : nullptr, rgb_or_null ? rgb_or_null + 2 * n : nullptr};
    return decode_impl(c, planes, stride, pfs, nframes, w, h, profile, sc, pl, frame_stride, dp, c->q.cs);
}

extern "C" int lumahip_quantize_array_device(lumahip_ctx *c, const float *in_dev, float *out_dev, size_t n, unsigned ch)
{
    if (!c || !in_dev || !out_dev)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    return n ? array_launch(c, in_dev, out_dev, n, ch, true) : LUMAHIP_OK;
}

extern "C" int lumahip_dequantize_array_device(lumahip_ctx *c, const float *in_dev, float *out_dev, size_t n, unsigned ch)
{
    if (!c || !in_dev || !out_dev)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    return n ? array_launch(c, in_dev, out_dev, n, ch, false) : LUMAHIP_OK;
}

