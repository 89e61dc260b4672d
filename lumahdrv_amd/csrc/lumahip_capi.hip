// lumahip_capi.hip -- implementation of include/lumahip.h: context, quantizer upload, kernel dispatch.
// Host-side only logic here; the arithmetic is in luma_device.hpp / luma_kernels.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lumahip.h"
#include "luma_kernels.hpp"
#include "lut_index.hpp"

using namespace lh;

// Largest search table (encode: threshold records, decode: the luminance table) a workgroup stages in LDS; beyond it
// the table is read from global memory (L2-resident).  gfx950 has 160 KiB of LDS per CU; tables beyond 53 KiB run as one
// or two 1024-thread workgroups per CU -- for the 136 KiB of PQ 13-bit records that is still twice as fast as gathering
// them from L2 (346 against 182 Gpixel/s, profiles/r02_perf_matrix.txt).  LUMAHIP_LDS_TABLE_MAX_KB overrides it.
static constexpr size_t LUMAHIP_LDS_TABLE_MAX_DEFAULT = 144 * 1024;
static constexpr size_t LUMAHIP_LDS_PER_WORKGROUP = 160 * 1024;

struct lumahip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    int num_cu = 256;

    bool have_quant = false;
    int ptf = 0;
    unsigned bitdepth = 0, bitdepthC = 0;
    QuantDev q{};
    ThreshIndex tix;
    float *d_lut = nullptr;
    uint32_t *d_rec = nullptr;
    bool lut_in_lds = true;  // decode side: tables up to 12 bits are staged in LDS
    float minLum = 0.0f;

    // staging for the _host entry points
    float *d_frame = nullptr;
    size_t d_frame_cap = 0;
    unsigned char *d_planes = nullptr;
    size_t d_planes_cap = 0;
    float *d_stats = nullptr;
    float *d_arr = nullptr;
    size_t d_arr_cap = 0;

    // 3-slot pipeline of the batched host entry points (H2D / kernel / D2H on three streams)
    struct Slot {
        float *d_frame = nullptr;
        unsigned char *d_planes = nullptr;
        float *d_stats = nullptr;
        hipEvent_t h2d = nullptr, kern = nullptr, d2h = nullptr;
    } slot[3];
    size_t slot_frame_cap = 0, slot_planes_cap = 0;
    hipStream_t s_h2d = nullptr, s_kern = nullptr, s_d2h = nullptr;
    float *h_stats = nullptr;  // pinned, 3 floats per frame
    size_t h_stats_cap = 0;

    // Pinned staging for pageable caller memory (see xfer_h2d): two chunks per direction, ping-pong
    struct Stage {
        unsigned char *h = nullptr;
        hipEvent_t ev = nullptr;
        bool pending = false;  // a DMA that reads / writes this chunk may still be in flight
    } stage_up[2], stage_dn[2];
    float *h_small = nullptr;  // pinned scratch for the few-float readbacks

    int block_threads = 256;
    bool block_forced = false;
    bool allow_alias = false;  // LUMAHIP_ALLOW_ALIASED_FRAMES=1: measurement tools alias all frames of a batch onto one
    int blocks_per_cu = 0;  // 0 = occupancy query
    long grid_override[2] = {0, 0};
    size_t lds_table_max = LUMAHIP_LDS_TABLE_MAX_DEFAULT;
};

static int fail(lumahip_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(c, expr)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail((c), LUMAHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                             \
    } while (0)

extern "C" int lumahip_abi_version(void) { return LUMAHIP_ABI_VERSION; }

extern "C" int lumahip_device_count(int *count)
{
    if (!count)
        return LUMAHIP_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return LUMAHIP_ERR_HIP;
    }
    *count = n;
    return LUMAHIP_OK;
}

extern "C" int lumahip_create(lumahip_ctx **out, int device)
{
    if (!out)
        return LUMAHIP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return LUMAHIP_ERR_HIP;  // no CPU fallback: the path needs a HIP device
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess)
            return LUMAHIP_ERR_HIP;
    }
    if (device >= n)
        return LUMAHIP_ERR_ARG;
    lumahip_ctx *c = new lumahip_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return LUMAHIP_ERR_HIP;
    }
    c->stream = c->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        c->num_cu = prop.multiProcessorCount;
    if (const char *e = getenv("LUMAHIP_BLOCK")) {
        int v = atoi(e);
        if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) {
            c->block_threads = v;
            c->block_forced = true;
        }
    }
    if (const char *e = getenv("LUMAHIP_BLOCKS_PER_CU"))
        c->blocks_per_cu = atoi(e);
    if (const char *e = getenv("LUMAHIP_GRID_ENC"))
        c->grid_override[0] = atol(e);
    if (const char *e = getenv("LUMAHIP_GRID_DEC"))
        c->grid_override[1] = atol(e);
    if (const char *e = getenv("LUMAHIP_ALLOW_ALIASED_FRAMES"))
        c->allow_alias = atoi(e) != 0;
    if (const char *e = getenv("LUMAHIP_LDS_TABLE_MAX_KB")) {
        const long kb = atol(e);
        if (kb >= 0 && kb <= 152)
            c->lds_table_max = (size_t)kb * 1024;
    }
    *out = c;
    return LUMAHIP_OK;
}

extern "C" void lumahip_destroy(lumahip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_rec);
    (void)hipFree(c->d_frame);
    (void)hipFree(c->d_planes);
    (void)hipFree(c->d_stats);
    (void)hipFree(c->d_arr);
    for (auto &sl : c->slot) {
        (void)hipFree(sl.d_frame);
        (void)hipFree(sl.d_planes);
        (void)hipFree(sl.d_stats);
        if (sl.h2d) (void)hipEventDestroy(sl.h2d);
        if (sl.kern) (void)hipEventDestroy(sl.kern);
        if (sl.d2h) (void)hipEventDestroy(sl.d2h);
    }
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    for (auto *st : {&c->stage_up[0], &c->stage_up[1], &c->stage_dn[0], &c->stage_dn[1]}) {
        if (st->ev) (void)hipEventSynchronize(st->ev);
        if (st->h) (void)hipHostFree(st->h);
        if (st->ev) (void)hipEventDestroy(st->ev);
    }
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d);
    if (c->s_kern) (void)hipStreamDestroy(c->s_kern);
    if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h);
    if (c->own_stream)
        (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char *lumahip_last_error(const lumahip_ctx *c) { return c ? c->err.c_str() : "null context"; }

extern "C" int lumahip_set_stream(lumahip_ctx *c, void *s)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    c->stream = (hipStream_t)s;  // NULL is a valid handle: the device's default (null) stream
    return LUMAHIP_OK;
}

extern "C" int lumahip_reset_stream(lumahip_ctx *c)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    c->stream = c->own_stream;
    return LUMAHIP_OK;
}

extern "C" int lumahip_sync(lumahip_ctx *c)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

// ---------------------------------------------------------------------------------------- quantizer

extern "C" int lumahip_set_quantizer(lumahip_ctx *c, int ptf, unsigned bitdepth, int cs, unsigned bitdepthC,
                                     float maxLum, float minLum, const float *lut, size_t n)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (bitdepth < 1 || bitdepth > 16 || bitdepthC < 1 || bitdepthC > 16)
        return fail(c, LUMAHIP_ERR_ARG, "bit depths must be 1..16 (got %u / %u)", bitdepth, bitdepthC);
    if (!lut || n != ((size_t)1 << bitdepth))
        return fail(c, LUMAHIP_ERR_ARG, "LUT must hold 2^bitdepth = %zu floats (got %zu)", (size_t)1 << bitdepth, n);
    if (ptf < 0 || ptf > 4)
        return fail(c, LUMAHIP_ERR_ARG, "unknown transfer function %d", ptf);
    // an unknown colour space is accepted here, as in the reference (setQuantizer stores it blindly,
    // src/luma_quantizer.cpp:181); the transform entry points then fail the way transformColorSpace does.
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));

    // Search index.  Every monotone finite table gets threshold records (lut_index.hpp): in LDS when they fit
    // lds_table_max, else in global memory
    // (L2-resident).  Anything else (NaNs, decreasing entries -- a decoder may be handed any attachment-434 table)
    // runs the reference's bisection literally.  LUMAHIP_FORCE_LITERAL is the tests' hook for that path.
    c->tix = ThreshIndex();
    // decode side: luminance table (+ Lu'v' chroma table, + the powf tables for YCbCr) staged in LDS
    const size_t powf_b = (cs == CS_YCBCR) ? sizeof(PowfTablesWide) : 0;
    c->lut_in_lds = bitdepthC <= 12 && (n + 4) * sizeof(float) <= std::max<size_t>(c->lds_table_max, 16 * 1024 + 16) &&
                    (n + 4) * sizeof(float) + ((size_t)4 << bitdepthC) + 64 + powf_b <= LUMAHIP_LDS_PER_WORKGROUP;
    int mode = n <= 4096 ? LUT_LITERAL_LDS : LUT_LITERAL_GLOBAL;
    if (!getenv("LUMAHIP_FORCE_LITERAL")) {
        c->tix = build_thresh_index(lut, (int)n, 1 << 19);
        if (c->tix.ok)
            mode = (c->tix.rec.size() * 4 <= c->lds_table_max && c->tix.rec.size() * 4 + 16 + powf_b <= LUMAHIP_LDS_PER_WORKGROUP)
                       ? LUT_THRESH_LDS
                       : LUT_THRESH_GLOBAL;
    }
    const size_t lut_floats = (n + 1 + 3) & ~(size_t)3;  // NaN padding up to a multiple of 16 bytes
    std::vector<float> padded(lut_floats, __builtin_nanf(""));
    memcpy(padded.data(), lut, n * sizeof(float));
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_rec);
    c->d_lut = nullptr;
    c->d_rec = nullptr;
    HIPCHK(c, hipMalloc(&c->d_lut, lut_floats * sizeof(float)));
    HIPCHK(c, hipMemcpy(c->d_lut, padded.data(), lut_floats * sizeof(float), hipMemcpyHostToDevice));
    if (c->tix.ok) {
        std::vector<uint32_t> r((c->tix.rec.size() + 3) & ~(size_t)3, 0u);
        memcpy(r.data(), c->tix.rec.data(), c->tix.rec.size() * sizeof(uint32_t));
        HIPCHK(c, hipMalloc(&c->d_rec, r.size() * sizeof(uint32_t)));
        HIPCHK(c, hipMemcpy(c->d_rec, r.data(), r.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    QuantDev &q = c->q;
    q.lut = c->d_lut;
    q.rec = c->d_rec;
    q.lut_len = (int)n;
    q.pad = (int)(lut_floats - n);
    q.maxVal = (int)n - 1;                                   // (int)pow(2,bitdepth)-1, src/luma_quantizer.cpp:180
    q.mode = mode;
    q.shift = c->tix.ok ? c->tix.shift : 0;
    q.kmin = c->tix.ok ? c->tix.kmin : 0;
    q.nbuckets = c->tix.ok ? c->tix.nbuckets : 0;
    q.maxC = (float)(((unsigned)1 << bitdepthC) - 1);        // src/luma_quantizer.cpp:183
    q.cs = cs;
    q.Lmax = maxLum;
    c->ptf = ptf;
    c->bitdepth = bitdepth;
    c->bitdepthC = bitdepthC;
    c->minLum = minLum;
    c->have_quant = true;
    return LUMAHIP_OK;
}

// host-only view of the threshold records (no GPU, no context): info = {ok, mant_bits, shift, kmin, nbuckets}
extern "C" int lumahip_thresh_index_host(const float *lut, size_t n, int info[5], uint32_t *rec_out, size_t rec_cap)
{
    if (!lut || !info || n < 2 || n > 65536)
        return LUMAHIP_ERR_ARG;
    const ThreshIndex ix = build_thresh_index(lut, (int)n, 1 << 19);
    info[0] = ix.ok ? 1 : 0;
    info[1] = ix.mant_bits;
    info[2] = ix.shift;
    info[3] = ix.kmin;
    info[4] = ix.nbuckets;
    if (rec_out && ix.ok) {
        if (rec_cap < ix.rec.size())
            return LUMAHIP_ERR_ARG;
        memcpy(rec_out, ix.rec.data(), ix.rec.size() * sizeof(uint32_t));
    }
    return LUMAHIP_OK;
}

// dynamic LDS of the encode-side kernels (search tables) and of the decode-side kernels (the table itself)
static size_t lds_bytes(const lumahip_ctx *c, bool encode_side, int cs_eff)
{
    const QuantDev &q = c->q;
    size_t b = 0;
    const size_t lut_b = ((size_t)(q.lut_len + q.pad) * 4 + 15) & ~(size_t)15;
    if (encode_side) {
        if (q.mode == LUT_LITERAL_LDS)
            b += lut_b;
        if (q.mode == LUT_THRESH_LDS)
            b += ((size_t)q.nbuckets * 4 + 15) & ~(size_t)15;
    } else if (c->lut_in_lds) {
        b += lut_b;
        if (cs_eff == CS_LUV)
            b += (((size_t)q.maxC + 1) * 4 + 15) & ~(size_t)15;  // u'v' table of the Lu'v' decode kernels
    }
    if (cs_eff == CS_YCBCR)
        b += sizeof(PowfTablesWide);
    return b;
}

// Workgroup size of the fused kernels: 256 threads unless LUMAHIP_BLOCK says otherwise; search tables beyond
// 32 KiB per workgroup would leave too few waves per CU at that size (160 KiB of LDS per CU), so the workgroup
// grows with the table.
//   - `few_waves` (the encode kernels of the HBM-bound colour spaces on long launches, see grid_for): three 256-thread
//     workgroups per CU are the fastest configuration measured, so the workgroup stays at 256 threads as long as three
//     copies of the table fit the CU's LDS (LOG-12's 42 KiB of records: 3.9 % faster than four 512-thread workgroups).
static int block_threads_for(const lumahip_ctx *c, size_t lds, bool few_waves = false)
{
    if (c->block_forced)
        return c->block_threads;
    if (few_waves && c->block_threads == 256 && 3 * lds <= LUMAHIP_LDS_PER_WORKGROUP)
        return 256;
    if (lds > 53 * 1024)
        return 1024;
    if (lds > 32 * 1024)
        return 512;
    return c->block_threads;
}

extern "C" int lumahip_quantizer_info(const lumahip_ctx *c, int info[5])
{
    if (!c || !info)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return LUMAHIP_ERR_STATE;
    info[0] = c->q.mode;
    info[1] = c->tix.ok ? c->tix.mant_bits : 0;
    info[2] = c->q.nbuckets;
    info[3] = c->tix.ok ? c->tix.shift : 0;
    info[4] = (int)lds_bytes(c, true, c->q.cs);
    return LUMAHIP_OK;
}

// ---------------------------------------------------------------------------------------- dispatch

typedef void (*enc_kernel_t)(const EncArgs);
typedef void (*dec_kernel_t)(const DecArgs);

template <int CS, bool SUB>
static enc_kernel_t pick_enc2(int vw, int mode)
{
    if (mode == LUT_THRESH_LDS)
        return vw == 4 ? k_encode<CS, SUB, 4, 3> : k_encode<CS, SUB, 2, 3>;
    if (mode == LUT_THRESH_GLOBAL)
        return vw == 4 ? k_encode<CS, SUB, 4, 4> : k_encode<CS, SUB, 2, 4>;
    if (mode == LUT_LITERAL_LDS)
        return k_encode<CS, SUB, 2, 0>;
    return k_encode<CS, SUB, 2, 2>;
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

template <int CS, bool SUB>
static dec_kernel_t pick_dec2(int vw, bool gl, bool disp)
{
    if (disp) {
        if (gl)
            return k_decode<CS, SUB, 2, true, true>;
        return vw == 4 ? k_decode<CS, SUB, 4, false, true> : k_decode<CS, SUB, 2, false, true>;
    }
    if (gl)
        return k_decode<CS, SUB, 2, true>;
    return vw == 4 ? k_decode<CS, SUB, 4, false> : k_decode<CS, SUB, 2, false>;
}

static dec_kernel_t pick_dec(int cs, bool sub, int vw, bool gl, bool disp)
{
    switch (cs) {
    case CS_LUV: return sub ? pick_dec2<CS_LUV, true>(vw, gl, disp) : pick_dec2<CS_LUV, false>(vw, gl, disp);
    case CS_RGB: return sub ? pick_dec2<CS_RGB, true>(vw, gl, disp) : pick_dec2<CS_RGB, false>(vw, gl, disp);
    case CS_YCBCR: return sub ? pick_dec2<CS_YCBCR, true>(vw, gl, disp) : pick_dec2<CS_YCBCR, false>(vw, gl, disp);
    case CS_XYZ: return sub ? pick_dec2<CS_XYZ, true>(vw, gl, disp) : pick_dec2<CS_XYZ, false>(vw, gl, disp);
    case CS_PACK: return sub ? pick_dec2<CS_PACK, true>(vw, gl, disp) : pick_dec2<CS_PACK, false>(vw, gl, disp);
    }
    return nullptr;
}

static int check_geom(lumahip_ctx *c, unsigned w, unsigned h, int profile, int cs_eff)
{
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set (call lumahip_set_quantizer first)");
    if (w == 0 || h == 0 || (w & 1) || (h & 1))
        return fail(c, LUMAHIP_ERR_ARG, "Invalid frame size %ux%u (must be even, non-zero)", w, h);
    if (profile < 0 || profile > 3)
        return fail(c, LUMAHIP_ERR_ARG, "profile must be 0..3 (got %d)", profile);
    if (cs_eff < 0 || cs_eff > CS_PACK)
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "Unrecognized color transformation (colour space %d)", cs_eff);
    return LUMAHIP_OK;
}

static bool make_geom(FrameGeom &g, unsigned w, unsigned h, int vw, int nw, unsigned nframes)
{
    g.w = (int)w;
    g.h = (int)h;
    g.unitsX = (int)w / vw;
    g.unitsY = (int)h / 2;
    g.tilesX = (g.unitsX + 63) / 64;
    g.tilesY = (g.unitsY + nw - 1) / nw;
    g.tilesPerFrame = g.tilesX * g.tilesY;
    const long long total = (long long)g.tilesPerFrame * nframes;
    if (w > 0x7fffffffu / 4 || h > 0x7fffffffu / 4 || total > 0x7fffffffLL)
        return false;  // tile indices are 32-bit
    g.totalTiles = (int)total;
    return true;
}

// Persistent workgroups: how many of them.  dir 0 = encode, 1 = decode.  The default is 2048 threads' worth per CU (8
// workgroups of 256), i.e. MORE than are resident at once for most kernels: the surplus is dispatched as resident ones
// retire, which evens out the tail of short launches.  The rules below are for long (batched) launches, each one found by
// running both settings in one process (tools/ab_inproc.py).  LUMAHIP_GRID_ENC / LUMAHIP_GRID_DEC (absolute) and
// LUMAHIP_BLOCKS_PER_CU (per CU, both directions) are measurement overrides.
static int grid_for(const lumahip_ctx *c, int threads, int total_tiles, int dir, bool few_writers = false, bool ycbcr = false)
{
    int per_cu = c->blocks_per_cu > 0 ? c->blocks_per_cu : 2048 / threads;
    // The 4:2:0 16-bit decode kernels write 12 of their 15 bytes per pixel, and fewer concurrent writers suit the memory
    // system better: 5 workgroups of 256 threads per CU instead of 8 is 2.7-3.6 % faster on batched launches, both builds
    // in one process (profiles/r02_grid_sweep.txt; the same change is 4 % SLOWER for 4:4:4 Lu'v' and 10 % slower for the
    // 8-bit profiles, so it is theirs only).  Only where the launch is long enough for the coarser tail not to matter.
    if (few_writers && c->blocks_per_cu == 0 && threads == 256 && total_tiles >= 12L * c->num_cu * 5)
        per_cu = 5;
    // The encode kernels with 256-thread workgroups (tables up to 32 KiB; not YCbCr, which is VALU-bound and wants the
    // waves) run 3-6 % faster on long launches with 3 workgroups per CU than with 8 -- every colour space / profile
    // variant, same build in one process (tools/ab_encode_grid.sh, profiles/r02_grid_sweep.txt); 6 per CU is 9 % SLOWER,
    // 4 about as good as 3.  Fewer resident waves draw less power at the package limit and keep fewer streams open in the
    // memory system.  Short launches keep 8 per CU for their tail.
    if (dir == 0 && c->blocks_per_cu == 0 && threads == 256 && total_tiles >= 40L * c->num_cu * 3)
        per_cu = 3;
    // The YCbCr kernels are VALU-bound and only three of their 512-thread workgroups (49 KiB of LDS each) are resident
    // per CU: many more, smaller static shares balance the CUs better than one share per resident workgroup -- 18 per CU
    // is 4.9 % faster than 4 for encode, 12 per CU 4.3 % for decode (same build in one process, profiles/r02_grid_sweep.txt).
    if (ycbcr && c->blocks_per_cu == 0 && threads == 512 && total_tiles >= 8L * c->num_cu * 18)
        per_cu = dir == 0 ? 18 : 12;
    long g = (long)c->num_cu * per_cu;
    if (c->grid_override[dir] > 0)
        g = c->grid_override[dir];
    if (g > total_tiles)
        g = total_tiles;
    if (g < 1)
        g = 1;
    return (int)g;
}

static bool is_aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

// a pair of timing events that cannot leak on an early return
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create()
    {
        hipError_t e = hipEventCreate(&e0);
        return e != hipSuccess ? e : hipEventCreate(&e1);
    }
    ~EventPair()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

__global__ void k_init_stats(float *s, int nframes)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nframes) {
        s[3 * i + 0] = 0.0f;
        s[3 * i + 1] = __builtin_inff();
        s[3 * i + 2] = -__builtin_inff();
    }
}

// rows and bytes per row of plane p as vpx_img_alloc lays it out (src/luma_encoder.cpp:121-128)
static void plane_dims(unsigned w, unsigned h, int profile, int p, int &rows, int &row_bytes)
{
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    rows = (p && sub) ? (int)(h + 1) / 2 : (int)h;
    row_bytes = ((p && sub) ? (int)(w + 1) / 2 : (int)w) * bps;
}

// the device entry points take caller-chosen strides: reject layouts in which rows or frames would overlap or the
// kernels would write outside a plane (negative / too small strides, frame strides smaller than a frame)
static int check_layout(lumahip_ctx *c, unsigned w, unsigned h, int profile, unsigned nframes, size_t frame_stride,
                        const int stride[3], const size_t pfs[3])
{
    for (int p = 0; p < 3; p++) {
        int rows, row_bytes;
        plane_dims(w, h, profile, p, rows, row_bytes);
        if (stride[p] < row_bytes)
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: stride %d < row bytes %d", p, stride[p], row_bytes);
        if (nframes > 1 && !c->allow_alias && pfs[p] < (size_t)rows * (size_t)stride[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: frame stride %zu < plane size %zu", p, pfs[p], (size_t)rows * stride[p]);
    }
    if (nframes > 1 && !c->allow_alias && frame_stride < (size_t)3 * w * h)
        return fail(c, LUMAHIP_ERR_ARG, "frame stride %zu < 3*w*h = %zu floats", frame_stride, (size_t)3 * w * h);
    return LUMAHIP_OK;
}

static int encode_frames_device_impl(lumahip_ctx *c, const float *rgb, size_t frame_stride, unsigned nframes, unsigned w,
                                     unsigned h, float sc, int profile, unsigned char *const planes[3], const int stride[3],
                                     const size_t pfs[3], float *stats, int cs_eff)
{
    if (!c || !rgb || !planes || !stride || !pfs || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    if ((rc = check_layout(c, w, h, profile, nframes, frame_stride, stride, pfs)))
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    const int mode = c->q.mode;
    const bool fast_search = (mode == LUT_THRESH_LDS || mode == LUT_THRESH_GLOBAL);
    int vw = (fast_search && (w % 4) == 0 && is_aligned(rgb, 16) && (frame_stride % 4) == 0) ? 4 : 2;
    if (!is_aligned(rgb, 8) || (frame_stride % 2) != 0)
        return fail(c, LUMAHIP_ERR_ARG, "frame base must be 8-byte aligned and frame stride even");
    EncArgs a{};
    a.q = c->q;
    const size_t lds = lds_bytes(c, true, cs_eff);
    const bool long_launch = (unsigned long long)w * h * nframes >= 60000000ull;   // >= 7 4K frames
    const int threads = block_threads_for(c, lds, long_launch && cs_eff != CS_YCBCR);
    if (!make_geom(a.g, w, h, vw, threads / 64, nframes))
        return fail(c, LUMAHIP_ERR_ARG, "batch too large: more than 2^31 tiles in one launch");
    a.src = rgb;
    a.frame_stride = frame_stride;
    a.sc = sc;
    a.bps = bps;
    a.stats = stats;
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
    enc_kernel_t kern = pick_enc(cs_eff, sub, vw, mode);
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = grid_for(c, threads, a.g.totalTiles, 0, false, cs_eff == CS_YCBCR);
    if (stats)
        hipLaunchKernelGGL(k_init_stats, dim3((nframes + 255) / 256), dim3(256), 0, c->stream, stats, (int)nframes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}

extern "C" int lumahip_encode_frames_device(lumahip_ctx *c, const float *rgb, size_t frame_stride, unsigned nframes,
                                            unsigned w, unsigned h, float sc, int profile,
                                            unsigned char *const planes[3], const int stride[3],
                                            const size_t pfs[3], float *stats)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    return encode_frames_device_impl(c, rgb, frame_stride, nframes, w, h, sc, profile, planes, stride, pfs, stats, c->q.cs);
}

struct DisplayParams {
    unsigned char *rgba = nullptr;
    int stride = 0;
    size_t frame_stride = 0;
    float exposure = 1.0f, gamma = 2.2f;
    int do_tmo = 0, ldr_sim = 0;
};

static int decode_impl(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3], const size_t pfs[3],
                       unsigned nframes, unsigned w, unsigned h, int profile, float sc, float *rgb, size_t frame_stride,
                       const DisplayParams &dp, int cs_eff)
{
    if (!c || (!rgb && !dp.rgba) || !planes || !stride || !pfs || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    if ((rc = check_layout(c, w, h, profile, nframes, rgb ? frame_stride : (size_t)3 * w * h, stride, pfs)))
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    const bool gl = !c->lut_in_lds;
    int vw = (!gl && (w % 4) == 0 && is_aligned(rgb, 16) && (frame_stride % 4) == 0) ? 4 : 2;
    if (!is_aligned(rgb, 8) || (frame_stride % 2) != 0)
        return fail(c, LUMAHIP_ERR_ARG, "frame base must be 8-byte aligned and frame stride even");
    if (dp.rgba && (!is_aligned(dp.rgba, 4) || (dp.stride % 4) != 0 || (dp.frame_stride % 4) != 0 || dp.stride < (int)(4 * w)))
        return fail(c, LUMAHIP_ERR_ARG, "display buffer must be 4-byte aligned with stride >= 4*w");
    DecArgs a{};
    a.q = c->q;
    const size_t lds = lds_bytes(c, false, cs_eff);
    const int threads = block_threads_for(c, lds);
    if (!make_geom(a.g, w, h, vw, threads / 64, nframes))
        return fail(c, LUMAHIP_ERR_ARG, "batch too large: more than 2^31 tiles in one launch");
    a.dst = rgb;
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
    dec_kernel_t kern = pick_dec(cs_eff, sub, vw, gl, dp.rgba != nullptr);
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = grid_for(c, threads, a.g.totalTiles, 1, sub && bps == 2 && cs_eff != CS_YCBCR && dp.rgba == nullptr, cs_eff == CS_YCBCR);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
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
    return decode_impl(c, planes, stride, pfs, nframes, w, h, profile, sc, rgb, frame_stride, DisplayParams(), c->q.cs);
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
    return decode_impl(c, planes, stride, pfs, nframes, w, h, profile, sc, rgb_or_null, frame_stride, dp, c->q.cs);
}

typedef void (*xf_kernel_t)(const XfArgs);
static xf_kernel_t pick_xf(int cs, bool fwd)
{
    switch (cs) {
    case CS_LUV: return fwd ? k_transform<CS_LUV, true> : k_transform<CS_LUV, false>;
    case CS_RGB: return fwd ? k_transform<CS_RGB, true> : k_transform<CS_RGB, false>;
    case CS_YCBCR: return fwd ? k_transform<CS_YCBCR, true> : k_transform<CS_YCBCR, false>;
    case CS_XYZ: return fwd ? k_transform<CS_XYZ, true> : k_transform<CS_XYZ, false>;
    }
    return nullptr;
}

extern "C" int lumahip_transform_color_space_device(lumahip_ctx *c, float *frames, size_t frame_stride, unsigned nframes,
                                                    unsigned w, unsigned h, int toCs, float sc)
{
    if (!c || !frames || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    if (w == 0 || h == 0)
        return fail(c, LUMAHIP_ERR_ARG, "empty frame");
    if (c->q.cs < 0 || c->q.cs > 3)
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "Error! Unrecognized color transformation");
    const size_t n = (size_t)w * h;
    if ((n & 1) || !is_aligned(frames, 8) || (frame_stride & 1))
        return fail(c, LUMAHIP_ERR_ARG, "transform needs an even pixel count and 8-byte aligned frames");
    HIPCHK(c, hipSetDevice(c->device));
    XfArgs a{};
    a.buf = frames;
    a.frame_stride = frame_stride;
    a.chan_stride = n;
    a.n2 = n / 2;
    a.nframes = (int)nframes;
    a.sc = sc;
    a.Lmax = c->q.Lmax;
    xf_kernel_t kern = pick_xf(c->q.cs, toCs != 0);
    size_t total = a.n2 * nframes;
    long grid = (long)((total + 255) / 256);
    const long cap = (long)c->num_cu * 8;
    if (grid > cap)
        grid = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}

extern "C" int lumahip_synth_frames_device(lumahip_ctx *c, float *dst, size_t frame_stride, unsigned nframes, unsigned w,
                                           unsigned h, uint64_t seed, uint64_t first_frame)
{
    if (!c || !dst || nframes == 0 || w == 0 || h == 0)
        return fail(c, LUMAHIP_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n3 = (size_t)3 * w * h;
    size_t total = n3 * nframes;
    long grid = (long)((total + 255) / 256);
    const long cap = (long)c->num_cu * 16;
    if (grid > cap)
        grid = cap;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)grid), dim3(256), 0, c->stream, dst, frame_stride, (int)nframes, n3, seed,
                       first_frame);
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}

extern "C" int lumahip_probe_encode_traffic_device(lumahip_ctx *c, const float *rgb, size_t frame_stride, unsigned nframes,
                                                   unsigned w, unsigned h, unsigned char *const planes[3],
                                                   const int stride[3], const size_t pfs[3], int iters, float *avg_ms)
{
    if (!c || !rgb || !planes || !stride || !pfs || nframes == 0 || iters <= 0 || !avg_ms)
        return fail(c, LUMAHIP_ERR_ARG, "bad argument");
    if (w == 0 || h == 0 || (w % 4) || (h & 1) || !is_aligned(rgb, 16) || (frame_stride % 4))
        return fail(c, LUMAHIP_ERR_ARG, "the traffic probe needs w % 4 == 0, even h and 16-byte aligned frames");
    for (int p = 0; p < 3; p++)
        if (!planes[p] || !is_aligned(planes[p], 8) || (stride[p] % (p ? 4 : 8)) || (pfs[p] % 8))
            return fail(c, LUMAHIP_ERR_ARG, "the traffic probe needs 8-byte aligned 16-bit 4:2:0 planes");
    HIPCHK(c, hipSetDevice(c->device));
    EncArgs a{};
    const int threads = 256;
    if (!make_geom(a.g, w, h, 4, threads / 64, nframes))
        return fail(c, LUMAHIP_ERR_ARG, "batch too large");
    a.src = rgb;
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

extern "C" int lumahip_time_launches(lumahip_ctx *c, int dir, int iters, const float *rgb, size_t frame_stride,
                                     unsigned nframes, unsigned w, unsigned h, float sc, int profile,
                                     unsigned char *const planes[3], const int stride[3], const size_t pfs[3],
                                     float *avg_ms)
{
    if (!c || iters <= 0 || !avg_ms)
        return fail(c, LUMAHIP_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    EventPair ev;
    HIPCHK(c, ev.create());
    int rc = LUMAHIP_OK;
    HIPCHK(c, hipEventRecord(ev.e0, c->stream));
    for (int i = 0; i < iters && rc == LUMAHIP_OK; i++) {
        if (dir == 0)
            rc = lumahip_encode_frames_device(c, rgb, frame_stride, nframes, w, h, sc, profile, planes, stride, pfs, nullptr);
        else
            rc = lumahip_decode_frames_device(c, (const unsigned char *const *)planes, stride, pfs, nframes, w, h, profile,
                                              sc, const_cast<float *>(rgb), frame_stride);
    }
    HIPCHK(c, hipEventRecord(ev.e1, c->stream));
    HIPCHK(c, hipEventSynchronize(ev.e1));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *avg_ms = ms / iters;
    return rc;
}

// ---------------------------------------------------------------------------------------- host entry points

// ---- host <-> device transfers of the _host entry points --------------------------------------------------------
// Caller memory is pageable unless the caller pinned it (hipHostMalloc, hipHostRegister / lumahip_host_register).
// Pinned memory is handed to the copy engine directly (asynchronous, the fast path of the batched entry points).
// Pageable memory is NOT handed to hipMemcpy*Async: the runtime then pins the caller's pages on the fly and caches
// that pinning, and on this stack (ROCm 7.2, MI355X) the GPU occasionally faulted on such a range when host buffers are
// allocated and freed at a high rate ("Memory access fault by GPU ... on address <host heap page>", about one run of
// the GPU test suite in twenty).  Pageable data therefore moves through two context-owned pinned chunks per direction:
// the CPU copy of chunk k+1 overlaps the DMA of chunk k.
static constexpr size_t XFER_CHUNK = (size_t)8 << 20;

static bool host_range_is_pinned(const void *p, size_t bytes)
{
    if (!p || !bytes)
        return false;
    const unsigned char *ends[2] = {(const unsigned char *)p, (const unsigned char *)p + bytes - 1};
    for (const unsigned char *q : ends) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();  // plain malloc memory: not an error of ours
            return false;
        }
        if (at.type != hipMemoryTypeHost)
            return false;
    }
    return true;
}

static int stage_ready(lumahip_ctx *c, lumahip_ctx::Stage &st)
{
    if (!st.h) {
        HIPCHK(c, hipHostMalloc((void **)&st.h, XFER_CHUNK, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    }
    if (st.pending) {
        HIPCHK(c, hipEventSynchronize(st.ev));
        st.pending = false;
    }
    return LUMAHIP_OK;
}

// rows x width bytes, host pitch hp, device pitch dp.  Returns once the copies are queued on `s` (the caller's buffer
// is no longer needed if it was pageable: it has been copied into the staging chunks).
static int xfer_h2d_2d(lumahip_ctx *c, void *dst, size_t dp, const void *src, size_t hp, size_t width, size_t rows, hipStream_t s)
{
    if (!width || !rows)
        return LUMAHIP_OK;
    if (host_range_is_pinned(src, (rows - 1) * hp + width)) {
        if (dp == width && hp == width)
            HIPCHK(c, hipMemcpyAsync(dst, src, width * rows, hipMemcpyHostToDevice, s));
        else
            HIPCHK(c, hipMemcpy2DAsync(dst, dp, src, hp, width, rows, hipMemcpyHostToDevice, s));
        return LUMAHIP_OK;
    }
    // staged: the device side is written as whole rows of dp bytes (the padding between rows belongs to the context's
    // own buffers), so that one chunk is one contiguous DMA
    const bool flat = (dp == width && hp == width);
    if (!flat && dp > XFER_CHUNK)
        return fail(c, LUMAHIP_ERR_ARG, "row pitch %zu exceeds the staging chunk", dp);
    const size_t total = flat ? width * rows : rows;                 // bytes or rows
    const size_t per = flat ? XFER_CHUNK : XFER_CHUNK / dp;          // per chunk
    int k = 0;
    for (size_t done = 0; done < total; k++) {
        lumahip_ctx::Stage &st = c->stage_up[k & 1];
        int rc = stage_ready(c, st);
        if (rc)
            return rc;
        const size_t n = total - done < per ? total - done : per;
        size_t bytes;
        if (flat) {
            memcpy(st.h, (const unsigned char *)src + done, n);
            bytes = n;
        } else {
            for (size_t r = 0; r < n; r++)
                memcpy(st.h + r * dp, (const unsigned char *)src + (done + r) * hp, width);
            bytes = (n - 1) * dp + width;
        }
        HIPCHK(c, hipMemcpyAsync((unsigned char *)dst + done * (flat ? 1 : dp), st.h, bytes, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(st.ev, s));
        st.pending = true;
        done += n;
    }
    return LUMAHIP_OK;
}

static int xfer_h2d(lumahip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    return xfer_h2d_2d(c, dst, bytes, src, bytes, bytes, 1, s);
}

// Device -> host.  Pinned destination: queued on `s`, the caller synchronises.  Pageable destination: the data is in
// `dst` when the call returns (everything queued on `s` before it has completed by then).
static int xfer_d2h_2d(lumahip_ctx *c, void *dst, size_t hp, const void *src, size_t dp, size_t width, size_t rows, hipStream_t s)
{
    if (!width || !rows)
        return LUMAHIP_OK;
    if (host_range_is_pinned(dst, (rows - 1) * hp + width)) {
        if (dp == width && hp == width)
            HIPCHK(c, hipMemcpyAsync(dst, src, width * rows, hipMemcpyDeviceToHost, s));
        else
            HIPCHK(c, hipMemcpy2DAsync(dst, hp, src, dp, width, rows, hipMemcpyDeviceToHost, s));
        return LUMAHIP_OK;
    }
    const bool flat = (dp == width && hp == width);
    if (!flat && dp > XFER_CHUNK)
        return fail(c, LUMAHIP_ERR_ARG, "row pitch %zu exceeds the staging chunk", dp);
    const size_t total = flat ? width * rows : rows;
    const size_t per = flat ? XFER_CHUNK : XFER_CHUNK / dp;
    size_t prev_done = 0, prev_n = 0;
    int k = 0;
    auto drain = [&](lumahip_ctx::Stage &st, size_t at, size_t n) -> int {
        HIPCHK(c, hipEventSynchronize(st.ev));
        st.pending = false;
        if (flat) {
            memcpy((unsigned char *)dst + at, st.h, n);
        } else {
            for (size_t r = 0; r < n; r++)
                memcpy((unsigned char *)dst + (at + r) * hp, st.h + r * dp, width);
        }
        return LUMAHIP_OK;
    };
    for (size_t done = 0; done < total; k++) {
        lumahip_ctx::Stage &st = c->stage_dn[k & 1];
        int rc = stage_ready(c, st);
        if (rc)
            return rc;
        const size_t n = total - done < per ? total - done : per;
        const size_t bytes = flat ? n : (n - 1) * dp + width;
        HIPCHK(c, hipMemcpyAsync(st.h, (const unsigned char *)src + done * (flat ? 1 : dp), bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(st.ev, s));
        st.pending = true;
        if (k > 0 && (rc = drain(c->stage_dn[(k - 1) & 1], prev_done, prev_n)))
            return rc;
        prev_done = done;
        prev_n = n;
        done += n;
    }
    return drain(c->stage_dn[(k - 1) & 1], prev_done, prev_n);
}

static int xfer_d2h(lumahip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    return xfer_d2h_2d(c, dst, bytes, src, bytes, bytes, 1, s);
}

// a few floats from the device: through the pinned scratch, synchronous
static int read_small(lumahip_ctx *c, float *dst, const float *src_dev, int n, hipStream_t s)
{
    if (!c->h_small)
        HIPCHK(c, hipHostMalloc((void **)&c->h_small, 64 * sizeof(float), hipHostMallocDefault));
    HIPCHK(c, hipMemcpyAsync(c->h_small, src_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    memcpy(dst, c->h_small, (size_t)n * sizeof(float));
    return LUMAHIP_OK;
}

static int ensure(lumahip_ctx *c, void **p, size_t *cap, size_t need)
{
    if (*cap >= need)
        return LUMAHIP_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIPCHK(c, hipMalloc(p, need));
    *cap = need;
    return LUMAHIP_OK;
}

struct PlaneLayout {
    int rows[3];
    int row_bytes[3];
    size_t off[3];
    size_t total;
};

static void plane_layout(PlaneLayout &L, unsigned w, unsigned h, int profile, const int stride[3])
{
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    size_t off = 0;
    for (int p = 0; p < 3; p++) {
        const int pw = (p && sub) ? (int)(w + 1) / 2 : (int)w;
        const int ph = (p && sub) ? (int)(h + 1) / 2 : (int)h;
        L.rows[p] = ph;
        L.row_bytes[p] = pw * bps;
        L.off[p] = off;
        off += ((size_t)ph * stride[p] + 255) & ~(size_t)255;
    }
    L.total = off;
}

// sequential fp32 sum of n device floats / (w*h), as the reference forms its mean luminance (see k_seq_sum)
static int seq_mean(lumahip_ctx *c, const float *chan0_dev, unsigned w, unsigned h, float *mean_host)
{
    const size_t n = (size_t)w * h;
    if (!c->d_stats)
        HIPCHK(c, hipMalloc(&c->d_stats, 3 * sizeof(float)));
    hipLaunchKernelGGL(k_seq_sum, dim3(1), dim3(64), 0, c->stream, chan0_dev, n, c->d_stats);
    HIPCHK(c, hipGetLastError());
    float sum = 0.0f;
    int rc = read_small(c, &sum, c->d_stats, 1, c->stream);
    if (rc)
        return rc;
    *mean_host = sum / (float)((int)w * (int)h);  // avg /= (w*h), src/luma_encoder.cpp:314
    return LUMAHIP_OK;
}

// mean of transformed channel 0 of ONE device-resident (untransformed) frame, summed exactly as the reference does
static int mean_luminance_reference_impl(lumahip_ctx *c, const float *rgb_dev, unsigned w, unsigned h, float sc, int cs_eff,
                                         float *mean_host)
{
    const size_t n = (size_t)w * h;
    int rc = ensure(c, (void **)&c->d_arr, &c->d_arr_cap, n * sizeof(float));
    if (rc)
        return rc;
    void (*kern)(const float *, size_t, size_t, float, float, float *) = nullptr;
    switch (cs_eff) {
    case CS_LUV: kern = k_channel0<CS_LUV>; break;
    case CS_RGB: kern = k_channel0<CS_RGB>; break;
    case CS_YCBCR: kern = k_channel0<CS_YCBCR>; break;
    case CS_XYZ: kern = k_channel0<CS_XYZ>; break;
    case CS_PACK: kern = k_channel0<CS_PACK>; break;
    }
    if (!kern)
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "Unrecognized color transformation (colour space %d)", cs_eff);
    long grid = (long)((n + 255) / 256);
    if (grid > (long)c->num_cu * 8)
        grid = (long)c->num_cu * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, c->stream, rgb_dev, n, n, sc, c->q.Lmax, c->d_arr);
    return seq_mean(c, c->d_arr, w, h, mean_host);
}

extern "C" int lumahip_mean_luminance_reference_device(lumahip_ctx *c, const float *rgb_dev, unsigned w, unsigned h, float sc,
                                                       float *mean_host)
{
    if (!c || !rgb_dev || !mean_host || w == 0 || h == 0)
        return fail(c, LUMAHIP_ERR_ARG, "bad argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    HIPCHK(c, hipSetDevice(c->device));
    return mean_luminance_reference_impl(c, rgb_dev, w, h, sc, c->q.cs, mean_host);
}

// The reference warns when its (sequentially summed) mean luminance is <= 1.  That fp32 sum is far from the true sum on
// large frames: once the running sum S is large, addends below ulp(S)/2 vanish and the rest are rounded to multiples of
// ulp(S) (measured: -0.2 % at 1080p, several % at 4K on wide-range content), whereas the kernels' statistic (per-wave
// partial sums) is accurate to ~1e-6.  Around the threshold S stays below N * 4, i.e. ulp(S)/2 <= 2 up to 8K frames, so
// the two can only disagree about `<= 1` when the accurate mean lies in [0.25, 4]: inside that band the host entry points
// replace the statistic by the reference's exact value (k_seq_sum), outside it the decision is the same either way.
static bool mean_near_threshold(float m) { return m >= 0.25f && m <= 4.0f; }

static int encode_frame_host_impl(lumahip_ctx *c, const float *rgb, unsigned w, unsigned h, float sc, int profile,
                                  unsigned char *const planes[3], const int stride[3], float *mean_lum,
                                  float *transformed_out, int cs_eff)
{
    if (!c || !rgb || !planes || !stride)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const int bps = profile > 1 ? 2 : 1;
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (int p = 0; p < 3; p++)
        if (!planes[p] || stride[p] < L.row_bytes[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: null or stride %d < row bytes %d", p, stride[p], L.row_bytes[p]);
    (void)bps;
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = ensure(c, (void **)&c->d_frame, &c->d_frame_cap, nfl * sizeof(float))))
        return rc;
    if ((rc = ensure(c, (void **)&c->d_planes, &c->d_planes_cap, L.total)))
        return rc;
    if (!c->d_stats)
        HIPCHK(c, hipMalloc(&c->d_stats, 3 * sizeof(float)));
    if ((rc = xfer_h2d(c, c->d_frame, rgb, nfl * sizeof(float), c->stream)))
        return rc;
    unsigned char *dp[3] = {c->d_planes + L.off[0], c->d_planes + L.off[1], c->d_planes + L.off[2]};
    const size_t pfs[3] = {0, 0, 0};
    rc = encode_frames_device_impl(c, c->d_frame, nfl, 1, w, h, sc, profile, dp, stride, pfs, c->d_stats, cs_eff);
    if (rc)
        return rc;
    for (int p = 0; p < 3; p++)
        if ((rc = xfer_d2h_2d(c, planes[p], stride[p], dp[p], stride[p], L.row_bytes[p], L.rows[p], c->stream)))
            return rc;
    if (transformed_out) {
        rc = lumahip_transform_color_space_device(c, c->d_frame, nfl, 1, w, h, 1, sc);
        if (rc)
            return rc;
        if ((rc = xfer_d2h(c, transformed_out, c->d_frame, nfl * sizeof(float), c->stream)))
            return rc;
    }
    float st[3] = {0, 0, 0};
    if ((rc = read_small(c, st, c->d_stats, 3, c->stream)))  // synchronises the stream
        return rc;
    if (mean_lum) {
        *mean_lum = st[0] / (float)((int)w * (int)h);  // avg /= (w*h), src/luma_encoder.cpp:314
        if (mean_near_threshold(*mean_lum))  // d_frame holds the caller's frame, or already its transformed version
            return transformed_out ? seq_mean(c, c->d_frame, w, h, mean_lum)
                                   : mean_luminance_reference_impl(c, c->d_frame, w, h, sc, cs_eff, mean_lum);
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_encode_frame_host(lumahip_ctx *c, const float *rgb, unsigned w, unsigned h, float sc, int profile,
                                         unsigned char *const planes[3], const int stride[3], float *mean_lum,
                                         float *transformed_out)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    return encode_frame_host_impl(c, rgb, w, h, sc, profile, planes, stride, mean_lum, transformed_out, c->q.cs);
}

static int decode_frame_host_impl(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3], unsigned w,
                                  unsigned h, int profile, float sc, float *rgb_out, int cs_eff)
{
    if (!c || !rgb_out || !planes || !stride)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (int p = 0; p < 3; p++)
        if (!planes[p] || stride[p] < L.row_bytes[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: null or stride %d < row bytes %d", p, stride[p], L.row_bytes[p]);
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = ensure(c, (void **)&c->d_frame, &c->d_frame_cap, nfl * sizeof(float))))
        return rc;
    if ((rc = ensure(c, (void **)&c->d_planes, &c->d_planes_cap, L.total)))
        return rc;
    unsigned char *dp[3] = {c->d_planes + L.off[0], c->d_planes + L.off[1], c->d_planes + L.off[2]};
    for (int p = 0; p < 3; p++)
        if ((rc = xfer_h2d_2d(c, dp[p], stride[p], planes[p], stride[p], L.row_bytes[p], L.rows[p], c->stream)))
            return rc;
    const size_t pfs[3] = {0, 0, 0};
    rc = decode_impl(c, dp, stride, pfs, 1, w, h, profile, sc, c->d_frame, nfl, DisplayParams(), cs_eff);
    if (rc)
        return rc;
    if ((rc = xfer_d2h(c, rgb_out, c->d_frame, nfl * sizeof(float), c->stream)))
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

extern "C" int lumahip_decode_frame_host(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                         unsigned w, unsigned h, int profile, float sc, float *rgb_out)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    return decode_frame_host_impl(c, planes, stride, w, h, profile, sc, rgb_out, c->q.cs);
}

// ---- batched host entry points: a 3-slot software pipeline over three streams.  Frame i's H2D copy runs while
// frame i-1's kernel and frame i-2's D2H copies are in flight; with pinned caller memory (lumahip_host_register) the
// two copy directions overlap as well and the rate approaches the PCIe H2D rate.
static int pipe_prepare(lumahip_ctx *c, size_t frame_bytes, size_t planes_bytes, unsigned nframes)
{
    if (!c->s_h2d) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_kern, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking));
        for (auto &sl : c->slot) {
            HIPCHK(c, hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&sl.kern, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&sl.d2h, hipEventDisableTiming));
            HIPCHK(c, hipMalloc(&sl.d_stats, 3 * sizeof(float)));
        }
    }
    if (c->slot_frame_cap < frame_bytes || c->slot_planes_cap < planes_bytes) {
        HIPCHK(c, hipDeviceSynchronize());
        for (auto &sl : c->slot) {
            (void)hipFree(sl.d_frame);
            (void)hipFree(sl.d_planes);
            sl.d_frame = nullptr;
            sl.d_planes = nullptr;
            HIPCHK(c, hipMalloc(&sl.d_frame, frame_bytes));
            HIPCHK(c, hipMalloc(&sl.d_planes, planes_bytes));
        }
        c->slot_frame_cap = frame_bytes;
        c->slot_planes_cap = planes_bytes;
    }
    if (c->h_stats_cap < nframes) {
        if (c->h_stats)
            (void)hipHostFree(c->h_stats);
        c->h_stats = nullptr;
        HIPCHK(c, hipHostMalloc(&c->h_stats, (size_t)nframes * 3 * sizeof(float), hipHostMallocDefault));
        c->h_stats_cap = nframes;
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_encode_frames_host(lumahip_ctx *c, const float *const *rgb, unsigned nframes, unsigned w, unsigned h,
                                          float sc, int profile, unsigned char *const *planes, const int stride[3],
                                          float *mean_lum)
{
    if (!c || !rgb || !planes || !stride || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, c->q.cs);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (unsigned i = 0; i < nframes; i++) {
        if (!rgb[i])
            return fail(c, LUMAHIP_ERR_ARG, "null frame %u", i);
        for (int p = 0; p < 3; p++)
            if (!planes[3 * i + p] || stride[p] < L.row_bytes[p])
                return fail(c, LUMAHIP_ERR_ARG, "frame %u plane %d: null or stride too small", i, p);
    }
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = pipe_prepare(c, nfl * sizeof(float), L.total, nframes)))
        return rc;
    hipStream_t saved = c->stream;
    const size_t pfs[3] = {0, 0, 0};
    // Frame i's upload and kernel are queued BEFORE frame i-1's planes are fetched: with pageable planes the fetch
    // blocks the host (xfer_d2h_2d), and this order keeps the GPU busy with frame i meanwhile.
    auto fetch = [&](unsigned i) -> int {
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
        (void)hipStreamWaitEvent(c->s_d2h, sl.kern, 0);
        int r = LUMAHIP_OK;
        for (int p = 0; p < 3 && r == LUMAHIP_OK; p++)
            r = xfer_d2h_2d(c, planes[3 * i + p], stride[p], dp[p], stride[p], L.row_bytes[p], L.rows[p], c->s_d2h);
        if (r)
            return r;
        (void)hipMemcpyAsync(c->h_stats + 3 * (size_t)i, sl.d_stats, 3 * sizeof(float), hipMemcpyDeviceToHost, c->s_d2h);
        (void)hipEventRecord(sl.d2h, c->s_d2h);
        return LUMAHIP_OK;
    };
    for (unsigned i = 0; i < nframes && rc == LUMAHIP_OK; i++) {
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
        if (i >= 3) {
            // slot reuse: the kernel of frame i-3 must have consumed d_frame, its D2H must have drained d_planes
            (void)hipStreamWaitEvent(c->s_h2d, sl.kern, 0);
            (void)hipStreamWaitEvent(c->s_kern, sl.d2h, 0);
        }
        if ((rc = xfer_h2d(c, sl.d_frame, rgb[i], nfl * sizeof(float), c->s_h2d)))
            break;
        (void)hipEventRecord(sl.h2d, c->s_h2d);
        (void)hipStreamWaitEvent(c->s_kern, sl.h2d, 0);
        c->stream = c->s_kern;
        rc = lumahip_encode_frames_device(c, sl.d_frame, nfl, 1, w, h, sc, profile, dp, stride, pfs, sl.d_stats);
        c->stream = saved;
        if (rc)
            break;
        (void)hipEventRecord(sl.kern, c->s_kern);
        if (i >= 1)
            rc = fetch(i - 1);
    }
    if (rc == LUMAHIP_OK)
        rc = fetch(nframes - 1);
    c->stream = saved;
    HIPCHK(c, hipStreamSynchronize(c->s_h2d));
    HIPCHK(c, hipStreamSynchronize(c->s_kern));
    HIPCHK(c, hipStreamSynchronize(c->s_d2h));
    if (rc == LUMAHIP_OK && mean_lum)
        for (unsigned i = 0; i < nframes && rc == LUMAHIP_OK; i++) {
            mean_lum[i] = c->h_stats[3 * (size_t)i] / (float)((int)w * (int)h);
            if (mean_near_threshold(mean_lum[i])) {  // rare: redo this frame's sum in the reference's order
                if ((rc = xfer_h2d(c, c->slot[0].d_frame, rgb[i], nfl * sizeof(float), c->stream)))
                    return rc;
                rc = mean_luminance_reference_impl(c, c->slot[0].d_frame, w, h, sc, c->q.cs, &mean_lum[i]);
            }
        }
    return rc;
}

extern "C" int lumahip_decode_frames_host(lumahip_ctx *c, const unsigned char *const *planes, const int stride[3],
                                          unsigned nframes, unsigned w, unsigned h, int profile, float sc,
                                          float *const *rgb_out)
{
    if (!c || !rgb_out || !planes || !stride || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, c->q.cs);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (unsigned i = 0; i < nframes; i++) {
        if (!rgb_out[i])
            return fail(c, LUMAHIP_ERR_ARG, "null output frame %u", i);
        for (int p = 0; p < 3; p++)
            if (!planes[3 * i + p] || stride[p] < L.row_bytes[p])
                return fail(c, LUMAHIP_ERR_ARG, "frame %u plane %d: null or stride too small", i, p);
    }
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = pipe_prepare(c, nfl * sizeof(float), L.total, nframes)))
        return rc;
    hipStream_t saved = c->stream;
    const size_t pfs[3] = {0, 0, 0};
    auto fetch = [&](unsigned i) -> int {  // as in lumahip_encode_frames_host: frame i-1 is fetched after frame i is queued
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        (void)hipStreamWaitEvent(c->s_d2h, sl.kern, 0);
        int r = xfer_d2h(c, rgb_out[i], sl.d_frame, nfl * sizeof(float), c->s_d2h);
        if (r)
            return r;
        (void)hipEventRecord(sl.d2h, c->s_d2h);
        return LUMAHIP_OK;
    };
    for (unsigned i = 0; i < nframes && rc == LUMAHIP_OK; i++) {
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
        if (i >= 3) {
            (void)hipStreamWaitEvent(c->s_h2d, sl.kern, 0);   // planes of frame i-3 consumed
            (void)hipStreamWaitEvent(c->s_kern, sl.d2h, 0);   // floats of frame i-3 copied out
        }
        for (int p = 0; p < 3 && rc == LUMAHIP_OK; p++)
            rc = xfer_h2d_2d(c, dp[p], stride[p], planes[3 * i + p], stride[p], L.row_bytes[p], L.rows[p], c->s_h2d);
        if (rc)
            break;
        (void)hipEventRecord(sl.h2d, c->s_h2d);
        (void)hipStreamWaitEvent(c->s_kern, sl.h2d, 0);
        c->stream = c->s_kern;
        rc = lumahip_decode_frames_device(c, dp, stride, pfs, 1, w, h, profile, sc, sl.d_frame, nfl);
        c->stream = saved;
        if (rc)
            break;
        (void)hipEventRecord(sl.kern, c->s_kern);
        if (i >= 1)
            rc = fetch(i - 1);
    }
    if (rc == LUMAHIP_OK)
        rc = fetch(nframes - 1);
    c->stream = saved;
    HIPCHK(c, hipStreamSynchronize(c->s_h2d));
    HIPCHK(c, hipStreamSynchronize(c->s_kern));
    HIPCHK(c, hipStreamSynchronize(c->s_d2h));
    return rc;
}

extern "C" int lumahip_transform_color_space_host(lumahip_ctx *c, float *frame, unsigned w, unsigned h, int toCs, float sc)
{
    if (!c || !frame)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    if (c->q.cs < 0 || c->q.cs > 3)
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "Error! Unrecognized color transformation");
    if (w == 0 || h == 0)
        return LUMAHIP_OK;  // the reference loops zero times and returns true
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)w * h, nfl = 3 * n;
    // device copy padded to an even pixel count per channel so the pair kernel applies to odd sizes too
    const size_t npad = (n + 1) & ~(size_t)1;
    int rc = ensure(c, (void **)&c->d_frame, &c->d_frame_cap, 3 * npad * sizeof(float));
    if (rc)
        return rc;
    if (npad == n) {
        if ((rc = xfer_h2d(c, c->d_frame, frame, nfl * sizeof(float), c->stream)))
            return rc;
    } else {
        HIPCHK(c, hipMemsetAsync(c->d_frame, 0, 3 * npad * sizeof(float), c->stream));
        for (int ch = 0; ch < 3; ch++)
            if ((rc = xfer_h2d(c, c->d_frame + ch * npad, frame + ch * n, n * sizeof(float), c->stream)))
                return rc;
    }
    // the kernel addresses channels at chan_stride = (w*h); present the padded buffer as a (npad x 1) frame
    rc = lumahip_transform_color_space_device(c, c->d_frame, 3 * npad, 1, (unsigned)npad, 1, toCs, sc);
    if (rc)
        return rc;
    if (npad == n) {
        if ((rc = xfer_d2h(c, frame, c->d_frame, nfl * sizeof(float), c->stream)))
            return rc;
    } else {
        for (int ch = 0; ch < 3; ch++)
            if ((rc = xfer_d2h(c, frame + ch * n, c->d_frame + ch * npad, n * sizeof(float), c->stream)))
                return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

static int array_launch(lumahip_ctx *c, const float *d_in, float *d_out, size_t n, unsigned ch, bool quant);

static int array_op(lumahip_ctx *c, const float *in, float *out, size_t n, unsigned ch, bool quant)
{
    if (!c || !in || !out)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    if (n == 0)
        return LUMAHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure(c, (void **)&c->d_arr, &c->d_arr_cap, 2 * n * sizeof(float));
    if (rc)
        return rc;
    if ((rc = xfer_h2d(c, c->d_arr, in, n * sizeof(float), c->stream)))
        return rc;
    if ((rc = array_launch(c, c->d_arr, c->d_arr + n, n, ch, quant)))
        return rc;
    if ((rc = xfer_d2h(c, out, c->d_arr + n, n * sizeof(float), c->stream)))
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
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

static int array_launch(lumahip_ctx *c, const float *d_in, float *d_out, size_t n, unsigned ch, bool quant)
{
    HIPCHK(c, hipSetDevice(c->device));
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

extern "C" int lumahip_quantize_array_host(lumahip_ctx *c, const float *in, float *out, size_t n, unsigned ch)
{
    return array_op(c, in, out, n, ch, true);
}

extern "C" int lumahip_dequantize_array_host(lumahip_ctx *c, const float *in, float *out, size_t n, unsigned ch)
{
    return array_op(c, in, out, n, ch, false);
}

// LumaEncoder::setChannels / LumaDecoder::getVpxChannels on their own: no colour transform.  Channel 0
// goes through the LUT; channels 1,2 through the LUT for RGB / XYZ (src/luma_quantizer.cpp:219,251) --
// which is the CS_RGB kernel with sc = 1 (x*1.0f and x/1.0f are exact) -- and through the colour quantizer
// otherwise (CS_PACK).
static int pack_cs(const lumahip_ctx *c) { return (c->q.cs == CS_RGB || c->q.cs == CS_XYZ) ? CS_RGB : CS_PACK; }

extern "C" int lumahip_pack_frame_host(lumahip_ctx *c, const float *transformed, unsigned w, unsigned h, int profile,
                                       unsigned char *const planes[3], const int stride[3], float *mean_lum)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    return encode_frame_host_impl(c, transformed, w, h, 1.0f, profile, planes, stride, mean_lum, nullptr, pack_cs(c));
}

extern "C" int lumahip_unpack_frame_host(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                         unsigned w, unsigned h, int profile, float *dequantized_out)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    return decode_frame_host_impl(c, planes, stride, w, h, profile, 1.0f, dequantized_out, pack_cs(c));
}

// ---------------------------------------------------------------------------------------- memory helpers

extern "C" int lumahip_powf_probe_device(lumahip_ctx *c, float *out_dev, uint32_t first_bits, size_t n, float y, int regular)
{
    if (!c || !out_dev || n == 0)
        return fail(c, LUMAHIP_ERR_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    long grid = (long)((n + 255) / 256);
    if (grid > (long)c->num_cu * 16)
        grid = (long)c->num_cu * 16;
    hipLaunchKernelGGL(k_powf_probe, dim3((unsigned)grid), dim3(256), 0, c->stream, out_dev, first_bits, n, y, regular);
    HIPCHK(c, hipGetLastError());
    return LUMAHIP_OK;
}

extern "C" int lumahip_quantize_probe_device(lumahip_ctx *c, uint16_t *out_dev, uint32_t first_bits, size_t n, int nonneg)
{
    if (!c || !out_dev || n == 0 || (n % 4) != 0 || !is_aligned(out_dev, 8))
        return fail(c, LUMAHIP_ERR_ARG, "bad argument (n must be a multiple of 4, out 8-byte aligned)");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t lds = lds_bytes(c, true, CS_PACK);
    void (*kern)(const QuantDev, uint16_t *, uint32_t, size_t) = nullptr;
    switch (c->q.mode) {
    case LUT_LITERAL_LDS: kern = nonneg ? k_quantize_probe<0, true> : k_quantize_probe<0, false>; break;
    case LUT_LITERAL_GLOBAL: kern = nonneg ? k_quantize_probe<2, true> : k_quantize_probe<2, false>; break;
    case LUT_THRESH_LDS: kern = nonneg ? k_quantize_probe<3, true> : k_quantize_probe<3, false>; break;
    case LUT_THRESH_GLOBAL: kern = nonneg ? k_quantize_probe<4, true> : k_quantize_probe<4, false>; break;
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

extern "C" int lumahip_host_register(lumahip_ctx *c, void *p, size_t bytes)
{
    if (!c || !p || !bytes)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostRegister(p, bytes, hipHostRegisterDefault));
    return LUMAHIP_OK;
}

extern "C" int lumahip_host_unregister(lumahip_ctx *c, void *p)
{
    if (!c || !p)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostUnregister(p));
    return LUMAHIP_OK;
}

extern "C" int lumahip_malloc(lumahip_ctx *c, void **p, size_t bytes)
{
    if (!c || !p)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(p, bytes));
    return LUMAHIP_OK;
}

extern "C" int lumahip_free(lumahip_ctx *c, void *p)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipFree(p));
    return LUMAHIP_OK;
}

extern "C" int lumahip_memcpy_h2d(lumahip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = xfer_h2d(c, dst, src, bytes, c->stream);
    if (rc)
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

extern "C" int lumahip_memcpy_d2h(lumahip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = xfer_d2h(c, dst, src, bytes, c->stream);
    if (rc)
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}
