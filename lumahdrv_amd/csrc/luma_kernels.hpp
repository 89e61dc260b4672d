// luma_kernels.hpp -- fused encode / decode kernels of the Luma HDRv quantize / dequantize path, gfx950.
//
// ENC = LumaQuantizer::transformColorSpace(frame,true,sc) (src/luma_quantizer.cpp:267-373) fused with
//       LumaEncoder::setChannels / setVpxChannel (src/luma_encoder.cpp:196-201,260-317);
// DEC = LumaDecoder::getVpxChannels (src/luma_decoder.cpp:205-240) fused with
//       LumaQuantizer::transformColorSpace(frame,false,sc) (src/luma_quantizer.cpp:374-479).
//
// Work decomposition (both directions): a *unit* is VW pixels x 2 rows (VW/2 chroma quads in 4:2:0); a
// thread owns one unit per tile; a wave owns 64 consecutive units of one row pair (64*VW*4 B = 1 KiB
// contiguous per load instruction per row and channel); a workgroup of NW waves owns a tile of
// 64*VW pixels x 2*NW rows; workgroups are persistent and stride over tiles (tile index is
// wave-uniform, so all index arithmetic on it is scalar).  The luminance search records (encode) /
// the transfer-function table (decode) are staged once per workgroup in LDS.  No inter-workgroup communication except the optional
// per-frame statistics (float atomics).
#pragma once

#include <type_traits>

#include "luma_device.hpp"


namespace lh {

// Minimum waves per SIMD the encode kernels are register-allocated for.  The light variants (Lu'v' / pack-only,
// 4:2:0 or VW=2) fit 80 VGPRs without spilling and run 6 waves per SIMD; the heavy ones (fp64 powf, three LUT
// searches per pixel, 4:4:4 with VW=4) would spill at that bound and keep the 4-wave / 128-VGPR budget.
template <int CS, bool SUB, int VW>
struct EncWaves {
    static constexpr int value = ((CS == CS_LUV || CS == CS_PACK) && (SUB || VW == 2)) ? 6 : 4;
};

struct FrameGeom {
    int w, h;            // luma size (even)
    int unitsX, unitsY;  // w/VW, h/2
    int tilesX, tilesY;  // ceil(unitsX/64), ceil(unitsY/NW)
    int tilesPerFrame;
    int totalTiles;      // tilesPerFrame * nframes
    int nframes;
    int interleave;      // 0: tiles in frame-major order; 1: tile t belongs to frame t % nframes (DecArgs::rot says why)
};

struct EncArgs {
    QuantDev q;
    FrameGeom g;
    const float *src[3];  // colour plane c of frame f at src[c] + f*frame_stride (the reference's LumaFrame,
                          // include/luma/luma_frame.h:84-87 there, is src[c] = base + c*w*h, frame_stride = 3*w*h)
    size_t frame_stride;  // floats
    unsigned char *dst[3];
    int stride[3];        // bytes
    size_t dst_frame_stride[3];
    float sc;
    int bps;              // bytes per sample: 1 or 2
    int aligned;          // 1: vector stores allowed
    float *stats;         // nullable: per frame STATS_SLOTS partial {sum,min,max} triples (k_fold_stats folds them)
    const float *half;    // LM == 6 only: the half-input table of this (sc, Lmax), HALF_TABLE_LEN floats padded to 16 B (luma_device.hpp half_lookup)
    // LM == 6 only, nullable: THIS launch's word in host-visible memory, which a workgroup sets to 1 when every unit of every
    // one of its waves held inputs that are not halves -- the stream is not binary16 data and the host stops picking this
    // kernel for a while (LagPolicy, lumahip_internal.hpp: the host reads the word after the launch's completion event).  Feedback only:
    // nothing a launch computes depends on it.
    uint32_t *half_flag;
};

struct DecArgs {
    QuantDev q;
    FrameGeom g;
    const unsigned char *src[3];
    int stride[3];
    size_t src_frame_stride[3];
    float *dst[3];        // colour plane c of frame f at dst[c] + f*frame_stride; dst[0] null when only the display output is wanted
    size_t frame_stride;
    // rot_on: PACKED frames (LumaFrame layout, channel c at base + c*w*h) spread over three buffers, frame f at
    // rot[f % 3] + (f / 3) * frame_stride -- with the three buffers in three HBM region groups and the tiles of a launch interleaved
    // over its frames (FrameGeom::interleave), the workgroups running at any moment write all three groups, which one launch
    // that writes a batch of packed frames into ONE buffer cannot (lumahip_decode_frames_device_rotating)
    float *rot[3];
    int rot_on;
    float sc;
    int bps;
    int aligned;
    // optional display epilogue (lumaplay's fragment shader, src/lumaplay_dequantizer.frag:145-156):
    // RGBA8, 4 bytes per pixel, rows disp_stride bytes apart
    unsigned char *disp;
    int disp_stride;
    size_t disp_frame_stride;
    float exposure, inv_gamma;
    int do_tmo, ldr_sim;
    // YCbCr decode with the per-stream red / blue tables (k_decode<..., RB>): rb[cb * lut_len + ycode] = blue,
    // rb[rb_plane + cr * lut_len + ycode] = red, final values (after / sc) of this call's preScaling; global memory (L2 / MALL)
    const float *rb;
    size_t rb_plane;
    // a wave takes the gathers for a unit only when most of its lanes' codes are close to each other and to their neighbour
    // lane's (rb_wave_local): luminance codes within rb_near_y, colour codes within rb_near_c; negative rb_near_y: always
    int rb_near_y, rb_near_c;
    // nullable: THIS launch's word in host-visible memory, set to 1 by every workgroup in which some wave took the tables for some
    // unit -- a launch that leaves it 0 found none of its pixels local, and the host sends the next launches to the kernels
    // without the test (lumahip_internal.hpp LagPolicy).  Feedback only.
    uint32_t *rb_flag;
};

// LDS layout: [powf tables (YCbCr only; FIRST, so that their addresses are immediates in the powf chains)]
// [lut: lut_len+pad floats, rounded to 16 B | records: nbuckets u32, rounded to 16 B]
// [YCbCr decode with per-stream tables: y table (as long as the lut), then the Cb and Cr chroma-term tables, maxC+1 floats each]
// [u'v' table: maxC+1 floats (Lu'v' decode only), see luv_chroma_uv | half-input table (YCbCr encode, LM == 6)].
// Which parts a kernel stages is a compile-time set (encode: records; decode: the table).
// STAGE_POWFN: the 768-byte powf tables instead of the wide ones (the half-input kernels: their LDS belongs to the half table).
enum : int { STAGE_LUT = 1, STAGE_REC = 4, STAGE_POWF = 8, STAGE_UV = 16, STAGE_YT = 32, STAGE_POWFN = 64, STAGE_HALF = 128, STAGE_CT = 256 };

// fill the LDS copy of the powf tables (pow_glibc.hpp); the caller synchronises
LH_DEV void stage_powf_tables(PowfTables *t)
{
    const double lt[16][2] = LH_POWF_LOG2_TAB;
    const uint64_t et[32] = LH_POWF_EXP2_TAB;
    const int tid = threadIdx.x;
    if (tid < 16) {
        t->log2_tab[tid][0] = lt[tid][0];
        t->log2_tab[tid][1] = lt[tid][1];
    }
    if (tid < 32)
        t->exp2_tab[tid] = et[tid];
}
LH_DEV void stage_powf_tables(PowfTablesWide *t)
{
    const double lt[16][2] = LH_POWF_LOG2_TAB;
    stage_powf_tables(static_cast<PowfTables *>(t));
    for (int e = threadIdx.x; e < 2048; e += blockDim.x)
        pw_wide_entry(e, lt, t->wide[e][0], t->wide[e][1]);
    for (int e = threadIdx.x; e < FOLD_A_LEN; e += blockDim.x)
        pw_fold_entry<0>(e, lt, t->foldA[e][0], t->foldA[e][1]);
    if (threadIdx.x < 16)
        pw_fold_entry<1>(threadIdx.x, lt, t->foldC[threadIdx.x][0], t->foldC[threadIdx.x][1]);
}

LH_DEV int lds_lut_bytes(const QuantDev &q) { return ((q.lut_len + q.pad) * 4 + 15) & ~15; }
LH_DEV int lds_rec_bytes(const QuantDev &q) { return (q.nbuckets * (q.mode == 7 ? 8 : 4) + 15) & ~15; }   // (7: {T, start} pairs)

// offset of the search table / records behind the powf tables
template <int WHAT>
constexpr int lds_table_offset()
{
    return (WHAT & STAGE_POWF) ? (int)sizeof(PowfTablesWide) : (WHAT & STAGE_POWFN) ? (int)sizeof(PowfTables) : 0;
}

constexpr int lds_half_bytes() { return (HALF_TABLE_LEN * 4 + 15) & ~15; }

template <int WHAT>
LH_DEV void stage_tables(unsigned char *smem, const QuantDev &q, const float *half = nullptr)
{
    static_assert(sizeof(PowfTablesWide) % 16 == 0 && sizeof(PowfTables) % 16 == 0, "the tables behind the powf tables must stay 16-byte aligned");
    static_assert(!((WHAT & STAGE_LUT) && (WHAT & STAGE_REC)), "one search table per kernel");
    static_assert(!((WHAT & STAGE_POWF) && (WHAT & STAGE_POWFN)), "one set of powf tables per kernel");
    const int tid = threadIdx.x, nt = blockDim.x;
    constexpr int off = lds_table_offset<WHAT>();
    if constexpr (WHAT & STAGE_POWF)
        stage_powf_tables(reinterpret_cast<PowfTablesWide *>(smem));
    if constexpr (WHAT & STAGE_POWFN)
        stage_powf_tables(reinterpret_cast<PowfTables *>(smem));
    if constexpr (WHAT & STAGE_HALF) {
        static_assert((WHAT & STAGE_REC) && !(WHAT & (STAGE_UV | STAGE_YT)), "the half-input table sits behind the records");
        const float4 *g = reinterpret_cast<const float4 *>(half);
        float4 *s4 = reinterpret_cast<float4 *>(smem + off + lds_rec_bytes(q));
        for (int i = tid; i < lds_half_bytes() / 16; i += nt)
            s4[i] = g[i];
    }
    if constexpr (WHAT & STAGE_LUT) {
        // table length + pad is a multiple of 4 floats on the host side (buffer is padded to 16 B)
        const int n4 = lds_lut_bytes(q) / 16;
        const float4 *g = reinterpret_cast<const float4 *>(q.lut);
        float4 *s = reinterpret_cast<float4 *>(smem + off);
        for (int i = tid; i < n4; i += nt)
            s[i] = g[i];
    }
    if constexpr (WHAT & STAGE_REC) {
        const int b4 = lds_rec_bytes(q) / 16;  // the device buffer is padded to 16 B
        const uint4 *gb = reinterpret_cast<const uint4 *>(q.rec);
        uint4 *sb = reinterpret_cast<uint4 *>(smem + off);
        for (int i = tid; i < b4; i += nt)
            sb[i] = gb[i];
    }
    if constexpr (WHAT & STAGE_YT) {
        static_assert((WHAT & STAGE_LUT) && !(WHAT & STAGE_UV), "the y table of the YCbCr decode kernels sits behind the luminance table");
        const int n4 = lds_lut_bytes(q) / 16;  // same length and padding as the luminance table
        const float4 *g = reinterpret_cast<const float4 *>(q.ytab);
        float4 *s = reinterpret_cast<float4 *>(smem + off + lds_lut_bytes(q));
        for (int i = tid; i < n4; i += nt)
            s[i] = g[i];
    }
    if constexpr (WHAT & STAGE_CT) {
        // YCbCr decode: the chroma terms of every colour code (luma_device.hpp ycbcr_chroma_term), Cb table then Cr table,
        // behind the luminance table and the y table
        static_assert((WHAT & STAGE_YT), "the chroma-term tables sit behind the y table");
        float *ct = reinterpret_cast<float *>(smem + off + 2 * lds_lut_bytes(q));
        const int n = (int)q.maxC + 1;
        for (int i = tid; i < n; i += nt) {
            ct[i] = ycbcr_chroma_term(i, q.maxC, 1.8814f);
            ct[n + i] = ycbcr_chroma_term(i, q.maxC, 1.4746f);
        }
    }
    if constexpr (WHAT & STAGE_UV) {
        static_assert(WHAT & STAGE_LUT, "the u'v' table sits behind the luminance table");
        float *uv = reinterpret_cast<float *>(smem + off + lds_lut_bytes(q));
        const int n = (int)q.maxC + 1;
        for (int i = tid; i < n; i += nt)
            uv[i] = uv_table_entry(i, q.maxC);
    }
    __syncthreads();
}

// ---- sample stores / loads ------------------------------------------------------------------------
// Every frame byte is touched exactly once, so all global accesses of the pixel stream are non-temporal
// (`nt`): measured +4-5 % on the encode access pattern (profiles/r01_membench.txt).
typedef float lh_v4f __attribute__((ext_vector_type(4)));
typedef float lh_v2f __attribute__((ext_vector_type(2)));
typedef unsigned lh_v2u __attribute__((ext_vector_type(2)));

LH_DEV void nt_store_u32x2(void *p, uint32_t a, uint32_t b)
{
    lh_v2u t = {a, b};
    __builtin_nontemporal_store(t, reinterpret_cast<lh_v2u *>(p));
}
LH_DEV void nt_store_u32(void *p, uint32_t a) { __builtin_nontemporal_store(a, reinterpret_cast<uint32_t *>(p)); }
LH_DEV void nt_store_u16(void *p, uint16_t a) { __builtin_nontemporal_store(a, reinterpret_cast<uint16_t *>(p)); }
LH_DEV void nt_store_u8(void *p, unsigned char a) { __builtin_nontemporal_store(a, reinterpret_cast<unsigned char *>(p)); }

// N consecutive samples (codes) at p; bps 1 or 2; little-endian 16-bit exactly as
// src/luma_encoder.cpp:301-307 produces (bl = res/256 at +1, bh = res - bl*256 at +0); the 8-bit path
// keeps the low byte (the reference's float->unsigned char conversion, quirk 2).
template <int N>
LH_DEV void store_samples(unsigned char *p, const int (&c)[N], int bps, int aligned)
{
    if (bps == 2) {
        if (aligned) {
            if constexpr (N == 4) {
                nt_store_u32x2(p, (uint32_t)(c[0] & 0xffff) | ((uint32_t)c[1] << 16),
                               (uint32_t)(c[2] & 0xffff) | ((uint32_t)c[3] << 16));
            } else if constexpr (N == 2) {
                nt_store_u32(p, (uint32_t)(c[0] & 0xffff) | ((uint32_t)c[1] << 16));
            } else {
                nt_store_u16(p, (uint16_t)c[0]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; i++) {
                p[2 * i] = (unsigned char)(c[i] & 0xff);
                p[2 * i + 1] = (unsigned char)((c[i] >> 8) & 0xff);
            }
        }
    } else {
        if (aligned) {
            if constexpr (N == 4) {
                nt_store_u32(p, (uint32_t)(c[0] & 0xff) | ((uint32_t)(c[1] & 0xff) << 8) |
                                    ((uint32_t)(c[2] & 0xff) << 16) | ((uint32_t)(c[3] & 0xff) << 24));
            } else if constexpr (N == 2) {
                nt_store_u16(p, (uint16_t)((c[0] & 0xff) | ((c[1] & 0xff) << 8)));
            } else {
                nt_store_u8(p, (unsigned char)(c[0] & 0xff));
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; i++)
                p[i] = (unsigned char)(c[i] & 0xff);
        }
    }
}

// src/luma_decoder.cpp:222-225: buf[2x+1]*256.0f + buf[2x] (exact in fp32) or buf[x]
template <int N>
LH_DEV void load_samples(const unsigned char *p, int (&c)[N], int bps, int aligned)
{
    if (bps == 2) {
        if (aligned) {
            if constexpr (N == 4) {
                const lh_v2u v = __builtin_nontemporal_load(reinterpret_cast<const lh_v2u *>(p));
                c[0] = v.x & 0xffff; c[1] = v.x >> 16; c[2] = v.y & 0xffff; c[3] = v.y >> 16;
            } else if constexpr (N == 2) {
                const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
                c[0] = v & 0xffff; c[1] = v >> 16;
            } else {
                c[0] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(p));
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; i++)
                c[i] = (int)p[2 * i] | ((int)p[2 * i + 1] << 8);
        }
    } else {
        if (aligned) {
            if constexpr (N == 4) {
                const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
                c[0] = v & 0xff; c[1] = (v >> 8) & 0xff; c[2] = (v >> 16) & 0xff; c[3] = v >> 24;
            } else if constexpr (N == 2) {
                const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(p));
                c[0] = v & 0xff; c[1] = v >> 8;
            } else {
                c[0] = p[0];
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; i++)
                c[i] = p[i];
        }
    }
}

template <int VW>
LH_DEV void load_px(const float *p, float (&v)[VW])
{
    if constexpr (VW == 4) {
        const lh_v4f t = __builtin_nontemporal_load(reinterpret_cast<const lh_v4f *>(p));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const lh_v2f t = __builtin_nontemporal_load(reinterpret_cast<const lh_v2f *>(p));
        v[0] = t.x; v[1] = t.y;
    }
}

// the same VW pixels from a frame uploaded as binary16 (the host entry points' half upload, lumahip_host.hip): exact widening
template <int VW>
LH_DEV void load_px_h(const _Float16 *p, float (&v)[VW])
{
    typedef _Float16 lh_v4h __attribute__((ext_vector_type(4)));
    typedef _Float16 lh_v2h __attribute__((ext_vector_type(2)));
    if constexpr (VW == 4) {
        const lh_v4h t = __builtin_nontemporal_load(reinterpret_cast<const lh_v4h *>(p));
        v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
    } else {
        const lh_v2h t = __builtin_nontemporal_load(reinterpret_cast<const lh_v2h *>(p));
        v[0] = (float)t.x; v[1] = (float)t.y;
    }
}

template <int VW>
LH_DEV void store_px(float *p, const float (&v)[VW])
{
    if constexpr (VW == 4) {
        const lh_v4f t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<lh_v4f *>(p));
    } else {
        const lh_v2f t = {v[0], v[1]};
        __builtin_nontemporal_store(t, reinterpret_cast<lh_v2f *>(p));
    }
}

LH_DEV void tile_coords(int t, const FrameGeom &g, int &f, int &bx, int &by)
{
    int r;
    if (g.interleave) {   // (kernel argument: uniform; all of this is scalar arithmetic)
        r = t / g.nframes;
        f = t - r * g.nframes;
    } else {
        f = t / g.tilesPerFrame;
        r = t - f * g.tilesPerFrame;
    }
    by = r / g.tilesX;
    bx = r - by * g.tilesX;
}

// ---- ENCODE ---------------------------------------------------------------------------------------
// CS: colour space; SUB: 4:2:0 (profiles 0/2) vs 4:4:4 (1/3); VW: pixels per thread per row (4 or 2);
// LM: luminance search mode (lut_index.hpp LutMode: 0 literal/LDS, 2 literal/global, 3 records/LDS, 4 records/global, 7 value-keyed records/LDS;
// 5 = YCbCr only: records in LDS for the composite luma -> code function, channel 0 carries the luma y, luma_device.hpp ycbcr_fwd;
// 6 = 5 + the half-input table in LDS: R', G', B' are three gathers for pixels whose inputs are binary16 values, luma_device.hpp half_lookup).
//
// Software pipeline: the six (VW=4: 16-byte) loads of the thread's NEXT unit are issued at the end of the current
// unit's iteration, one full iteration before they are needed (without this the kernel sat at ~45 % SQ_WAIT_ANY,
// profiles/r01_early_pmc_summary.txt); see the loop in k_encode for the order of loads and stores.

template <int VW>
struct EncUnit {
    float in[3][2][VW];
    int f, ux, uy;
    bool valid;
};

// IN16: the frames are binary16 (a.src[c] points at halves; offsets and strides count elements either way)
template <int VW, bool IN16 = false>
LH_DEV void enc_load(EncUnit<VW> &u, const EncArgs &a, int t, int tx, int ty, int NW)
{
    u.valid = false;
    if (t >= a.g.totalTiles)
        return;
    int bx, by;
    tile_coords(t, a.g, u.f, bx, by);
    u.ux = bx * 64 + tx;
    u.uy = by * NW + ty;
    if (u.ux >= a.g.unitsX || u.uy >= a.g.unitsY)
        return;
    u.valid = true;
    const size_t off = (size_t)u.f * a.frame_stride + (size_t)(2 * u.uy) * a.g.w + (size_t)u.ux * VW;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if constexpr (IN16) {
            const _Float16 *hp = reinterpret_cast<const _Float16 *>(a.src[c]);
            load_px_h<VW>(hp + off, u.in[c][0]);
            load_px_h<VW>(hp + off + a.g.w, u.in[c][1]);
        } else {
            load_px<VW>(a.src[c] + off, u.in[c][0]);
            load_px<VW>(a.src[c] + off + a.g.w, u.in[c][1]);
        }
    }
}

struct EncStats {
    float sum, mn, mx;
    int frame;
};

// Per-frame statistics: every wave adds its partial {sum, min, max} with three atomics.  Thousands of waves on ONE triple
// serialise at the L2 (a single 4K frame: 8192 waves, and the float min / max of the HIP headers are compare-and-swap loops
// -- the kernel took 1.08 ms instead of 30 us, profiles/r03_hostfed_trace.txt), so the arrivals are spread over STATS_SLOTS
// triples per frame (slot = workgroup index mod STATS_SLOTS) which k_fold_stats folds afterwards, and min / max use the
// hardware's integer atomics on the float's bit pattern: for v >= 0 the signed-integer order of the bits is the float order,
// for v < 0 the unsigned order is the reversed float order, and a negative float's bits exceed every non-negative one's as
// unsigned and undercut them as signed -- so min(v) = v >= 0 ? atomicMin(int) : atomicMax(unsigned), and the mirror image
// for max, are exact for any mix of signs starting from +inf / -inf.
constexpr int STATS_SLOTS = 32;

LH_DEV void atomic_min_f32(float *addr, float v)
{
    if (v >= 0.0f)
        atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else
        atomicMax(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}
LH_DEV void atomic_max_f32(float *addr, float v)
{
    if (v >= 0.0f)
        atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else
        atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

LH_DEV void stats_flush(EncStats &st, float *stats, int tx)
{
    if (st.frame >= 0) {
        const float s = wave_sum(st.sum), mn = wave_min(st.mn), mx = wave_max(st.mx);
        if (tx == 0) {
            float *p = stats + 3 * ((size_t)st.frame * STATS_SLOTS + (blockIdx.x & (STATS_SLOTS - 1)));
            atomicAdd(p + 0, s);
            if (mn == mn)   // (an all-NaN wave leaves +inf / -inf untouched, as fminf / fmaxf did within the wave)
                atomic_min_f32(p + 1, mn);
            if (mx == mx)
                atomic_max_f32(p + 2, mx);
        }
    }
    st.sum = 0.0f;
    st.mn = __builtin_inff();
    st.mx = -__builtin_inff();
}

// colour transform of one unit (row-major pixel order inside the unit: j = r*VW + i)
// HALF (YCbCr, LM == 6): the pixels go through the half-input table at `s_half` (LDS), which has `* sc` folded in
// Returns (HALF only) whether this thread's unit fell back to the general functions.
template <int CS, int VW, bool YCODE = false, bool HALF = false, typename K>
LH_DEV bool enc_transform(EncUnit<VW> &u, const EncArgs &a, const K &k, float (&c0)[2 * VW],
                          float (&c1)[2 * VW], float (&c2)[2 * VW], EncStats &st, const float *s_half = nullptr)
{
    bool general = false;
    if (!HALF && k.sc != 1.0f) {  // wave-uniform; x*1.0f == x, so the multiply is skipped for the default preScaling
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < VW; i++)
                    u.in[c][r][i] *= k.sc;
    }
    if constexpr (CS == CS_YCBCR) {
        float r8[2 * VW], g8[2 * VW], b8[2 * VW];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < VW; i++) {
                r8[r * VW + i] = u.in[0][r][i];
                g8[r * VW + i] = u.in[1][r][i];
                b8[r * VW + i] = u.in[2][r][i];
            }
        if constexpr (HALF)
            general = ycbcr_fwd_half_n<2 * VW>(r8, g8, b8, k, s_half, c0, c1, c2);
        else
            ycbcr_fwd_n<2 * VW, YCODE>(r8, g8, b8, k, c0, c1, c2);  // one "redo with the complete powf" decision per unit
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < VW; i++)
                xform_fwd<CS>(u.in[0][r][i], u.in[1][r][i], u.in[2][r][i], k, c0[r * VW + i], c1[r * VW + i], c2[r * VW + i]);
    }

    if (a.stats) {
#pragma unroll
        for (int j = 0; j < 2 * VW; j++) {
            st.sum += c0[j];
            st.mn = fminf(st.mn, c0[j]);
            st.mx = fmaxf(st.mx, c0[j]);
        }
    }
    return general;
}

// quantize + subsample + pack + store one transformed unit
template <int CS, bool SUB, int VW, int LM, typename LutPtr, typename IdxPtr>
LH_DEV void enc_emit(int f, int ux, int uy, const float (&c0)[2 * VW], const float (&c1)[2 * VW],
                     const float (&c2)[2 * VW], const EncArgs &a, LutPtr lut, IdxPtr idx)
{
    constexpr bool LUT_ALL = (CS == CS_RGB || CS == CS_XYZ);  // planes 1,2 also go through the LUT
    const float maxC = a.q.maxC;
    // plane 0, one row (VW searches in flight) at a time: keeps the live register set small enough for
    // 6 waves per SIMD
    {
        unsigned char *d = a.dst[0] + (size_t)f * a.dst_frame_stride[0] + (size_t)(2 * uy) * a.stride[0] +
                           (size_t)ux * VW * a.bps;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            float v[VW];
            int row[VW];
#pragma unroll
            for (int i = 0; i < VW; i++)
                v[i] = c0[r * VW + i];
            quantize_lut<LM, VW, CS == CS_LUV>(v, row, lut, idx, a.q);  // Lu'v': Y is >= 1e-4 or NaN
            store_samples<VW>(d + (size_t)r * a.stride[0], row, a.bps, a.aligned);
        }
    }

    // planes 1, 2
    if constexpr (SUB) {
        constexpr int NQ = VW / 2;
        float a1[NQ], a2[NQ];
#pragma unroll
        for (int qd = 0; qd < NQ; qd++) {
            // src/luma_encoder.cpp:287-289: 0.25f*(src[i] + src[i+1] + src[i+2w] + src[i+2w+1]); the factor is applied
            // below (LUT_ALL) or folded into the colour quantizer's scale (quantize_color_sum4)
            a1[qd] = ((c1[2 * qd] + c1[2 * qd + 1]) + c1[VW + 2 * qd]) + c1[VW + 2 * qd + 1];
            a2[qd] = ((c2[2 * qd] + c2[2 * qd + 1]) + c2[VW + 2 * qd]) + c2[VW + 2 * qd + 1];
        }
        int k1[NQ], k2[NQ];
        if constexpr (LUT_ALL) {
#pragma unroll
            for (int qd = 0; qd < NQ; qd++) {
                a1[qd] *= 0.25f;
                a2[qd] *= 0.25f;
            }
            quantize_lut<LM, NQ>(a1, k1, lut, idx, a.q);
            quantize_lut<LM, NQ>(a2, k2, lut, idx, a.q);
        } else {
            const float qc = 0.25f * maxC;
#pragma unroll
            for (int qd = 0; qd < NQ; qd++) {
                // Lu'v' chroma is positive or NaN (xform_fwd<CS_LUV>), so is the sum of four
                k1[qd] = quantize_color_sum4<CS == CS_LUV>(a1[qd], maxC, qc);
                k2[qd] = quantize_color_sum4<CS == CS_LUV>(a2[qd], maxC, qc);
            }
        }
        store_samples<NQ>(a.dst[1] + (size_t)f * a.dst_frame_stride[1] + (size_t)uy * a.stride[1] +
                              (size_t)ux * NQ * a.bps, k1, a.bps, a.aligned);
        store_samples<NQ>(a.dst[2] + (size_t)f * a.dst_frame_stride[2] + (size_t)uy * a.stride[2] +
                              (size_t)ux * NQ * a.bps, k2, a.bps, a.aligned);
    } else {
        int k1[2 * VW], k2[2 * VW];
        if constexpr (LUT_ALL) {
            quantize_lut<LM, 2 * VW>(c1, k1, lut, idx, a.q);
            quantize_lut<LM, 2 * VW>(c2, k2, lut, idx, a.q);
        } else {
#pragma unroll
            for (int j = 0; j < 2 * VW; j++) {
                k1[j] = quantize_color<CS == CS_LUV>(c1[j], maxC);
                k2[j] = quantize_color<CS == CS_LUV>(c2[j], maxC);
            }
        }
#pragma unroll
        for (int pl = 1; pl < 3; pl++) {
            unsigned char *d = a.dst[pl] + (size_t)f * a.dst_frame_stride[pl] + (size_t)(2 * uy) * a.stride[pl] +
                               (size_t)ux * VW * a.bps;
            int row[VW];
#pragma unroll
            for (int r = 0; r < 2; r++) {
#pragma unroll
                for (int i = 0; i < VW; i++)
                    row[i] = (pl == 1) ? k1[r * VW + i] : k2[r * VW + i];
                store_samples<VW>(d + (size_t)r * a.stride[pl], row, a.bps, a.aligned);
            }
        }
    }
}

template <int CS, bool SUB, int VW, int LM, bool IN16 = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(EncWaves<CS, SUB, VW>::value))) void k_encode(const EncArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert((LM != 5 && LM != 6) || CS == CS_YCBCR, "the composite records belong to the YCbCr kernels");
    constexpr bool HALF = (LM == 6);
    __shared__ int s_votes[HALF ? 2 : 1];   // HALF: waves of this workgroup that had units / that left the table in every one
    if (HALF && threadIdx.x == 0)
        s_votes[0] = s_votes[HALF ? 1 : 0] = 0;   // (stage_tables synchronises)
    constexpr int WHAT = (LM == 0 ? STAGE_LUT : 0) | ((LM == 3 || LM == 5 || LM == 6 || LM == 7) ? STAGE_REC : 0) |
                         (CS == CS_YCBCR ? (HALF ? STAGE_POWFN | STAGE_HALF : STAGE_POWF) : 0);
    stage_tables<WHAT>(smem, a.q, a.half);

    const float *s_lut = reinterpret_cast<const float *>(smem + lds_table_offset<WHAT>());        // LM == 0
    const uint32_t *s_rec = reinterpret_cast<const uint32_t *>(smem + lds_table_offset<WHAT>());  // LM == 3, 5, 6
    const float *s_half = reinterpret_cast<const float *>(smem + lds_table_offset<WHAT>() + lds_rec_bytes(a.q));  // LM == 6
    using PowTab = typename std::conditional<HALF, PowfTables, PowfTablesWide>::type;
    const XformConstT<PowTab> k = make_xform_const<CS, PowTab>(a.sc, a.q.Lmax, reinterpret_cast<const PowTab *>(smem));

    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int G = gridDim.x;

    EncStats st;
    st.frame = -1;
    st.sum = 0.0f;
    st.mn = __builtin_inff();
    st.mx = -__builtin_inff();

    // Per unit: transform, search / pack / store, THEN issue the next unit's loads (the current inputs are dead by then,
    // so both units share one set of registers); the other waves of the SIMD cover the load latency.  Issuing the loads
    // before the stores (round 1) measured 1.0-1.3 % slower HBM-fed, same-box, 5 of 5 interleaved rounds: with the
    // stores first the write bursts of a wave are not queued behind its own 6 KiB of reads.
    EncUnit<VW> u;
    int n_units = 0, n_general = 0;   // HALF: this wave's units, and those in which some lane left the table (wave-uniform)
    enc_load<VW, IN16>(u, a, blockIdx.x, tx, ty, NW);
    for (int t = blockIdx.x; t < a.g.totalTiles; t += G) {
        if (a.stats) {
            const int f = t / a.g.tilesPerFrame;  // wave-uniform
            if (f != st.frame) {
                stats_flush(st, a.stats, tx);
                st.frame = f;
            }
        }
        const bool valid = u.valid;
        const int f = u.f, ux = u.ux, uy = u.uy;
        float c0[2 * VW], c1[2 * VW], c2[2 * VW];
        bool general = false;
        if (valid)
            general = enc_transform<CS, VW, LM == 5 || LM == 6, HALF>(u, a, k, c0, c1, c2, st, s_half);
        if constexpr (HALF) {
            // a tile that overhangs the frame may leave this wave without a single pixel: such a unit is no evidence either way
            n_units += __builtin_amdgcn_ballot_w64(valid) != 0;
            n_general += __builtin_amdgcn_ballot_w64(general) != 0;
        }
        if (valid) {
            if constexpr (LM == 6)
                enc_emit<CS, SUB, VW, 5>(f, ux, uy, c0, c1, c2, a, s_lut, s_rec);
            else if constexpr (LM == 3 || LM == 5 || LM == 7)
                enc_emit<CS, SUB, VW, LM>(f, ux, uy, c0, c1, c2, a, s_lut, s_rec);
            else if constexpr (LM == 4)
                enc_emit<CS, SUB, VW, LM>(f, ux, uy, c0, c1, c2, a, a.q.lut, a.q.rec);
            else if constexpr (LM == 0)
                enc_emit<CS, SUB, VW, LM>(f, ux, uy, c0, c1, c2, a, s_lut, s_rec);
            else
                enc_emit<CS, SUB, VW, LM>(f, ux, uy, c0, c1, c2, a, a.q.lut, s_rec);
        }
        enc_load<VW, IN16>(u, a, t + G, tx, ty, NW);
    }
    if (a.stats)
        stats_flush(st, a.stats, tx);
    if constexpr (HALF) {
        // Feedback: this workgroup reports "not binary16 data" when EVERY unit of EVERY one of its waves had a lane on the
        // general path -- with 1 % of the pixels off the table that is nearly every workgroup (and the table kernel then costs
        // 1.4 x the per-pixel one), with 0.1 % (where the table still wins) a wave sees such a unit 40 % of the time and
        // sixteen waves in a row essentially never do.
        if (a.half_flag) {   // (kernel argument: uniform)
            if (tx == 0 && n_units > 0) {
                atomicAdd(&s_votes[0], 1);
                if (n_general == n_units)
                    atomicAdd(&s_votes[1], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0 && s_votes[0] > 0 && s_votes[0] == s_votes[1])
                __hip_atomic_store(a.half_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- DECODE ---------------------------------------------------------------------------------------
// GL: LUT read from global memory (bitdepth > 12) instead of LDS.  Same software pipeline as encode: the
// sample loads of the next unit are issued one iteration ahead, after the current unit's stores.
template <bool SUB, int VW>
struct DecUnit {
    int y[2][VW];
    int c1[SUB ? VW / 2 : 2 * VW];
    int c2[SUB ? VW / 2 : 2 * VW];
    int f, ux, uy;
    bool valid;
};

template <bool SUB, int VW>
LH_DEV void dec_load(DecUnit<SUB, VW> &u, const DecArgs &a, int t, int tx, int ty, int NW)
{
    u.valid = false;
    if (t >= a.g.totalTiles)
        return;
    int bx, by;
    tile_coords(t, a.g, u.f, bx, by);
    u.ux = bx * 64 + tx;
    u.uy = by * NW + ty;
    if (u.ux >= a.g.unitsX || u.uy >= a.g.unitsY)
        return;
    u.valid = true;
    const int f = u.f, ux = u.ux, uy = u.uy;
    {
        const unsigned char *s = a.src[0] + (size_t)f * a.src_frame_stride[0] + (size_t)(2 * uy) * a.stride[0] +
                                 (size_t)ux * VW * a.bps;
        load_samples<VW>(s, u.y[0], a.bps, a.aligned);
        load_samples<VW>(s + a.stride[0], u.y[1], a.bps, a.aligned);
    }
    if constexpr (SUB) {
        constexpr int NQ = VW / 2;
        load_samples<NQ>(a.src[1] + (size_t)f * a.src_frame_stride[1] + (size_t)uy * a.stride[1] + (size_t)ux * NQ * a.bps,
                         u.c1, a.bps, a.aligned);
        load_samples<NQ>(a.src[2] + (size_t)f * a.src_frame_stride[2] + (size_t)uy * a.stride[2] + (size_t)ux * NQ * a.bps,
                         u.c2, a.bps, a.aligned);
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            int row[VW];
            load_samples<VW>(a.src[1] + (size_t)f * a.src_frame_stride[1] + (size_t)(2 * uy + r) * a.stride[1] +
                                 (size_t)ux * VW * a.bps, row, a.bps, a.aligned);
#pragma unroll
            for (int i = 0; i < VW; i++)
                u.c1[r * VW + i] = row[i];
            load_samples<VW>(a.src[2] + (size_t)f * a.src_frame_stride[2] + (size_t)(2 * uy + r) * a.stride[2] +
                                 (size_t)ux * VW * a.bps, row, a.bps, a.aligned);
#pragma unroll
            for (int i = 0; i < VW; i++)
                u.c2[r * VW + i] = row[i];
        }
    }
}

// ---- the same loads as a REAL prefetch (round 6) ---------------------------------------------------------------------
// dec_load unpacks every row right behind its load, so the compiler has to wait for each load (s_waitcnt vmcnt(0): the unit's
// four loads are four serial round trips, each also waiting for every store issued before it) -- harmless where twenty waves per
// CU hide it, but the YCbCr kernels run four waves per SIMD with microseconds of arithmetic per unit and spent most of a unit's
// time in those waits.  DecRaw keeps the rows as loaded (six registers for 4:2:0 16-bit); dec_issue issues the loads of the NEXT
// unit before the current unit is processed, and dec_finish unpacks them after the current unit's arithmetic and BEFORE its
// stores (dec_process's hook): the one wait of an iteration then sits where only those loads and the previous unit's stores --
// a whole unit's arithmetic old -- are outstanding, and this unit's stores are not waited for until the next one's arithmetic
// is done.  (The compiler's waits are vmcnt(0) throughout: one counter for loads and stores on gfx950, and the conditional
// stores of the loop defeat exact counting.)  Rows the vector loads cannot take (a.aligned == 0: odd strides of a
// lossy upstream decoder) are loaded by dec_finish as dec_load does.
template <bool SUB, int VW>
struct DecRaw {
    uint32_t y[2][2];
    uint32_t c1[SUB ? 1 : 2][2], c2[SUB ? 1 : 2][2];
    int f, ux, uy;
    bool valid;
};

template <int N>
LH_DEV void load_raw(const unsigned char *p, uint32_t (&r)[2], int bps)
{
    const int bytes = N * bps;   // 8, 4, 2 or 1; uniform
    if (bytes == 8) {
        const lh_v2u v = __builtin_nontemporal_load(reinterpret_cast<const lh_v2u *>(p));
        r[0] = v.x;
        r[1] = v.y;
    } else if (bytes == 4) {
        r[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
    } else if (bytes == 2) {
        r[0] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(p));
    } else {
        r[0] = p[0];
    }
}

template <int N>
LH_DEV void unpack_raw(const uint32_t (&r)[2], int (&c)[N], int bps)
{
    if (bps == 2) {
        if constexpr (N == 4) {
            c[0] = r[0] & 0xffff; c[1] = r[0] >> 16; c[2] = r[1] & 0xffff; c[3] = r[1] >> 16;
        } else if constexpr (N == 2) {
            c[0] = r[0] & 0xffff; c[1] = r[0] >> 16;
        } else {
            c[0] = r[0] & 0xffff;
        }
    } else {
        if constexpr (N == 4) {
            c[0] = r[0] & 0xff; c[1] = (r[0] >> 8) & 0xff; c[2] = (r[0] >> 16) & 0xff; c[3] = r[0] >> 24;
        } else if constexpr (N == 2) {
            c[0] = r[0] & 0xff; c[1] = (r[0] >> 8) & 0xff;
        } else {
            c[0] = r[0] & 0xff;
        }
    }
}

template <bool SUB, int VW>
LH_DEV void dec_issue(DecRaw<SUB, VW> &u, const DecArgs &a, int t, int tx, int ty, int NW)
{
    u.valid = false;
    if (t >= a.g.totalTiles)
        return;
    int bx, by;
    tile_coords(t, a.g, u.f, bx, by);
    u.ux = bx * 64 + tx;
    u.uy = by * NW + ty;
    if (u.ux >= a.g.unitsX || u.uy >= a.g.unitsY)
        return;
    u.valid = true;
    if (!a.aligned)
        return;   // (dec_finish loads these rows itself)
    const int f = u.f, ux = u.ux, uy = u.uy;
    const unsigned char *s = a.src[0] + (size_t)f * a.src_frame_stride[0] + (size_t)(2 * uy) * a.stride[0] + (size_t)ux * VW * a.bps;
    load_raw<VW>(s, u.y[0], a.bps);
    load_raw<VW>(s + a.stride[0], u.y[1], a.bps);
    if constexpr (SUB) {
        constexpr int NQ = VW / 2;
        load_raw<NQ>(a.src[1] + (size_t)f * a.src_frame_stride[1] + (size_t)uy * a.stride[1] + (size_t)ux * NQ * a.bps, u.c1[0], a.bps);
        load_raw<NQ>(a.src[2] + (size_t)f * a.src_frame_stride[2] + (size_t)uy * a.stride[2] + (size_t)ux * NQ * a.bps, u.c2[0], a.bps);
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            load_raw<VW>(a.src[1] + (size_t)f * a.src_frame_stride[1] + (size_t)(2 * uy + r) * a.stride[1] + (size_t)ux * VW * a.bps, u.c1[r], a.bps);
            load_raw<VW>(a.src[2] + (size_t)f * a.src_frame_stride[2] + (size_t)(2 * uy + r) * a.stride[2] + (size_t)ux * VW * a.bps, u.c2[r], a.bps);
        }
    }
}

template <bool SUB, int VW>
LH_DEV void dec_finish(DecUnit<SUB, VW> &u, const DecRaw<SUB, VW> &w, const DecArgs &a)
{
    u.f = w.f;
    u.ux = w.ux;
    u.uy = w.uy;
    u.valid = w.valid;
    if (!w.valid)
        return;
    if (a.aligned) {
        unpack_raw<VW>(w.y[0], u.y[0], a.bps);
        unpack_raw<VW>(w.y[1], u.y[1], a.bps);
        if constexpr (SUB) {
            unpack_raw<VW / 2>(w.c1[0], u.c1, a.bps);
            unpack_raw<VW / 2>(w.c2[0], u.c2, a.bps);
        } else {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                int row[VW];
                unpack_raw<VW>(w.c1[r], row, a.bps);
#pragma unroll
                for (int i = 0; i < VW; i++)
                    u.c1[r * VW + i] = row[i];
                unpack_raw<VW>(w.c2[r], row, a.bps);
#pragma unroll
                for (int i = 0; i < VW; i++)
                    u.c2[r * VW + i] = row[i];
            }
        }
        return;
    }
    const int f = u.f, ux = u.ux, uy = u.uy;
    const unsigned char *s = a.src[0] + (size_t)f * a.src_frame_stride[0] + (size_t)(2 * uy) * a.stride[0] + (size_t)ux * VW * a.bps;
    load_samples<VW>(s, u.y[0], a.bps, 0);
    load_samples<VW>(s + a.stride[0], u.y[1], a.bps, 0);
    if constexpr (SUB) {
        constexpr int NQ = VW / 2;
        load_samples<NQ>(a.src[1] + (size_t)f * a.src_frame_stride[1] + (size_t)uy * a.stride[1] + (size_t)ux * NQ * a.bps, u.c1, a.bps, 0);
        load_samples<NQ>(a.src[2] + (size_t)f * a.src_frame_stride[2] + (size_t)uy * a.stride[2] + (size_t)ux * NQ * a.bps, u.c2, a.bps, 0);
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            int row[VW];
            load_samples<VW>(a.src[1] + (size_t)f * a.src_frame_stride[1] + (size_t)(2 * uy + r) * a.stride[1] + (size_t)ux * VW * a.bps, row, a.bps, 0);
#pragma unroll
            for (int i = 0; i < VW; i++)
                u.c1[r * VW + i] = row[i];
            load_samples<VW>(a.src[2] + (size_t)f * a.src_frame_stride[2] + (size_t)(2 * uy + r) * a.stride[2] + (size_t)ux * VW * a.bps, row, a.bps, 0);
#pragma unroll
            for (int i = 0; i < VW; i++)
                u.c2[r * VW + i] = row[i];
        }
    }
}

// the unit's samples must be in registers HERE (the compiler may not sink the unpacking, and with it the wait for the loads,
// below the stores that follow)
template <bool SUB, int VW>
LH_DEV void dec_pin(DecUnit<SUB, VW> &u)
{
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < VW; i++)
            asm volatile("" : "+v"(u.y[r][i]));
#pragma unroll
    for (int j = 0; j < (SUB ? VW / 2 : 2 * VW); j++) {
        asm volatile("" : "+v"(u.c1[j]));
        asm volatile("" : "+v"(u.c2[j]));
    }
}

// Whether this wave reads red and blue of the current unit from the per-stream tables (k_decode<..., RB>) or computes them.
// What a gather costs is decided by the CU's 32 KiB vector L1 (profiles/r05_ycbcr_decode_tables.txt, TCP_TCC_READ_REQ): a
// table line holds 32 consecutive luminance codes of ONE colour code; with the codes of a picture a lane's eight pixels read
// two lines per table, its neighbours and the rows above and below read the same ones, 99 % of the lanes hit the L1 and the
// launch takes 0.80 ms per 20 x 4K against 1.23 for six powf per pixel.  With unrelated codes in every pixel (the synthetic
// stream of SURVEY 8(d)) every lane of every gather is its own L2 request -- 3.2e8 per launch, the L2's request rate -- and the
// same launch takes 1.84 ms.  So a wave looks at its codes first, per unit, wave-uniformly (no divergence), ~20 instructions:
// a lane is "local" when its eight luminance codes lie within `near_y` of each other, its first luminance code within near_y
// of its left neighbour lane's, and its first colour codes within near_c (|dCb| + |dCr|) of that neighbour's; the wave gathers
// when at least RB_LOCAL_LANES of its lanes are.  Whichever path runs, the results are the same bits.
constexpr int RB_LOCAL_LANES = 48;

LH_DEV uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t acc)
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));   // |a - b| + acc
    return r;
}

// the value of lane - 1 (lane 0: 0)
LH_DEV uint32_t left_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }

template <bool SUB, int VW>
LH_DEV bool rb_wave_local(const DecUnit<SUB, VW> &u, int near_y, int near_c)
{
    if (near_y < 0)
        return true;
    int lo = u.y[0][0], hi = u.y[0][0];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < VW; i++) {
            lo = min(lo, u.y[r][i]);
            hi = max(hi, u.y[r][i]);
        }
    const uint32_t y0 = (uint32_t)u.y[0][0], b0 = (uint32_t)u.c1[0], r0 = (uint32_t)u.c2[0];
    const uint32_t dy = sad_u32(y0, left_lane(y0), 0u);
    const uint32_t dc = sad_u32(r0, left_lane(r0), sad_u32(b0, left_lane(b0), 0u));
    const bool local = (hi - lo) <= near_y && dy <= (uint32_t)near_y && dc <= (uint32_t)near_c;
    return __builtin_popcountll(__builtin_amdgcn_ballot_w64(local)) >= RB_LOCAL_LANES;
}

// SCFAST (YCbCr): the straight-line code divides by sc with the short division (XformConst::sc_mode == 1, ycbcr_inv_n)
// Returns (RB only; wave-uniform) whether the wave took red and blue of this unit from the tables.
// before_stores: called once between the unit's arithmetic and its stores (the prefetching loop of k_decode completes the NEXT
// unit's loads there)
struct DecNoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <int CS, bool SUB, int VW, bool DISP, bool UVTAB, bool YT = false, bool SCFAST = false, bool RB = false, typename LutPtr, typename K,
          typename Hook = DecNoHook>
LH_DEV bool dec_process(const DecUnit<SUB, VW> &u, const DecArgs &a, const K &k, LutPtr lut, const float *s_uv, Hook before_stores = Hook())
{
    bool gathered = false;
    static_assert(!RB || (YT && CS == CS_YCBCR), "the red / blue tables belong to the YCbCr kernels with the y table");
    constexpr bool LUT_ALL = (CS == CS_RGB || CS == CS_XYZ);
    const float maxC = a.q.maxC;
    const int maxVal = a.q.maxVal;
    float c0[2 * VW], c1[2 * VW], c2[2 * VW];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < VW; i++)
            c0[r * VW + i] = dequantize_lut(u.y[r][i], YT ? LutPtr(s_uv) : lut, maxVal);   // YT: the y table (behind the table in LDS)
    if constexpr (CS == CS_YCBCR && YT) {
        // (the chroma samples are read as terms from the per-code tables below)
    } else if constexpr (SUB) {
        constexpr int NQ = VW / 2;
#pragma unroll
        for (int qd = 0; qd < NQ; qd++) {
            const float v1 = LUT_ALL ? dequantize_lut(u.c1[qd], lut, maxVal) : dequantize_color(u.c1[qd], maxC);
            const float v2 = LUT_ALL ? dequantize_lut(u.c2[qd], lut, maxVal) : dequantize_color(u.c2[qd], maxC);
            // src/luma_decoder.cpp:229-234: the sample is replicated to its 2x2 block
            c1[2 * qd] = c1[2 * qd + 1] = c1[VW + 2 * qd] = c1[VW + 2 * qd + 1] = v1;
            c2[2 * qd] = c2[2 * qd + 1] = c2[VW + 2 * qd] = c2[VW + 2 * qd + 1] = v2;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2 * VW; j++) {
            c1[j] = LUT_ALL ? dequantize_lut(u.c1[j], lut, maxVal) : dequantize_color(u.c1[j], maxC);
            c2[j] = LUT_ALL ? dequantize_lut(u.c2[j], lut, maxVal) : dequantize_color(u.c2[j], maxC);
        }
    }
    float out[3][2][VW];
    if constexpr (CS == CS_LUV) {
        // chroma-only factors once per quad (4:2:0) / per pixel (4:4:4), on the short division path when the
        // codes are in range (always, unless a lossy upstream decoder produced garbage)
        constexpr int NC = SUB ? VW / 2 : 2 * VW;
        const int maxCi = (int)maxC;
        const float rmaxC = rcp_nr(maxC);
        LuvChroma ch[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) {
            if (u.c1[j] <= maxCi && u.c2[j] <= maxCi) {
                if constexpr (UVTAB)
                    ch[j] = luv_chroma_uv(s_uv[u.c1[j]], s_uv[u.c2[j]]);
                else
                    ch[j] = luv_chroma<true>(dequantize_color_safe(u.c1[j], maxC, rmaxC), dequantize_color_safe(u.c2[j], maxC, rmaxC));
            } else {
                ch[j] = luv_chroma<false>(dequantize_color(u.c1[j], maxC), dequantize_color(u.c2[j], maxC));
            }
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < VW; i++)
                luv_apply(c0[r * VW + i], ch[SUB ? i / 2 : r * VW + i], out[0][r][i], out[1][r][i], out[2][r][i]);
    } else if constexpr (CS == CS_YCBCR) {
        float r8[2 * VW], g8[2 * VW], b8[2 * VW];
        if constexpr (YT) {
            // the chroma terms from the per-code tables behind the y table (stage_tables, STAGE_CT); the codes themselves go
            // along for the complete functions, which a code beyond maxC (garbage from a lossy upstream decoder) calls for
            constexpr int NC = SUB ? VW / 2 : 2 * VW;
            const int maxCi = (int)maxC;
            const float *s_cb = s_uv + (a.q.lut_len + a.q.pad), *s_cr = s_cb + (maxCi + 1);
            bool bad = false;
#pragma unroll
            for (int j = 0; j < NC; j++) {
                bad = bad || u.c1[j] > maxCi || u.c2[j] > maxCi;
                const float t1 = s_cb[min(u.c1[j], maxCi)], t2 = s_cr[min(u.c2[j], maxCi)];
                if constexpr (SUB) {
                    c1[2 * j] = c1[2 * j + 1] = c1[VW + 2 * j] = c1[VW + 2 * j + 1] = t1;
                    c2[2 * j] = c2[2 * j + 1] = c2[VW + 2 * j] = c2[VW + 2 * j + 1] = t2;
                } else {
                    c1[j] = t1;
                    c2[j] = t2;
                }
            }
            bool gather = false;
            if constexpr (RB)
                gather = rb_wave_local<SUB, VW>(u, a.rb_near_y, a.rb_near_c);   // wave-uniform
            gathered = gather;
            if (RB && gather) {
                // red and blue of every pixel from the per-stream tables: two 4-byte gathers from global memory, issued before
                // green's two powf chains (which cover their latency); plain loads -- these lines are worth caching
                float tr[2 * VW], tb[2 * VW];
                const size_t n = (size_t)a.q.lut_len;
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int i = 0; i < VW; i++) {
                        const int j = SUB ? i / 2 : r * VW + i;
                        const size_t yc = (size_t)min(u.y[r][i], maxVal);
                        tb[r * VW + i] = a.rb[(size_t)min(u.c1[j], maxCi) * n + yc];
                        tr[r * VW + i] = a.rb[a.rb_plane + (size_t)min(u.c2[j], maxCi) * n + yc];
                    }
                const bool redo = ycbcr_inv_green_n<2 * VW, SCFAST ? 1 : 0, NC, SUB>(c0, c1, c2, k, r8, g8, b8, u.c1, u.c2, maxC, bad);
                if (!redo) {
#pragma unroll
                    for (int j = 0; j < 2 * VW; j++) {
                        r8[j] = tr[j];
                        b8[j] = tb[j];
                    }
                }
            } else {
                ycbcr_inv_n<2 * VW, true, SCFAST ? 1 : 0, true, NC, SUB>(c0, c1, c2, k, r8, g8, b8, u.c1, u.c2, maxC, bad);
            }
        } else {
            ycbcr_inv_n<2 * VW, YT, SCFAST ? 1 : 0>(c0, c1, c2, k, r8, g8, b8);
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < VW; i++) {
                out[0][r][i] = r8[r * VW + i];
                out[1][r][i] = g8[r * VW + i];
                out[2][r][i] = b8[r * VW + i];
            }
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < VW; i++)
                xform_inv<CS>(c0[r * VW + i], c1[r * VW + i], c2[r * VW + i], k, out[0][r][i], out[1][r][i], out[2][r][i]);
    }
    if (CS != CS_YCBCR && k.sc != 1.0f) {  // wave-uniform; x/1.0f == x, so the division is skipped for the default preScaling (YCbCr: ycbcr_inv_n has divided)
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < VW; i++)
                    out[c][r][i] = div_ieee(out[c][r][i], k.sc);
    }

    before_stores();
    if (!DISP || a.dst[0]) {
        const size_t px = (size_t)(2 * u.uy) * a.g.w + (size_t)u.ux * VW;
        if (a.rot_on) {   // (kernel argument: uniform; u.f is wave-uniform, so the base is a scalar select)
            // (readfirstlane: the frame index IS wave-uniform, and saying so keeps the choice among the three bases a scalar one.
            //  Without it the compiler -- once `a` is reachable through the loop's lambdas -- reads a.rot[j] with a VECTOR load from
            //  the kernel arguments and waits for it with vmcnt(0) right in front of the stores, i.e. for every store of the
            //  previous unit: 0.744 -> 0.683 of the roofline on this path, found in two default bench runs.)
            const int fu = __builtin_amdgcn_readfirstlane(u.f);
            const int k = fu / 3, j = fu - 3 * k;
            float *base = (j == 0 ? a.rot[0] : j == 1 ? a.rot[1] : a.rot[2]) + (size_t)k * a.frame_stride + px;
            const size_t n1 = (size_t)a.g.w * a.g.h;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                store_px<VW>(base + c * n1, out[c][0]);
                store_px<VW>(base + c * n1 + a.g.w, out[c][1]);
            }
        } else {
            const size_t off = (size_t)u.f * a.frame_stride + px;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                store_px<VW>(a.dst[c] + off, out[c][0]);
                store_px<VW>(a.dst[c] + off + a.g.w, out[c][1]);
            }
        }
    }
    if constexpr (DISP) {
        // Display-side transform of the player (src/lumaplay_dequantizer.frag:145-156) on the decoded,
        // already /sc-scaled RGB: exposure, optional 8-bit LDR simulation, optional sigmoid tone curve,
        // display gamma, 8-bit UNORM.  Not bit-pinned by the reference (its LUT texture is GL_LINEAR filtered),
        // so the fast fp32 pow (v_exp(v_log)) is used here; tolerance +-1 code against a float64 evaluation.
#pragma unroll
        for (int r = 0; r < 2; r++) {
            uint32_t px[VW];
#pragma unroll
            for (int i = 0; i < VW; i++) {
                uint32_t packed = 0xff000000u;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float v = out[c][r][i];
                    if (a.ldr_sim)
                        v = a.exposure * fmaxf(1.0f, fminf(256.0f, floorf(256.0f * v))) * (1.0f / 256.0f);
                    else
                        v = v * a.exposure;
                    if (a.do_tmo) {
                        const float vn = __powf(fmaxf(v, 0.0f), 0.8f);
                        v = vn / (vn + 0.83650957f);  // pow(0.8, 0.8)
                    }
                    v = __powf(fmaxf(v, 0.0f), a.inv_gamma);
                    v = fminf(fmaxf(v, 0.0f), 1.0f);  // NaN -> 0
                    packed |= (uint32_t)(v * 255.0f + 0.5f) << (8 * c);
                }
                px[i] = packed;
            }
            unsigned char *d = a.disp + (size_t)u.f * a.disp_frame_stride + (size_t)(2 * u.uy + r) * a.disp_stride +
                               (size_t)u.ux * VW * 4;
#pragma unroll
            for (int i = 0; i < VW; i++)
                reinterpret_cast<uint32_t *>(d)[i] = px[i];
        }
    }
    return gathered;
}

// DISP: additionally (or only) emit the RGBA8 display image -- a separate instantiation so that the plain
// decoder does not carry the epilogue's registers (it cost 8 % when it was a run-time branch)
// YT (YCbCr, table in LDS): additionally stage the per-stream y table and skip the first PQ evaluation of every pixel
// RB (with YT): red and blue from the per-stream (Y', Cr) / (Y', Cb) tables in global memory, green computed (DecArgs::rb)
template <int CS, bool SUB, int VW, bool GL, bool DISP = false, bool YT = false, bool RB = false>
__global__ __launch_bounds__(1024) void k_decode(const DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(!YT || (CS == CS_YCBCR && !GL), "the y table belongs to the YCbCr kernels with the table in LDS");
    static_assert(!RB || YT, "the red / blue tables need the y table");
    // GL: transfer-function table or chroma depth beyond 12 bits -- tables stay in global memory / are not built
    constexpr bool UVTAB = (CS == CS_LUV && !GL);
    // (The 16-entry powf tables instead -- no LDS bank conflicts, ten more issue cycles per powf -- were tried for this kernel
    // once the round-4 changes had cut its VALU work by 13 %: 3.3 % slower, 1.296 against 1.253 ms per 20 x 4K, same box.)
    using DecTab = PowfTablesWide;
    constexpr int WHAT = (GL ? 0 : STAGE_LUT) | (CS == CS_YCBCR ? STAGE_POWF : 0) | (UVTAB ? STAGE_UV : 0) | (YT ? STAGE_YT | STAGE_CT : 0);
    __shared__ int s_gathered[1];
    if (RB && threadIdx.x == 0)
        s_gathered[0] = 0;   // (before the barrier inside stage_tables: no wave can report ahead of the reset; the one reader
                             //  waits at the barrier at the end)
    stage_tables<WHAT>(smem, a.q);
    const float *s_lut = reinterpret_cast<const float *>(smem + lds_table_offset<WHAT>());
    const float *s_uv = reinterpret_cast<const float *>(smem + lds_table_offset<WHAT>() + lds_lut_bytes(a.q));
    const XformConstT<DecTab> k = make_xform_const<CS, DecTab>(a.sc, a.q.Lmax, reinterpret_cast<const DecTab *>(smem));

    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int G = gridDim.x;

    bool any_gather = false;
#ifndef LH_DEC_PF
#define LH_DEC_PF 1
#endif
    // PF: the real prefetch of DecRaw -- the kernels with microseconds of arithmetic per unit and four waves per SIMD (YCbCr)
    // How the next unit's rows are loaded (same-box A/B per kernel family, profiles/r06_decode_prefetch.txt):
    //   PFM 1  raw loads BEFORE this unit is processed, unpacked between its arithmetic and its stores (DecRaw above): the YCbCr
    //          kernels (-2.5 % on unrelated pixels, +5 % on pictures) and every 4:4:4 kernel (six row loads per unit that used to be
    //          six serial round trips: -10 ... -17 %);
    //   PFM 2  raw loads AFTER this unit's stores, unpacked at the top of the next iteration: the four loads go out back to back
    //          instead of one round trip each, in the order (stores, then loads) of the HBM-bound kernels: good for the HBM-bound
    //          4:2:0 kernels with 8-bit samples (-5 %), bad for 16-bit ones (below); not instantiated by default;
    //   PFM 0  dec_load as before round 6: the HBM-bound 4:2:0 kernels with 16-BIT samples (BASELINE's profile 2), where either
    //          form above is 4 - 6 % SLOWER -- four loads in flight per wave instead of one round trip after another is more
    //          concurrency than that traffic mix likes (the same finding as "5 workgroups per CU, not 8", lumahip_launch.hip).
    // The 4:2:0 kernels of the HBM-bound colour spaces keep PFM 0 for both sample sizes (carrying both loops and picking by the
    // sample size at run time gave 8-bit 4:2:0 -3.6 % and cost the 13-bit tables +1 %: not worth a second loop in BASELINE's kernel).
#ifndef LH_DEC_PFM_SUB
#define LH_DEC_PFM_SUB 0
#endif
    auto run = [&](auto pfm_tag) {
    constexpr int PFM = decltype(pfm_tag)::value;
    constexpr bool PF = PFM == 1;
    DecUnit<SUB, VW> cur, nxt;
    DecRaw<SUB, VW> nxt_w;
    if constexpr (PFM != 0) {
        dec_issue<SUB, VW>(nxt_w, a, blockIdx.x, tx, ty, NW);
        dec_finish<SUB, VW>(cur, nxt_w, a);
    } else {
        dec_load<SUB, VW>(cur, a, blockIdx.x, tx, ty, NW);
    }
    for (int t = blockIdx.x; t < a.g.totalTiles; t += G) {
        if constexpr (PF)
            dec_issue<SUB, VW>(nxt_w, a, t + G, tx, ty, NW);
        auto hook = [&]() {
            if constexpr (PF) {
                dec_finish<SUB, VW>(nxt, nxt_w, a);
                if (nxt.valid)
                    dec_pin<SUB, VW>(nxt);
            }
        };
        if (cur.valid) {
            if constexpr (CS == CS_YCBCR) {
                // two copies of the unit's code, chosen by a kernel argument: see ycbcr_inv_n on why not a run-time choice inside
                if (k.sc_mode == 1) {
                    if constexpr (GL)
                        dec_process<CS, SUB, VW, DISP, false, false, true>(cur, a, k, a.q.lut, s_uv, hook);
                    else
                        any_gather |= dec_process<CS, SUB, VW, DISP, UVTAB, YT, true, RB>(cur, a, k, s_lut, s_uv, hook);
                } else {
                    if constexpr (GL)
                        dec_process<CS, SUB, VW, DISP, false>(cur, a, k, a.q.lut, s_uv, hook);
                    else
                        any_gather |= dec_process<CS, SUB, VW, DISP, UVTAB, YT, false, RB>(cur, a, k, s_lut, s_uv, hook);
                }
            } else if constexpr (GL) {
                dec_process<CS, SUB, VW, DISP, false>(cur, a, k, a.q.lut, s_uv, hook);
            } else {
                dec_process<CS, SUB, VW, DISP, UVTAB, YT>(cur, a, k, s_lut, s_uv, hook);
            }
        } else {
            hook();
        }
        if constexpr (PFM == 2) {
            dec_issue<SUB, VW>(nxt_w, a, t + G, tx, ty, NW);
            dec_finish<SUB, VW>(nxt, nxt_w, a);   // (nothing between the loads depends on them: the waits sit here, behind all of them)
        } else if constexpr (PFM == 0) {
            dec_load<SUB, VW>(nxt, a, t + G, tx, ty, NW);
        }
        cur = nxt;
    }
    };   // run
    if constexpr (LH_DEC_PF == 0)
        run(std::integral_constant<int, 0>());
    else if constexpr (CS == CS_YCBCR || !SUB)
        run(std::integral_constant<int, 1>());
    else
        run(std::integral_constant<int, LH_DEC_PFM_SUB>());
    if constexpr (RB) {
        if (a.rb_flag) {   // (kernel argument: uniform)
            if (any_gather)
                s_gathered[0] = 1;   // (same value from every writer)
            __syncthreads();
            if (threadIdx.x == 0 && s_gathered[0])
                __hip_atomic_store(a.rb_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- stand-alone colour transform (LumaQuantizer::transformColorSpace as public API) ---------------
struct XfArgs {
    float *buf;
    size_t frame_stride;
    size_t chan_stride;  // w*h
    size_t n2;           // pixel pairs per frame
    int nframes;
    float sc;
    float Lmax;
};

template <int CS, bool FWD>
__global__ __launch_bounds__(256) void k_transform(const XfArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[CS == CS_YCBCR ? sizeof(PowfTablesWide) : 16];
    PowfTablesWide &s_pw = *reinterpret_cast<PowfTablesWide *>(s_raw);
    if constexpr (CS == CS_YCBCR) {
        stage_powf_tables(&s_pw);
        __syncthreads();
    }
    const XformConst k = make_xform_const<CS>(a.sc, a.Lmax, &s_pw);
    const size_t total = a.n2 * a.nframes;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / a.n2, j = i - f * a.n2;
        float *p = a.buf + f * a.frame_stride + 2 * j;
        float v0[2], v1[2], v2[2], o0[2], o1[2], o2[2];
        load_px<2>(p, v0);
        load_px<2>(p + a.chan_stride, v1);
        load_px<2>(p + 2 * a.chan_stride, v2);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if constexpr (FWD) {
                xform_fwd<CS>(v0[e] * k.sc, v1[e] * k.sc, v2[e] * k.sc, k, o0[e], o1[e], o2[e]);
            } else {
                xform_inv<CS>(v0[e], v1[e], v2[e], k, o0[e], o1[e], o2[e]);
                o0[e] = div_ieee(o0[e], k.sc);
                o1[e] = div_ieee(o1[e], k.sc);
                o2[e] = div_ieee(o2[e], k.sc);
            }
        }
        store_px<2>(p, o0);
        store_px<2>(p + a.chan_stride, o1);
        store_px<2>(p + 2 * a.chan_stride, o2);
    }
}

// ---- the reference's mean luminance, exactly ---------------------------------------------------------
// LumaEncoder::setVpxChannel accumulates plane 0 into ONE fp32 variable in raster order (src/luma_encoder.cpp:276,294,314)
// and warns when avg / (w*h) <= 1.  A sequential fp32 sum is not associative: the encode kernels' per-frame statistics
// (tree / atomic order) give a more accurate but different number, typically 1e-4 relative apart at 4K.  When the caller
// needs the reference's value -- the host entry points do when the fast mean is within 1 % of the threshold -- these two
// kernels reproduce it: k_channel0 writes the transformed channel 0 of one frame, k_seq_sum adds it up in the reference's
// order (one wave; lanes load 64 consecutive values at a time, the adds run lane-uniformly in raster order; ~25 ms at 4K).
template <int CS, bool IN16 = false>
__global__ __launch_bounds__(256) void k_channel0(const float *src, size_t chan_stride, size_t n, float sc, float Lmax, float *out)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[CS == CS_YCBCR ? sizeof(PowfTablesWide) : 16];
    PowfTablesWide &s_pw = *reinterpret_cast<PowfTablesWide *>(s_raw);
    if constexpr (CS == CS_YCBCR) {
        stage_powf_tables(&s_pw);
        __syncthreads();
    }
    const XformConst k = make_xform_const<CS>(sc, Lmax, &s_pw);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float c0, c1, c2;
        float r, g, b;
        if constexpr (IN16) {   // the frame was uploaded as binary16
            const _Float16 *hp = reinterpret_cast<const _Float16 *>(src);
            r = (float)hp[i];
            g = (float)hp[i + chan_stride];
            b = (float)hp[i + 2 * chan_stride];
        } else {
            r = src[i];
            g = src[i + chan_stride];
            b = src[i + 2 * chan_stride];
        }
        xform_fwd<CS>(r * sc, g * sc, b * sc, k, c0, c1, c2);
        out[i] = c0;
    }
}

// Test probe: quantize_lut<LM, 4, NONNEG> -- the instantiation the Lu'v' encode kernels call for a row of four
// luminances -- over consecutive fp32 bit patterns (tests/test_gpu_exhaustive.py; NONNEG promises v >= 0 or NaN, so
// that sweep covers 0 .. 0x7fffffff plus the sign-set NaNs 0xff800001 .. 0xffffffff).
template <int LM, bool NONNEG>
__global__ __launch_bounds__(256) void k_quantize_probe(const QuantDev q, uint16_t *out, uint32_t first_bits, size_t n4)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    stage_tables<(LM == 0 ? STAGE_LUT : 0) | ((LM == 3 || LM == 7) ? STAGE_REC : 0)>(smem, q);
    const float *s_lut = reinterpret_cast<const float *>(smem);
    const uint32_t *s_rec = reinterpret_cast<const uint32_t *>(smem);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float v[4];
        int c[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            v[j] = __uint_as_float(first_bits + (uint32_t)(4 * i + j));
        if constexpr (LM == 3 || LM == 7)
            quantize_lut<LM, 4, NONNEG>(v, c, s_lut, s_rec, q);
        else if constexpr (LM == 4)
            quantize_lut<LM, 4, NONNEG>(v, c, q.lut, q.rec, q);
        else if constexpr (LM == 0)
            quantize_lut<LM, 4, NONNEG>(v, c, s_lut, s_rec, q);
        else
            quantize_lut<LM, 4, NONNEG>(v, c, q.lut, s_rec, q);
        store_samples<4>(reinterpret_cast<unsigned char *>(out + 4 * i), c, 2, 1);
    }
}


}  // namespace lh
