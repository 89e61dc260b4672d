// host_lut.cpp -- host-side construction of the transfer-function table, i.e. the part of
// LumaQuantizer::setQuantizer that stays on the CPU (src/luma_quantizer.cpp:114-169,172-212): at most
// 65536 libm calls once per stream.  It calls the host libm's powf / log10f exactly as the reference does,
// so the table is bit-identical to what the reference would write into MKV attachment 434 on this host.
// PSI / JND-HDR-VDP tables are data (captured by tools/capture_ptf_tables.py) loaded from
// lumahdrv_amd/data/ptf_<name>_<bits>.f32.
#include <dlfcn.h>
#include <math.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define LUMAHIP_EXPERIMENTAL   /* the library defines what the experimental section of the header declares */
#include "../../include/lumahip.h"
#include "host_lut.hpp"
#include "pow_glibc.hpp"
#include "lut_index.hpp"

namespace {

// LumaQuantizer::transformPQ, decode branch (src/luma_quantizer.cpp:496-500); constants are double
// literals narrowed to float, as in the reference
float pq_decode_host(float L, float val)
{
    const float m = 78.8438, n = 0.1593, c1 = 0.8359, c2 = 18.8516, c3 = 18.6875;
    float Vp = powf(val, 1.0f / m);
    return L * powf(std::max(0.0f, (Vp - c1)) / (c2 - c3 * Vp), 1.0f / n);
}

// LumaQuantizer::transformPQ, encode branch (src/luma_quantizer.cpp:491-494)
float pq_encode_host(float L, float val)
{
    const float m = 78.8438, n = 0.1593, c1 = 0.8359, c2 = 18.8516, c3 = 18.6875;
    float Lp = powf(val / L, n);
    return powf((c1 + c2 * Lp) / (1 + c3 * Lp), m);
}

// LumaQuantizer::transformLog, decode branch (src/luma_quantizer.cpp:509)
float log_decode_host(float Lmax, float Lmin, float val)
{
    return powf(10.0f, val * (log10f(Lmax) - log10f(Lmin)) + log10f(Lmin));
}

std::string data_dir()
{
    if (const char *e = getenv("LUMAHIP_DATA_DIR"))
        return e;
    Dl_info info;
    if (dladdr((const void *)&lumahip_build_lut, &info) && info.dli_fname) {
        std::string p = info.dli_fname;  // .../lumahdrv_amd/lib/liblumahip.so
        size_t s = p.rfind('/');
        if (s != std::string::npos) {
            p = p.substr(0, s);
            s = p.rfind('/');
            if (s != std::string::npos)
                return p.substr(0, s) + "/data";
        }
    }
    return "lumahdrv_amd/data";
}

}  // namespace

extern "C" int lumahip_build_lut(int ptf, unsigned bitdepth, float maxLum, float minLum, float *out, size_t n)
{
    if (!out || bitdepth < 1 || bitdepth > 16 || n != ((size_t)1 << bitdepth))
        return LUMAHIP_ERR_ARG;
    const unsigned maxVal = (unsigned)((int)powf(2.0f, (float)bitdepth) - 1);
    switch (ptf) {
    case LUMAHIP_PTF_PQ:
        for (size_t i = 0; i <= maxVal; i++)
            out[i] = pq_decode_host(maxLum, (float)i / maxVal);
        return LUMAHIP_OK;
    case LUMAHIP_PTF_LOG:
        for (size_t i = 0; i <= maxVal; i++)
            out[i] = log_decode_host(maxLum, minLum, (float)i / maxVal);
        return LUMAHIP_OK;
    case LUMAHIP_PTF_LINEAR:
        for (size_t i = 0; i <= maxVal; i++)
            out[i] = maxLum * ((float)i / maxVal);
        return LUMAHIP_OK;
    case LUMAHIP_PTF_JND_HDRVDP:
    case LUMAHIP_PTF_PSI:
    default: {
        // the reference selects the 10- or 11-bit table for those depths and the 12-bit table for every
        // other depth, then copies maxVal+1 entries (src/luma_quantizer.cpp:128-169) -- reading past the
        // table when bitdepth > 12; that case is rejected here.
        const unsigned tb = (bitdepth == 10 || bitdepth == 11) ? bitdepth : 12;
        if (n > ((size_t)1 << tb))
            return LUMAHIP_ERR_UNSUPPORTED;
        const char *nm = (ptf == LUMAHIP_PTF_JND_HDRVDP) ? "jnd_hdrvdp" : "psi";
        char path[1024];
        snprintf(path, sizeof path, "%s/ptf_%s_%u.f32", data_dir().c_str(), nm, tb);
        FILE *f = fopen(path, "rb");
        if (!f)
            return LUMAHIP_ERR_STATE;
        std::vector<float> t((size_t)1 << tb);
        const size_t got = fread(t.data(), sizeof(float), t.size(), f);
        fclose(f);
        if (got != t.size())
            return LUMAHIP_ERR_STATE;
        memcpy(out, t.data(), n * sizeof(float));
        return LUMAHIP_OK;
    }
    }
}

// ---- the two per-stream tables of the YCbCr kernels (luma_device.hpp: ycbcr_fwd / ycbcr_inv), built with the host libm -- the
// very function the reference calls -- once per stream instead of two powf per pixel on the device.
namespace lh {

// Encode side: the luminance code the reference gives a pixel as a function of t = 219 y + 16, the numerator of
// src/luma_quantizer.cpp:337 (y = the pixel's luma): code(t) = search(PQdec(t / 255)), PQdec = :496-500, search = :222-235.
// (Keyed by t rather than y because the codes are roughly uniform in t -- four binary octaves, 16 ... 256, hold every
// threshold, so the records are a few KiB; keyed by y they span eleven octaves and 46 KiB.)
// The luma is a sum of non-negative products, so t >= 16 (or NaN) in the kernels; below 16 the function is continued as a
// constant, which keeps the thresholds of the never-used codes below code(16) -- spread over dozens of octaves -- out of the records.
int ycbcr_luma_code_host(float t, const float *lut, int maxVal, float Lmax)
{
    if (t < 16.0f)
        t = 16.0f;
    const float c0 = pq_decode_host(Lmax, t / 255.0f);
    return quantize_literal_host(c0, lut, maxVal);
}

// Decode side: what src/luma_quantizer.cpp:447-448 make of a table value, y = (255 PQenc(lut[i]) - 16) / 219.
void ycbcr_ytab_host(const float *lut, size_t n, float Lmax, float *out)
{
    for (size_t i = 0; i < n; i++) {
        float y = pq_encode_host(Lmax, lut[i]);
        out[i] = (255.0f * y - 16.0f) / 219.0f;
    }
}


// Encode side, binary16 inputs (the reference's EXR reader produces nothing else, src/exr_interface.cpp:77-146): entry i, i the
// bit pattern of a half in +0 ... +inf, holds PQenc(std::max(x * sc, 1e-10f)) (src/luma_quantizer.cpp:331-333, 491-494) for that
// half's value x -- the non-linear R' / G' / B' of a pixel as a function of 16 input bits, per (sc, Lmax).  Returns false, and
// the kernels keep evaluating PQenc per pixel, unless sc is a positive finite number, Lmax lies in [1e-6, 1e9] and every entry is a NaN or lies in
// [7e-7, 2]: the range the kernels' short divisions behind the table are licensed for (luma_device.hpp ycbcr_fwd_half_n).
bool ycbcr_half_table_host(float sc, float Lmax, float *out)
{
    if (!(sc > 0.0f) || !(sc <= 3.0e38f) || !(Lmax >= 1e-6f && Lmax <= 1e9f))   // (the peak range of the kernels' general path, ycbcr_fwd_n)
        return false;
    for (unsigned i = 0; i <= 0x7C00u; i++) {
        // binary16 -> binary32, exactly
        const unsigned e = i >> 10, m = i & 0x3ffu;
        float x;
        if (e == 0)
            x = ldexpf((float)m, -24);
        else if (e == 31)
            x = INFINITY;  // i == 0x7C00, m == 0
        else
            x = ldexpf((float)(m | 0x400u), (int)e - 25);
        const float arg = std::max(x * sc, 1e-10f);
        const float v = pq_encode_host(Lmax, arg);
        out[i] = v;
        if (!(v != v) && !(v >= 7e-7f && v <= 2.0f))
            return false;
        // A pixel that misses the table in the same launch goes through the kernels' restatement of glibc 2.35's powf
        // (pow_glibc.hpp); what the table holds came from THIS host's libm.  On the hosts this was written for the two are the
        // same function bit for bit (tools/verify_powf.cpp); on a host whose libm differs (musl, another glibc) they would
        // not be, and a plane would then depend on whether a pixel's inputs happen to be halves.  So every entry is checked
        // against the restatement evaluated on the host, and the table is refused -- the kernels evaluate every pixel -- on
        // the first difference.
        {
            const float m = 78.8438, n = 0.1593, c1 = 0.8359, c2 = 18.8516, c3 = 18.6875;
            const float Lp = powf_glibc(arg / Lmax, n, kPowfTablesHost);
            const float r = powf_glibc((c1 + c2 * Lp) / (1 + c3 * Lp), m, kPowfTablesHost);
            if (std::memcmp(&r, &v, sizeof r) != 0 && !(r != r && v != v))
                return false;
        }
    }
    return true;
}

}  // namespace lh

// ---- LumaQuantizer::quantize / dequantize for ONE value (src/luma_quantizer.cpp:215-264), on the host.  The reference's are
// ~50 ns scalar calls and code written against it may loop over them (its own plane loops do, src/luma_encoder.cpp:293); the
// facade's per-value members come here instead of paying a kernel launch per sample.  Frames and arrays never do: those are
// the kernels' job.  Same arithmetic as the reference: the literal bisection + nearest-of-two on the table, std::min / std::max
// argument order, floor(maxC*v + 0.5f) for the colour channels.
extern "C" int lumahip_quantize_value_host(const float *lut, size_t lut_len, int colorspace, unsigned bitdepthC, float val,
                                           unsigned ch, float *out)
{
    if (!lut || !out || lut_len < 2 || lut_len > 65536 || bitdepthC < 1 || bitdepthC > 16)
        return LUMAHIP_ERR_ARG;
    if (ch == 0 || colorspace == LUMAHIP_CS_RGB || colorspace == LUMAHIP_CS_XYZ) {
        *out = (float)lh::quantize_literal_host(val, lut, (int)lut_len - 1);
    } else {
        const unsigned maxC = (1u << bitdepthC) - 1;
        float res = floorf(maxC * val + 0.5f);
        res = std::max(0.0f, std::min((float)maxC, res));
        *out = res;
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_dequantize_value_host(const float *lut, size_t lut_len, int colorspace, unsigned bitdepthC, float val,
                                             unsigned ch, float *out)
{
    if (!lut || !out || lut_len < 2 || lut_len > 65536 || bitdepthC < 1 || bitdepthC > 16)
        return LUMAHIP_ERR_ARG;
    if (ch == 0 || colorspace == LUMAHIP_CS_RGB || colorspace == LUMAHIP_CS_XYZ) {
        const unsigned maxVal = (unsigned)lut_len - 1;
        if (val < 0)
            *out = lut[0];
        else if (val >= maxVal)
            *out = lut[maxVal];
        else
            *out = lut[(val != val) ? maxVal : (unsigned)(int)val];   // (a NaN index is undefined in the reference; the array kernels answer the top entry too)
    } else {
        const unsigned maxC = (1u << bitdepthC) - 1;
        *out = std::max(val / maxC, 1e-10f);
    }
    return LUMAHIP_OK;
}
