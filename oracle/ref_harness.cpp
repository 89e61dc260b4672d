/*
 * ref_harness.cpp -- thin extern "C" window onto the REAL reference LumaQuantizer.
 * TEST INFRASTRUCTURE ONLY; built only where /root/reference exists (this container).
 *
 * oracle/Makefile compiles /root/reference/src/luma_quantizer.cpp *where it lies*, unmodified,
 * together with this file into oracle/_ref/libluma_ref.so.  Nothing from the reference is copied
 * into the repo; this file only #includes the reference's public header and forwards calls.
 *
 * The plane loops LumaEncoder::setVpxChannel / LumaDecoder::getVpxChannels (src/luma_encoder.cpp:260-317,
 * src/luma_decoder.cpp:205-240) are reached through the separate oracle/ref_planes_harness.cpp build
 * (`make ref_planes`); this library stays the light one the bench's cpu_baseline times.
 */
#include "luma_quantizer.h" /* /root/reference/include/luma, via -I */

#include <cstring>

extern "C" {

void *ref_create() { return new LumaQuantizer(); }
void ref_destroy(void *q) { delete static_cast<LumaQuantizer *>(q); }

void ref_set_quantizer(void *q, int ptf, unsigned bitdepth, int cs, unsigned bitdepthC, float maxLum, float minLum)
{
    static_cast<LumaQuantizer *>(q)->setQuantizer(static_cast<LumaQuantizer::ptf_t>(ptf), bitdepth,
                                                  static_cast<LumaQuantizer::colorSpace_t>(cs), bitdepthC,
                                                  maxLum, minLum);
}

/* number of LUT entries = getSize()+1 (include/luma/luma_quantizer.h:109) */
unsigned ref_get_size(void *q) { return static_cast<LumaQuantizer *>(q)->getSize(); }
const float *ref_get_mapping(void *q) { return static_cast<LumaQuantizer *>(q)->getMapping(); }

/* what LumaDecoder::initialize does with attachment 434 (src/luma_decoder.cpp:122) */
void ref_overwrite_mapping(void *q, const float *lut, unsigned n)
{
    std::memcpy((void *)static_cast<LumaQuantizer *>(q)->getMapping(), lut, n * sizeof(float));
}

float ref_quantize(void *q, float v, unsigned ch) { return static_cast<LumaQuantizer *>(q)->quantize(v, ch); }
float ref_dequantize(void *q, float v, unsigned ch) { return static_cast<LumaQuantizer *>(q)->dequantize(v, ch); }

void ref_quantize_array(void *q, const float *in, float *out, size_t n, unsigned ch)
{
    LumaQuantizer *lq = static_cast<LumaQuantizer *>(q);
    for (size_t i = 0; i < n; i++)
        out[i] = lq->quantize(in[i], ch);
}

void ref_dequantize_array(void *q, const float *in, float *out, size_t n, unsigned ch)
{
    LumaQuantizer *lq = static_cast<LumaQuantizer *>(q);
    for (size_t i = 0; i < n; i++)
        out[i] = lq->dequantize(in[i], ch);
}

/* in-place on caller memory: a LumaFrame is pointed at the caller's buffer for the duration of the call */
int ref_transform_color_space(void *q, float *buf, unsigned w, unsigned h, int toCs, float sc)
{
    LumaFrame f;
    f.width = w;
    f.height = h;
    f.channels = 3;
    f.buffer = buf;
    bool ok = static_cast<LumaQuantizer *>(q)->transformColorSpace(&f, toCs != 0, sc);
    f.buffer = NULL; /* do not let ~LumaFrame free the caller's memory */
    return ok ? 1 : 0;
}

/* Whole-frame encode for the bench's cpu_baseline ("kind": "reference"): the REAL
 * LumaQuantizer::transformColorSpace and LumaQuantizer::quantize (one call per sample, as the reference makes
 * them), driven by a restatement of the plane loop of LumaEncoder::setVpxChannel (src/luma_encoder.cpp:260-317)
 * (the loop itself, compiled from the reference, is oracle/_ref/ref_planes_tool; tests compare the two).  Mutates `buf` like the
 * reference does.  profile as in the reference: 0/2 = 4:2:0, 1/3 = 4:4:4; > 1 = 16-bit samples. */
void ref_encode_frame(void *qv, float *buf, unsigned w, unsigned h, float sc, int profile, unsigned char *const planes[3],
                      const int stride[3], float *avg_out)
{
    LumaQuantizer *q = static_cast<LumaQuantizer *>(qv);
    ref_transform_color_space(qv, buf, w, h, 1, sc);
    const bool sub = (profile == 0 || profile == 2);
    const int m = profile > 1 ? 2 : 1;
    for (int plane = 0; plane < 3; plane++) {
        const float *src = buf + (size_t)plane * w * h;
        unsigned char *out = planes[plane];
        const int pw = (plane && sub) ? (int)((w + 1) >> 1) : (int)w, ph = (plane && sub) ? (int)((h + 1) >> 1) : (int)h;
        float avg = 0.0f;
        for (int y = 0; y < ph; y++)
            for (int x = 0; x < pw; x++) {
                float res;
                if (plane && sub) {
                    const size_t i1 = 2 * (size_t)x + 4 * (size_t)y * pw, i2 = i1 + 2 * (size_t)pw;
                    res = 0.25f * (src[i1] + src[i1 + 1] + src[i2] + src[i2 + 1]);
                } else {
                    res = src[x + (size_t)y * pw];
                    avg += res;
                }
                res = q->quantize(res, plane);
                if (profile > 1) {
                    unsigned char bl = res / 256;
                    unsigned char bh = res - bl * 256;
                    out[m * x + (size_t)y * stride[plane] + 1] = bl;
                    out[m * x + (size_t)y * stride[plane]] = bh;
                } else {
                    out[m * x + (size_t)y * stride[plane]] = (unsigned char)(int)res;
                }
            }
        if (plane == 0 && avg_out)
            *avg_out = avg / (pw * ph);
    }
}

} /* extern "C" */
