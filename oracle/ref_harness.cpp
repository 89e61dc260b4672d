/*
 * ref_harness.cpp -- thin extern "C" window onto the REAL reference LumaQuantizer.
 * TEST INFRASTRUCTURE ONLY; built only where /root/reference exists (this container).
 *
 * oracle/Makefile compiles /root/reference/src/luma_quantizer.cpp *where it lies*, unmodified,
 * together with this file into oracle/_ref/libluma_ref.so.  Nothing from the reference is copied
 * into the repo; this file only #includes the reference's public header and forwards calls.
 *
 * What is NOT reachable this way: LumaEncoder::setVpxChannel / LumaDecoder::getVpxChannels
 * (src/luma_encoder.cpp:260-317, src/luma_decoder.cpp:205-240) live in translation units that
 * need libvpx (a tarball whose headers include a configure-generated vpx_config.h), libebml and
 * libmatroska -- "unbuildable here" under the no-stand-ins rule.  Those two loops are pinned by the
 * Y/U/V plane digests that SURVEY.md 8(c) recorded from the full reference build instead.
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

} /* extern "C" */
