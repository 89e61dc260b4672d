// Facade LumaQuantizer: host bookkeeping + C-ABI calls.  The table is built by lumahip_build_lut (host libm,
// as the reference does) and uploaded with lumahip_set_quantizer; transformColorSpace runs on the GPU.
#include "../../../include/luma/luma_quantizer.h"

#include <cstdio>

#include "../../../include/luma/luma_exception.h"
#include "../../../include/lumahip.h"

LumaQuantizer::LumaQuantizer()
    : m_ctx(NULL), m_ptf(PTF_PSI), m_colorSpace(CS_LUV), m_Lmax(10000.0f), m_Lmin(0.005f), m_maxVal(0),
      m_maxValColor(0), m_bitdepth(0), m_bitdepthColor(0), m_configured(false)
{
}

LumaQuantizer::~LumaQuantizer()
{
    if (m_ctx)
        lumahip_destroy(m_ctx);
}

void LumaQuantizer::requireContext()
{
    if (m_ctx)
        return;
    const int rc = lumahip_create(&m_ctx, -1);
    if (rc != LUMAHIP_OK)
        throw LumaException("No usable HIP device for the Luma HDRv quantizer (there is no CPU fallback)");
}

std::string LumaQuantizer::name(ptf_t ptf)
{
    switch (ptf) {
    case PTF_PQ: return "Perceptual quantizer (PQ, SMPTE ST 2084)";
    case PTF_LOG: return "Logarithmic";
    case PTF_JND_HDRVDP: return "JND HDR-VDP";
    case PTF_PSI: return "Perceptual - Ferwerda's t.v.i.";
    case PTF_LINEAR: return "Linear scaling";
    }
    return "Undefined";
}

std::string LumaQuantizer::name(colorSpace_t cs)
{
    switch (cs) {
    case CS_LUV: return "Lu'v'";
    case CS_RGB: return "RGB";
    case CS_YCBCR: return "YCbCr (ITU-R BT.2020)";
    case CS_XYZ: return "XYZ";
    }
    return "Undefined";
}

void LumaQuantizer::setQuantizer(ptf_t ptf, unsigned int bitdepth, colorSpace_t cs, unsigned int bitdepthC, float maxLum,
                                 float minLum)
{
    requireContext();
    if (bitdepth < 1 || bitdepth > 16 || bitdepthC < 1 || bitdepthC > 16)
        throw LumaException("PTF / colour bit depth must be in 1..16");
    m_ptf = ptf;
    m_bitdepth = bitdepth;
    m_maxVal = (1u << bitdepth) - 1;
    m_colorSpace = cs;
    m_bitdepthColor = bitdepthC;
    m_maxValColor = (1u << bitdepthC) - 1;
    m_Lmax = maxLum;
    m_Lmin = minLum;
    m_mapping.assign((size_t)m_maxVal + 1, 0.0f);
    const int rc = lumahip_build_lut((int)ptf, bitdepth, maxLum, minLum, m_mapping.data(), m_mapping.size());
    if (rc == LUMAHIP_ERR_UNSUPPORTED)
        throw LumaException("PSI / JND-HDR-VDP tables exist for at most 12 bits");
    if (rc != LUMAHIP_OK)
        throw LumaException("Cannot build the transfer function table (missing lumahdrv_amd/data/ptf_*.f32?)");
    m_configured = true;
    syncMapping();
}

void LumaQuantizer::syncMapping()
{
    if (!m_configured)
        return;
    const int rc = lumahip_set_quantizer(m_ctx, (int)m_ptf, m_bitdepth, (int)m_colorSpace, m_bitdepthColor, m_Lmax, m_Lmin,
                                         m_mapping.data(), m_mapping.size());
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_ctx));
}

// Per-value API of the reference (src/luma_quantizer.cpp:215-264; its own plane loops call these per sample, and so may code
// written against it).  Evaluated on the host table m_mapping -- the one getMapping() hands out, so a table written through that
// pointer is seen at once, as in the reference -- by the library's host-only scalar forms: a call costs what the reference's
// does (tens of ns), not a kernel launch.  Bulk work belongs to the frame-level calls and lumahip_quantize_array_host, which
// run the kernels; nothing in this library processes a frame through these two members.
float LumaQuantizer::quantize(const float val, const unsigned int ch) const
{
    if (!m_configured)
        throw LumaException("LumaQuantizer::quantize before setQuantizer");
    float out = 0.0f;
    if (lumahip_quantize_value_host(m_mapping.data(), m_mapping.size(), (int)m_colorSpace, m_bitdepthColor, val, ch, &out) != LUMAHIP_OK)
        throw LumaException("LumaQuantizer::quantize: bad quantizer state");
    return out;
}

float LumaQuantizer::dequantize(const float val, const unsigned int ch) const
{
    if (!m_configured)
        throw LumaException("LumaQuantizer::dequantize before setQuantizer");
    float out = 0.0f;
    if (lumahip_dequantize_value_host(m_mapping.data(), m_mapping.size(), (int)m_colorSpace, m_bitdepthColor, val, ch, &out) != LUMAHIP_OK)
        throw LumaException("LumaQuantizer::dequantize: bad quantizer state");
    return out;
}

bool LumaQuantizer::transformColorSpace(LumaFrame *frame, bool toCs, float sc)
{
    if (!frame || !frame->buffer)
        return false;
    if (!m_configured) {
        // a default-constructed reference quantizer is CS_LUV with no table; the transform needs none
        requireContext();
        m_ptf = PTF_LINEAR;
        m_bitdepth = 1;
        m_bitdepthColor = 8;
        m_maxVal = 1;
        m_maxValColor = 255;
        m_mapping.assign(2, 0.0f);
        m_mapping[1] = m_Lmax;
        m_configured = true;
        syncMapping();
    }
    const int rc = lumahip_transform_color_space_host(m_ctx, frame->buffer, frame->width, frame->height, toCs ? 1 : 0, sc);
    if (rc == LUMAHIP_ERR_UNSUPPORTED) {
        fprintf(stderr, toCs ? "Error! Unrecognized color transformation XYZ --> ?\n"
                             : "Error! Unrecognized color transformation ? --> RGB\n");
        return false;
    }
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_ctx));
    return true;
}
