// luma_quantizer.h -- LumaQuantizer with the reference's public interface
// (include/luma/luma_quantizer.h:89-126 there), implemented on the MI355X through the C ABI of
// include/lumahip.h.  Everything that touches frames, planes or arrays runs as HIP kernels.  On the host, as in the
// reference: the construction of the transfer-function table, and the per-value quantize() / dequantize() members
// (scalar calls on the host copy of the table, ~tens of ns like the reference's -- a kernel launch per sample would
// make any caller that loops over them a thousand times slower).
#ifndef LUMA_HIP_QUANTIZER_H
#define LUMA_HIP_QUANTIZER_H

#include <string>
#include <vector>

#include "luma_frame.h"

struct lumahip_ctx;

class LumaQuantizer {
public:
    // numeric values are part of the stream format (attachments 432 / 433)
    enum ptf_t { PTF_PSI, PTF_PQ, PTF_LOG, PTF_JND_HDRVDP, PTF_LINEAR };
    enum colorSpace_t { CS_LUV, CS_RGB, CS_YCBCR, CS_XYZ };

    LumaQuantizer();
    ~LumaQuantizer();
    LumaQuantizer(const LumaQuantizer &) = delete;
    LumaQuantizer &operator=(const LumaQuantizer &) = delete;

    static std::string name(ptf_t ptf);
    static std::string name(colorSpace_t cs);

    // builds the transfer-function table on the host exactly as the reference does and uploads it
    void setQuantizer(ptf_t ptf, unsigned int bitdepth, colorSpace_t cs, unsigned int bitdepthC, float maxLum,
                      float minLum);

    float quantize(const float val, const unsigned int ch) const;
    float dequantize(const float val, const unsigned int ch) const;

    // in place on a host frame, on the GPU; false (plus the reference's stderr line) on an unknown colour space
    bool transformColorSpace(LumaFrame *frame, bool toCs, float sc);

    // The reference hands out its internal table and LumaDecoder::initialize writes attachment 434 through
    // the pointer (src/luma_decoder.cpp:122).  Same here; call syncMapping() after writing so the device
    // copy and the search index follow (the facade's decoder does).
    const float *getMapping() { return m_mapping.data(); }
    void syncMapping();
    unsigned int getSize() { return m_maxVal; }  // = table length - 1, as in the reference
    float getMaxLum() { return m_Lmax; }
    float getMinLum() { return m_Lmin; }

    // ---- additions (not in the reference) ----
    lumahip_ctx *context() { return m_ctx; }      // the C-ABI context all frame-level calls go through
    unsigned int getColorSize() const { return m_maxValColor; }
    colorSpace_t getColorSpace() const { return m_colorSpace; }

private:
    void requireContext();
    lumahip_ctx *m_ctx;
    ptf_t m_ptf;
    colorSpace_t m_colorSpace;
    std::vector<float> m_mapping;
    float m_Lmax, m_Lmin;
    unsigned int m_maxVal, m_maxValColor, m_bitdepth, m_bitdepthColor;
    bool m_configured;
};

#endif
