// What LumaEncoder::initialize and LumaBatchEncoder::initialize share: the parameter fix-ups, the metadata attachments and
// the option banner of the reference's LumaEncoder::initialize (src/luma_encoder.cpp:63-193 there).
#ifndef LUMA_HIP_ENCODER_COMMON_H
#define LUMA_HIP_ENCODER_COMMON_H

#include <cstdio>

#include "../../../include/luma/luma_encoder.h"
#include "../../../include/lumahip.h"

namespace luma_detail {

inline void checkGeometry(LumaEncoderParams &p, unsigned int w, unsigned int h)
{
    // profiles 0/1 are the 8-bit layouts, 2/3 the high-bit-depth ones: follow bitDepth like the reference
    if (p.profile > 1 && p.bitDepth == 8)
        p.profile -= 2;
    if (p.profile < 2 && p.bitDepth > 8)
        p.profile += 2;
    if (w == 0 || h == 0 || (w % 2) != 0 || (h % 2) != 0)
        throw LumaException("Invalid frame size");
    if (p.profile > 3)
        throw LumaException("Invalid encoding profile");
}

// stream metadata, ids and payloads as the reference writes them (attachments 430..436); note that the table attachment
// carries getSize() = maxVal floats, one fewer than the table holds
inline void writeAttachments(LumaPlaneSink *sink, const LumaEncoderParams &p, const float *mapping, unsigned int size)
{
    const unsigned int ptfBits = p.ptfBitDepth, colBits = p.colorBitDepth;
    const int ptfId = (int)p.ptf, csId = (int)p.colorSpace;
    const float range[2] = {p.maxLum, p.minLum};
    sink->addAttachment(430, &ptfBits, sizeof ptfBits, "PTF bit depth");
    sink->addAttachment(431, &colBits, sizeof colBits, "Color bit depth");
    sink->addAttachment(432, &ptfId, sizeof ptfId, "PTF description");
    sink->addAttachment(433, &csId, sizeof csId, "Color space");
    sink->addAttachment(434, mapping, (size_t)size * sizeof(float), "PTF");
    sink->addAttachment(435, &p.preScaling, sizeof(float), "Scaling");
    sink->addAttachment(436, range, sizeof range, "Luminance range");
    sink->writeAttachments();
}

inline void printBanner(const LumaEncoderParams &p, const char *outputFile, const char *transform)
{
    const bool sub = (p.profile % 2) == 0;
    fprintf(stderr, "Encoding options:\n");
    fprintf(stderr, "-------------------------------------------------------------------\n");
    fprintf(stderr, "Transfer function (PTF):   %s\n", LumaQuantizer::name(p.ptf).c_str());
    fprintf(stderr, "Color space:               %s\n", LumaQuantizer::name(p.colorSpace).c_str());
    fprintf(stderr, "PTF bit depth:             %d\n", p.ptfBitDepth);
    fprintf(stderr, "Color bit depth:           %d\n", p.colorBitDepth);
    if (p.ptf == LumaQuantizer::PTF_PQ || p.ptf == LumaQuantizer::PTF_LOG || p.ptf == LumaQuantizer::PTF_LINEAR)
        fprintf(stderr, "Encoding luminance range:  %.4f-%.2f\n", p.minLum, p.maxLum);
    fprintf(stderr, "Encoding profile:          %d (4%d%d)\n", p.profile, sub ? 2 : 4, sub ? 2 : 4);
    fprintf(stderr, "Encoding bit depth:        %d\n", (p.bitDepth == 8 || p.profile < 2) ? 8 : (p.bitDepth == 10 ? 10 : 12));
    fprintf(stderr, "Transform:                 %s (lumahip ABI %d)\n", transform, lumahip_abi_version());
    fprintf(stderr, "Output:                    %s\n", outputFile);
    fprintf(stderr, "-------------------------------------------------------------------\n\n");
}

inline void warnMean(float avg)
{
    if (avg <= 1.0f)
        fprintf(stderr, "\n\tWarning! Mean luminance is %f cd/m2. Is input calibrated to physical units? \n", avg);
}

}  // namespace luma_detail

#endif
