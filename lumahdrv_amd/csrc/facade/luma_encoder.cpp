// Facade LumaEncoder: parameter handling and metadata as in the reference's initialize(), hot path through
// the C ABI.
#include "../../../include/luma/luma_encoder.h"

#include <cstdio>

#include "../../../include/lumahip.h"

LumaEncoder::LumaEncoder()
    : m_frameCount(0), m_sink(NULL), m_inPlaceCompat(false), m_lastMean(0.0f), m_w(0), m_h(0)
{
}

LumaEncoder::~LumaEncoder() {}

bool LumaEncoder::initialize(const char *outputFile, const unsigned int w, const unsigned int h, bool verbose)
{
    if (!m_sink)
        m_sink = &m_rawWriter;
    // profiles 0/1 are the 8-bit layouts, 2/3 the high-bit-depth ones: follow bitDepth like the reference
    if (m_params.profile > 1 && m_params.bitDepth == 8)
        m_params.profile -= 2;
    if (m_params.profile < 2 && m_params.bitDepth > 8)
        m_params.profile += 2;
    if (w == 0 || h == 0 || (w % 2) != 0 || (h % 2) != 0)
        throw LumaException("Invalid frame size");
    if (m_params.profile > 3)
        throw LumaException("Invalid encoding profile");

    m_sink->open(outputFile, w, h, (int)m_params.profile, m_params.fps);
    m_quant.setQuantizer(m_params.ptf, m_params.ptfBitDepth, m_params.colorSpace, m_params.colorBitDepth, m_params.maxLum,
                         m_params.minLum);

    // stream metadata, ids and payloads as the reference writes them (attachments 430..436); note that the
    // table attachment carries getSize() = maxVal floats, one fewer than the table holds
    const unsigned int ptfBits = m_params.ptfBitDepth, colBits = m_params.colorBitDepth;
    const int ptfId = (int)m_params.ptf, csId = (int)m_params.colorSpace;
    const float range[2] = {m_params.maxLum, m_params.minLum};
    m_sink->addAttachment(430, &ptfBits, sizeof ptfBits, "PTF bit depth");
    m_sink->addAttachment(431, &colBits, sizeof colBits, "Color bit depth");
    m_sink->addAttachment(432, &ptfId, sizeof ptfId, "PTF description");
    m_sink->addAttachment(433, &csId, sizeof csId, "Color space");
    m_sink->addAttachment(434, m_quant.getMapping(), (size_t)m_quant.getSize() * sizeof(float), "PTF");
    m_sink->addAttachment(435, &m_params.preScaling, sizeof(float), "Scaling");
    m_sink->addAttachment(436, range, sizeof range, "Luminance range");
    m_sink->writeAttachments();

    m_rawFrame.allocate(w, h, (int)m_params.profile);
    m_w = w;
    m_h = h;

    const bool sub = (m_params.profile % 2) == 0;
    fprintf(stderr, "Encoding options:\n");
    fprintf(stderr, "-------------------------------------------------------------------\n");
    fprintf(stderr, "Transfer function (PTF):   %s\n", LumaQuantizer::name(m_params.ptf).c_str());
    fprintf(stderr, "Color space:               %s\n", LumaQuantizer::name(m_params.colorSpace).c_str());
    fprintf(stderr, "PTF bit depth:             %d\n", m_params.ptfBitDepth);
    fprintf(stderr, "Color bit depth:           %d\n", m_params.colorBitDepth);
    if (m_params.ptf == LumaQuantizer::PTF_PQ || m_params.ptf == LumaQuantizer::PTF_LOG || m_params.ptf == LumaQuantizer::PTF_LINEAR)
        fprintf(stderr, "Encoding luminance range:  %.4f-%.2f\n", m_quant.getMinLum(), m_quant.getMaxLum());
    fprintf(stderr, "Encoding profile:          %d (4%d%d)\n", m_params.profile, sub ? 2 : 4, sub ? 2 : 4);
    fprintf(stderr, "Encoding bit depth:        %d\n", (m_params.bitDepth == 8 || m_params.profile < 2) ? 8 : (m_params.bitDepth == 10 ? 10 : 12));
    fprintf(stderr, "Transform:                 HIP / gfx950 (lumahip ABI %d)\n", lumahip_abi_version());
    fprintf(stderr, "Output:                    %s\n", outputFile);
    fprintf(stderr, "-------------------------------------------------------------------\n\n");
    (void)verbose;
    m_frameCount = 0;
    m_initialized = true;
    return true;
}

void LumaEncoder::warnMean(float avg)
{
    m_lastMean = avg;
    if (avg <= 1.0f)
        fprintf(stderr, "\n\tWarning! Mean luminance is %f cd/m2. Is input calibrated to physical units? \n", avg);
}

void LumaEncoder::setChannels(LumaFrame *frame)
{
    if (!m_initialized)
        throw LumaException("Encoder not initialized");
    if (frame->width != m_w || frame->height != m_h || frame->channels < 3)
        throw LumaException("Frame size differs from the size the encoder was initialized with");
    LumaPlanes &im = m_rawFrame.image();
    float avg = 0.0f;
    const int rc = lumahip_pack_frame_host(m_quant.context(), frame->buffer, m_w, m_h, (int)m_params.profile, im.planes,
                                           im.stride, &avg);
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_quant.context()));
    warnMean(avg);
}

bool LumaEncoder::encode(LumaFrame *frame)
{
    if (!m_initialized)
        throw LumaException("Encoder not initialized");
    if (frame->width != m_w || frame->height != m_h || frame->channels < 3)
        throw LumaException("Frame size differs from the size the encoder was initialized with");
    LumaPlanes &im = m_rawFrame.image();
    float avg = 0.0f;
    const int rc = lumahip_encode_frame_host(m_quant.context(), frame->buffer, m_w, m_h, m_params.preScaling,
                                             (int)m_params.profile, im.planes, im.stride, &avg,
                                             m_inPlaceCompat ? frame->buffer : NULL);
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_quant.context()));
    warnMean(avg);
    return run();
}

bool LumaEncoder::run()
{
    if (!m_initialized)
        return false;
    m_frameCount++;
    if (!m_sink->addFrame(m_rawFrame.image()))
        fprintf(stderr, "Failed to encode frame\n");
    return true;
}

void LumaEncoder::finish()
{
    if (m_sink)
        m_sink->close();
}
