// Facade LumaEncoder: parameter handling and metadata as in the reference's initialize(), hot path through
// the C ABI.
#include "../../../include/luma/luma_encoder.h"

#include <cstdio>

#include "encoder_common.h"

LumaEncoder::LumaEncoder()
    : m_frameCount(0), m_sink(NULL), m_inPlaceCompat(false), m_lastMean(0.0f), m_w(0), m_h(0)
{
}

LumaEncoder::~LumaEncoder() {}

bool LumaEncoder::initialize(const char *outputFile, const unsigned int w, const unsigned int h, bool verbose)
{
    if (!m_sink)
        m_sink = &m_rawWriter;
    luma_detail::checkGeometry(m_params, w, h);

    m_sink->open(outputFile, w, h, (int)m_params.profile, m_params.fps);
    m_quant.setQuantizer(m_params.ptf, m_params.ptfBitDepth, m_params.colorSpace, m_params.colorBitDepth, m_params.maxLum,
                         m_params.minLum);

    luma_detail::writeAttachments(m_sink, m_params, m_quant.getMapping(), m_quant.getSize());

    m_rawFrame.allocate(w, h, (int)m_params.profile);
    m_w = w;
    m_h = h;

    LumaEncoderParams shown = m_params;
    shown.minLum = m_quant.getMinLum();
    shown.maxLum = m_quant.getMaxLum();
    luma_detail::printBanner(shown, outputFile, "HIP / gfx950");
    (void)verbose;
    m_frameCount = 0;
    m_initialized = true;
    return true;
}

bool LumaEncoder::initialize(const char *outputFile, const unsigned int w, const unsigned int h, const float ma, const float mi,
                             bool verbose)
{
    // what the reference's base class does: the container is opened with the range, nothing else changes
    if (!m_sink)
        m_sink = &m_rawWriter;
    m_sink->setLuminanceRange(ma, mi);
    m_sink->open(outputFile, w, h, (int)m_params.profile, m_params.fps);
    (void)verbose;
    return true;
}

void LumaEncoder::warnMean(float avg)
{
    m_lastMean = avg;
    luma_detail::warnMean(avg);
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
