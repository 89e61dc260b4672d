// Facade LumaEncoder: parameter handling and metadata as in the reference's initialize(), hot path through
// the C ABI.
#include "../../../include/luma/luma_encoder.h"

#include <cstdio>

#include "encoder_common.h"

LumaEncoder::LumaEncoder()
    : m_delivered(NULL), m_pushed(0), m_pipelined(false), m_frameCount(0), m_sink(NULL), m_inPlaceCompat(false), m_lastMean(0.0f),
      m_w(0), m_h(0)
{
}

LumaEncoder::~LumaEncoder()
{
    // frames still in flight write into m_rawFrame / m_rawFrame2: complete them before the buffers go away
    if (m_initialized && m_quant.context())
        while (lumahip_encode_stream_pending(m_quant.context()) > 0)
            if (lumahip_encode_stream_pop(m_quant.context(), NULL) != LUMAHIP_OK)
                break;
}

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
    if (m_pipelined) {
        if (m_inPlaceCompat)
            throw LumaException("setPipelined(true) cannot be combined with setInPlaceCompat(true)");
        m_rawFrame2.allocate(w, h, (int)m_params.profile);
    }
    m_delivered = NULL;
    m_pushed = 0;
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
    if (m_pipelined) {
        // frame i+1 goes up and is launched; only then is frame i completed and handed on (luma_encoder.h: setPipelined)
        LumaPlaneBuffer &buf = (m_pushed & 1) ? m_rawFrame2 : m_rawFrame;
        LumaPlanes &pim = buf.image();
        const int prc = lumahip_encode_stream_push(m_quant.context(), frame->buffer, m_w, m_h, m_params.preScaling,
                                                   (int)m_params.profile, pim.planes, pim.stride);
        if (prc != LUMAHIP_OK)
            throw LumaException(lumahip_last_error(m_quant.context()));
        m_pushed++;
        if (lumahip_encode_stream_pending(m_quant.context()) >= 2)
            return deliverOldest();
        return true;
    }
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

bool LumaEncoder::deliverOldest()
{
    const int pending = lumahip_encode_stream_pending(m_quant.context());
    if (pending <= 0)
        return false;
    const unsigned int seq = m_pushed - (unsigned int)pending;       // sequence number of the oldest frame in flight
    float avg = 0.0f;
    if (lumahip_encode_stream_pop(m_quant.context(), &avg) != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_quant.context()));
    warnMean(avg);
    m_delivered = (seq & 1) ? &m_rawFrame2 : &m_rawFrame;
    m_frameCount++;
    if (!m_sink->addFrame(m_delivered->image()))
        fprintf(stderr, "Failed to encode frame\n");
    return true;
}

void LumaEncoder::finish()
{
    if (m_initialized && m_pipelined)
        while (lumahip_encode_stream_pending(m_quant.context()) > 0)
            deliverOldest();
    if (m_sink)
        m_sink->close();
}
