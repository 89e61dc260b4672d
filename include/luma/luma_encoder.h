// luma_encoder.h -- LumaEncoder with the reference's interface (include/luma/luma_encoder.h:59-168 there)
// for the hot path: setParams / initialize / encode(LumaFrame*) / setChannels / run / finish.
//
// encode() = ONE fused HIP kernel (colour transform + PTF-LUT quantize + chroma subsample + plane pack)
// that fills the same Y/U/V plane memory the reference's m_rawFrame holds, followed by run(), which hands
// the planes to the downstream stage.  Downstream is a LumaPlaneSink (luma_planes.h): the VP9 + Matroska
// stages of the reference are out of scope and unchanged; a sink wrapping vpx_codec_encode + MkvInterface
// attaches there.  Without one, a raw plane stream is written.
#ifndef LUMA_HIP_ENCODER_H
#define LUMA_HIP_ENCODER_H

#include "luma_exception.h"
#include "luma_frame.h"
#include "luma_planes.h"
#include "luma_quantizer.h"

struct LumaEncoderParamsBase {
    LumaEncoderParamsBase()
        : quantizerScale(2), ptfBitDepth(11), colorBitDepth(8), preScaling(1.0f), fps(25.0f), minLum(0.005f),
          maxLum(1e4f), ptf(LumaQuantizer::PTF_PQ), colorSpace(LumaQuantizer::CS_LUV)
    {
    }
    unsigned int quantizerScale, ptfBitDepth, colorBitDepth;
    float preScaling, fps, minLum, maxLum;
    LumaQuantizer::ptf_t ptf;
    LumaQuantizer::colorSpace_t colorSpace;
};

struct LumaEncoderParams : LumaEncoderParamsBase {
    LumaEncoderParams() : bitrate(10000), profile(2), keyframeInterval(0), bitDepth(12), lossLess(false) {}
    unsigned int bitrate, profile, keyframeInterval, bitDepth;
    bool lossLess;
};

class LumaEncoderBase {
public:
    LumaEncoderBase() : m_initialized(false) {}
    virtual ~LumaEncoderBase() {}
    // The reference's base class opens ITS container with a luminance range here (luma_encoder.h:83-91 there:
    // m_writer.openWrite(outputFile, w, h, ma, mi)) and nothing else -- no parameters, no quantizer, m_initialized stays false;
    // LumaEncoder's own four-argument initialize() hides it, as it does in the reference.  Same here: the derived class
    // forwards the range to its sink and opens it.
    virtual bool initialize(const char *outputFile, const unsigned int w, const unsigned int h, const float ma, const float mi,
                            bool verbose = 0)
    {
        (void)outputFile, (void)w, (void)h, (void)ma, (void)mi, (void)verbose;
        return true;
    }
    virtual bool run() = 0;
    virtual void setChannels(LumaFrame *frame) = 0;
    virtual bool encode(LumaFrame *frame) = 0;
    virtual void finish() = 0;
    bool initialized() { return m_initialized; }

protected:
    bool m_initialized;
    LumaQuantizer m_quant;
};

class LumaEncoder : public LumaEncoderBase {
public:
    LumaEncoder();
    ~LumaEncoder();

    // throws LumaException("Invalid frame size") for zero or odd dimensions, like the reference
    bool initialize(const char *outputFile, const unsigned int w, const unsigned int h, bool verbose = 0);
    // LumaEncoderBase::initialize(outputFile, w, h, ma, mi, verbose): opens the sink with the container-level luminance range
    // only (see the base class); reachable through a LumaEncoderBase reference, hidden here by the overload above
    bool initialize(const char *outputFile, const unsigned int w, const unsigned int h, const float ma, const float mi,
                    bool verbose) override;
    bool run();
    // quantize + pack an ALREADY colour-transformed frame (what the reference's setChannels expects)
    void setChannels(LumaFrame *frame);
    // colour transform + quantize + pack, fused; then run().  The reference transforms `frame` in place as
    // a side effect; that write-back costs a 12 B/pixel D2H copy and both reference callers discard the
    // frame, so it happens only after setInPlaceCompat(true).
    bool encode(LumaFrame *frame);
    void finish();

    LumaEncoderParams getParams() { return m_params; }
    void setParams(LumaEncoderParams params) { m_params = params; }

    // ---- additions ----
    void setSink(LumaPlaneSink *sink) { m_sink = sink; }     // not owned; default: raw plane stream
    void setInPlaceCompat(bool on) { m_inPlaceCompat = on; }
    // Pipelined mode (opt-in, before initialize): encode(frame i+1) uploads and launches frame i+1 and only THEN completes frame i
    // and hands its planes to the sink -- one frame of latency, finish() delivers the last one.  The tail of a frame (kernel,
    // download of the planes, copy out of the staging buffers) then runs under the next frame's upload, which a synchronous
    // call cannot do: pageable 4K LumaFrames 3.2 -> 4 Gpixel/s (PCIe).  `frame` may be reused as soon as encode() returns, as in
    // the default mode; the sink sees the same planes in the same order.  getRawFrame() / lastMeanLuminance() and the mean
    // luminance warning refer to the frame most recently DELIVERED.  Not combinable with setInPlaceCompat(true).
    void setPipelined(bool on) { m_pipelined = on; }
    bool pipelined() const { return m_pipelined; }
    const LumaPlanes &getRawFrame() const { return m_delivered ? m_delivered->image() : m_rawFrame.image(); }  // the filled Y/U/V planes
    LumaQuantizer *getQuantizer() { return &m_quant; }
    float lastMeanLuminance() const { return m_lastMean; }

private:
    void warnMean(float avg);
    bool deliverOldest();          // pipelined mode: complete the oldest frame in flight and hand it to the sink
    LumaPlaneBuffer m_rawFrame;
    LumaPlaneBuffer m_rawFrame2;   // pipelined mode: the planes of two frames alternate between m_rawFrame and this one
    const LumaPlaneBuffer *m_delivered;
    unsigned int m_pushed;
    bool m_pipelined;
    unsigned int m_frameCount;
    LumaEncoderParams m_params;
    LumaPlaneSink *m_sink;
    LumaRawStreamWriter m_rawWriter;
    bool m_inPlaceCompat;
    float m_lastMean;
    unsigned int m_w, m_h;
};

#endif
