// luma_decoder.h -- LumaDecoder with the reference's interface (include/luma/luma_decoder.h:60-175 there)
// for the hot path: LumaDecoderBase (constructor, seekToTime, getQuantizer, getReader, getFrame on the base, as there) and
// LumaDecoder(file) / initialize / run / decode / getBuffer / getParams.
//
// decode() = run() (fetch the next frame's Y/U/V planes from the upstream stage) + ONE fused HIP kernel
// (unpack + LUT / colour dequantize + 2x2 chroma replicate + inverse colour transform) into the decoder's
// own LumaFrame, returned by pointer and valid until the next decode(), as in the reference.  Upstream is
// a LumaPlaneSource (luma_planes.h); the reference's MKV demux + vpx_codec_decode attach there.
#ifndef LUMA_HIP_DECODER_H
#define LUMA_HIP_DECODER_H

#include "luma_exception.h"
#include "luma_frame.h"
#include "luma_planes.h"
#include "luma_quantizer.h"

struct LumaDecoderParamsBase {
    LumaDecoderParamsBase()
        : ptf(LumaQuantizer::PTF_PSI), colorSpace(LumaQuantizer::CS_LUV), preScaling(1.0f), minLum(0.005f), maxLum(1e4f)
    {
    }
    LumaQuantizer::ptf_t ptf;
    LumaQuantizer::colorSpace_t colorSpace;
    float preScaling, minLum, maxLum;
};

// The reference's base class (include/luma/luma_decoder.h:81-104 there): constructor (inputFile, verbose), the pure
// virtual initialize / run / decode, and -- non-virtual, on the base -- seekToTime, getQuantizer, getReader, getFrame,
// initialized.  The reference's base owns an MkvInterface (m_reader); this one owns the upstream stage it stands for here, a
// LumaPlaneSource (default: the raw plane stream reader), which answers the two questions the reference's callers put to
// getReader() (lumaplay.cpp:200,443: getDuration / getFrameDuration).
class LumaDecoderBase {
public:
    LumaDecoderBase(const char *inputFile = NULL, bool verbose = 0)
        : m_initialized(false), m_input(inputFile), m_source(NULL), m_time(0.0f), m_firstFrame(false)
    {
        (void)verbose;
    }
    virtual ~LumaDecoderBase() {}
    virtual bool initialize(const char *inputFile, bool verbose = 0) = 0;
    virtual bool run() = 0;
    void seekToTime(float tm, bool absolute = false);
    virtual LumaFrame *decode() = 0;
    LumaQuantizer *getQuantizer() { return &m_quant; }
    LumaPlaneSource *getReader() { return m_source ? m_source : &m_rawReader; }
    LumaFrame *getFrame() { return &m_frame; }
    bool initialized() { return m_initialized; }

    // ---- addition ----
    void setSource(LumaPlaneSource *src) { m_source = src; }  // not owned; default: the raw plane stream reader

protected:
    virtual void beforeSeek() {}   // a decoder with frames in flight drops them here (LumaDecoder's pipelined mode)
    bool m_initialized;
    const char *m_input;
    LumaQuantizer m_quant;
    LumaFrame m_frame;
    LumaPlaneSource *m_source;
    LumaRawStreamReader m_rawReader;
    float m_time;
    bool m_firstFrame;             // frame 0 was fetched by initialize() and is what the next run() returns
};

struct LumaDecoderParams : LumaDecoderParamsBase {
    LumaDecoderParams() : ptfBitDepth(11), colorBitDepth(8), highBitDepth(true), stride(NULL), profile(2)
    {
        for (int i = 0; i < 3; i++)
            width[i] = height[i] = 0;
    }
    unsigned int ptfBitDepth, colorBitDepth;
    bool highBitDepth;
    int *stride, profile, width[3], height[3];   // stride: the decoder's own copy of the three plane strides (bytes)
};

class LumaDecoder : public LumaDecoderBase {
public:
    LumaDecoder(const char *inputFile = NULL, bool verbose = 0);
    ~LumaDecoder();

    // throws LumaException("Failed to locate Luma HDRv meta data in '<file>'") when the stream lacks the
    // attachments 430..434, like the reference
    bool initialize(const char *inputFile, bool verbose = 0);
    bool run();
    LumaFrame *decode();  // NULL at end of stream

    unsigned char **getBuffer() { return m_planePtrs; }
    LumaDecoderParams getParams() { return m_params; }
    void setParams(LumaDecoderParams params) { m_params = params; }

    // ---- additions ----
    // Pipelined mode (opt-in): decode() reads and uploads the planes of frame i+1 and queues its kernel BEFORE it completes
    // frame i and returns it, so the download of a frame (12 B/pixel, the heavy direction here) runs under the next frame's
    // read, upload and kernel.  Same frames in the same order; the returned frame is valid until the next decode(), as always.
    // getFrame() is then not meaningful between calls, run() / getBuffer() refer to the frame read AHEAD, and seekToTime()
    // first drops what is in flight.
    void setPipelined(bool on) { m_pipelined = on; }
    bool pipelined() const { return m_pipelined; }

protected:
    void beforeSeek() { dropInFlight(); }

private:
    LumaDecoderParams m_params;
    const LumaPlanes *m_vpxFrame;
    unsigned char *m_planePtrs[3];
    int m_stride[3];               // what m_params.stride points at
    bool pushNext();               // pipelined mode: read the next frame's planes and start it; false at the end of the stream
    void dropInFlight();
    LumaFrame m_frame2;            // pipelined mode: decoded frames alternate between m_frame and this one
    unsigned int m_pushed;
    bool m_pipelined;
};

#endif
