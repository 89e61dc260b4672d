// luma_decoder.h -- LumaDecoder with the reference's interface (include/luma/luma_decoder.h:60-175 there)
// for the hot path: LumaDecoder(file) / initialize / run / decode / getBuffer / getParams / getQuantizer.
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

struct LumaDecoderParams : LumaDecoderParamsBase {
    LumaDecoderParams() : ptfBitDepth(11), colorBitDepth(8), highBitDepth(true), stride(NULL), profile(2)
    {
        for (int i = 0; i < 3; i++)
            width[i] = height[i] = 0;
    }
    unsigned int ptfBitDepth, colorBitDepth;
    bool highBitDepth;
    const int *stride;
    int profile, width[3], height[3];
};

class LumaDecoderBase {
public:
    LumaDecoderBase() : m_initialized(false), m_input(NULL) {}
    virtual ~LumaDecoderBase() {}
    virtual bool initialize(const char *inputFile, bool verbose = 0) = 0;
    virtual bool run() = 0;
    virtual LumaFrame *decode() = 0;
    LumaQuantizer *getQuantizer() { return &m_quant; }
    LumaFrame *getFrame() { return &m_frame; }
    bool initialized() { return m_initialized; }

protected:
    bool m_initialized;
    const char *m_input;
    LumaQuantizer m_quant;
    LumaFrame m_frame;
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
    void seekToTime(float tm, bool absolute = false);

    unsigned char **getBuffer() { return m_planePtrs; }
    LumaDecoderParams getParams() { return m_params; }
    void setParams(LumaDecoderParams params) { m_params = params; }

    // the reference returns its MkvInterface here (luma_decoder.h:113 there; lumaplay.cpp:200,443 ask it for
    // getDuration() / getFrameDuration()); this build's upstream stage is a LumaPlaneSource with the same two queries
    LumaPlaneSource *getReader() { return m_source; }

    // ---- additions ----
    void setSource(LumaPlaneSource *src) { m_source = src; }  // not owned; default: raw plane stream
    // Pipelined mode (opt-in): decode() reads and uploads the planes of frame i+1 and queues its kernel BEFORE it completes
    // frame i and returns it, so the download of a frame (12 B/pixel, the heavy direction here) runs under the next frame's
    // read, upload and kernel.  Same frames in the same order; the returned frame is valid until the next decode(), as always.
    // getFrame() is then not meaningful between calls, run() / getBuffer() refer to the frame read AHEAD, and seekToTime()
    // first drops what is in flight.
    void setPipelined(bool on) { m_pipelined = on; }
    bool pipelined() const { return m_pipelined; }

private:
    LumaDecoderParams m_params;
    const LumaPlanes *m_vpxFrame;
    unsigned char *m_planePtrs[3];
    bool m_firstFrame;
    LumaPlaneSource *m_source;
    LumaRawStreamReader m_rawReader;
    float m_time;
    bool pushNext();               // pipelined mode: read the next frame's planes and start it; false at the end of the stream
    void dropInFlight();
    LumaFrame m_frame2;            // pipelined mode: decoded frames alternate between m_frame and this one
    unsigned int m_pushed;
    bool m_pipelined;
};

#endif
