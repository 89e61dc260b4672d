// luma_batch_encoder.h -- LumaEncoder for callers that hold SEVERAL frames: the reference's frame loop
// `for (...) encoder.encode(&frame)` (lumaenc.cpp:205-243 there) spread over every GPU of the node.
//
// encode(frames, n) splits the n frames into contiguous blocks, one per shard (a shard = one lumahip context on one GPU with
// its own host thread; lumahip.h "many GPUs in one process"), runs the fused colour transform + quantize + pack kernels on
// all GPUs at once, and then hands the filled Y/U/V planes to the sink IN FRAME ORDER -- the VP9 stage downstream is
// sequential (src/luma_encoder.cpp:229-257), and block sharding leaves every shard's output already ordered.  The
// transfer-function table is built once on the host exactly as LumaQuantizer::setQuantizer builds it and reaches the GPUs
// by one RCCL broadcast.  Parameters, metadata attachments 430-436, messages and error conventions are LumaEncoder's.
#ifndef LUMA_HIP_BATCH_ENCODER_H
#define LUMA_HIP_BATCH_ENCODER_H

#include <vector>

#include "luma_encoder.h"

struct lumahip_multi;

class LumaBatchEncoder {
public:
    LumaBatchEncoder();
    ~LumaBatchEncoder();
    LumaBatchEncoder(const LumaBatchEncoder &) = delete;
    LumaBatchEncoder &operator=(const LumaBatchEncoder &) = delete;

    LumaEncoderParams getParams() { return m_params; }
    void setParams(LumaEncoderParams params) { m_params = params; }
    void setSink(LumaPlaneSink *sink) { m_sink = sink; }  // not owned; default: raw plane stream

    // devices == NULL: one shard per visible GPU (nshards > 0: that many shards, round-robin over the GPUs); otherwise
    // nshards entries of `devices` (a GPU may appear more than once).  Throws LumaException like LumaEncoder::initialize.
    bool initialize(const char *outputFile, const unsigned int w, const unsigned int h, bool verbose = 0,
                    const int *devices = NULL, int nshards = 0);
    bool initialized() const { return m_initialized; }

    // n frames of the size given to initialize(), in stream order; returns after every frame has reached the sink
    bool encode(LumaFrame *const *frames, unsigned int n);
    bool encode(LumaFrame *frame) { return encode(&frame, 1); }
    void finish();

    unsigned int shards() const;
    bool quantizerCameOverRccl() const;            // true once the table has been broadcast with RCCL
    unsigned int framesEncoded() const { return m_frameCount; }
    float lastMeanLuminance() const { return m_lastMean; }
    const float *getMapping() const { return m_mapping.data(); }
    unsigned int getSize() const { return m_maxVal; }   // = table length - 1, as LumaQuantizer::getSize()

private:
    lumahip_multi *m_multi;
    LumaEncoderParams m_params;
    LumaPlaneSink *m_sink;
    LumaRawStreamWriter m_rawWriter;
    std::vector<LumaPlaneBuffer> m_planes;   // one per frame of the largest batch seen so far
    std::vector<float> m_mapping;
    unsigned int m_maxVal;
    unsigned int m_frameCount;
    unsigned int m_w, m_h;
    float m_lastMean;
    bool m_initialized;
};

#endif
