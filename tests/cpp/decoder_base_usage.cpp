// tests/cpp/decoder_base_usage.cpp -- compile-only: code that holds a LumaDecoderBase* and uses nothing but what the reference
// documents on the base class and the two parameter structs (include/luma/luma_decoder.h:60-120 of the reference).
// tests/test_host_side.py compiles this same file against the reference's headers and against include/luma/ of this repo.
#include <cstddef>

#include <luma_decoder.h>

struct FrameCounter : LumaDecoderBase {
    // the reference's base constructor: (inputFile = NULL, verbose = 0)
    FrameCounter(const char *inputFile) : LumaDecoderBase(inputFile, true), n(0) { m_initialized = false; }
    bool initialize(const char *inputFile, bool verbose = 0)
    {
        m_input = inputFile;
        m_initialized = inputFile != NULL && !verbose;
        return m_initialized;
    }
    bool run() { return ++n < 3; }
    LumaFrame *decode() { return run() ? &m_frame : NULL; }
    int n;
};

static float drive(LumaDecoderBase *d)
{
    if (!d->initialized())
        d->initialize("in.mkv");
    d->seekToTime(1.5f);          // relative
    d->seekToTime(0.0f, true);    // absolute
    float acc = 0.0f;
    while (LumaFrame *f = d->decode())
        acc += (float)(f->width * f->height * f->channels);
    LumaQuantizer *q = d->getQuantizer();
    acc += q->getMaxLum() + q->getMinLum() + (float)q->getSize();
    acc += d->getReader()->getDuration() + d->getReader()->getFrameDuration();   // lumaplay.cpp:200,443
    acc += (float)d->getFrame()->width;
    return acc;
}

int main()
{
    LumaDecoderBase *none = NULL;
    LumaDecoderParamsBase pb;
    pb.ptf = LumaQuantizer::PTF_PQ;
    pb.colorSpace = LumaQuantizer::CS_LUV;
    pb.preScaling = pb.minLum = pb.maxLum = 1.0f;
    LumaDecoderParams p;
    int strides[3] = {0, 0, 0};
    p.stride = strides;           // int *, as in the reference
    int *s = p.stride;
    p.profile = p.width[0] = p.height[2] = (int)(p.ptfBitDepth + p.colorBitDepth) + (p.highBitDepth ? s[0] : 1);
    return (none != NULL && drive(none) > 0.0f) ? 1 : 0;
}
