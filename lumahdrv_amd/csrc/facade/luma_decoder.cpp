// Facade LumaDecoder: metadata handling as in the reference's initialize(), hot path through the C ABI.
#include "../../../include/luma/luma_decoder.h"

#include <cstdio>
#include <cstring>

#include "../../../include/lumahip.h"

LumaDecoder::LumaDecoder(const char *inputFile, bool verbose)
    : LumaDecoderBase(inputFile, verbose), m_vpxFrame(NULL), m_pushed(0), m_pipelined(false)
{
    m_planePtrs[0] = m_planePtrs[1] = m_planePtrs[2] = NULL;
    m_stride[0] = m_stride[1] = m_stride[2] = 0;
    if (inputFile != NULL)
        initialize(inputFile, verbose);
}

LumaDecoder::~LumaDecoder()
{
    // frames still in flight are being written into m_frame / m_frame2: complete them before the buffers go away
    if (m_quant.context())
        dropInFlight();
}

void LumaDecoder::dropInFlight()
{
    while (lumahip_decode_stream_pending(m_quant.context()) > 0)
        if (lumahip_decode_stream_pop(m_quant.context()) != LUMAHIP_OK)
            break;
}

static void allocateFrame(LumaFrame &f, unsigned int w, unsigned int h)
{
    if (f.width)
        return;
    f.width = w;
    f.height = h;
    f.channels = 3;
    bool ok = false;
    try {
        ok = f.init();
    } catch (const std::exception &) {   // std::bad_alloc: callers catch LumaException, as the reference's apps do
    }
    if (!ok) {
        f.width = f.height = 0;
        throw LumaException("Cannot allocate memory for the decoded frame");
    }
}

bool LumaDecoder::pushNext()
{
    if (!run())
        return false;
    LumaFrame &dst = (m_pushed & 1) ? m_frame2 : m_frame;
    allocateFrame(dst, m_vpxFrame->d_w, m_vpxFrame->d_h);
    const unsigned char *pl[3] = {m_vpxFrame->planes[0], m_vpxFrame->planes[1], m_vpxFrame->planes[2]};
    const int rc = lumahip_decode_stream_push(m_quant.context(), pl, m_vpxFrame->stride, m_vpxFrame->d_w, m_vpxFrame->d_h,
                                              m_vpxFrame->profile(), m_params.preScaling, dst.buffer);
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_quant.context()));
    m_pushed++;
    return true;
}

bool LumaDecoder::initialize(const char *inputFile, bool verbose)
{
    if (inputFile == NULL)
        return false;
    m_input = inputFile;
    if (!m_source)
        m_source = &m_rawReader;
    m_source->open(inputFile);
    (void)verbose;

    // collect the metadata attachments; 430..434 are mandatory, 435 / 436 keep their defaults when absent
    unsigned char *buffer = NULL;
    unsigned int id = 0, size = 0, mappingBytes = 0;
    const float *mapping = NULL;
    bool have430 = false, have431 = false, have432 = false, have433 = false;
    for (unsigned int idx = 0; m_source->getAttachment(idx, &buffer, id, size); idx++) {
        const unsigned int need = (id == 436) ? 8u : (id == 434 ? 0u : 4u);
        if (id >= 430 && id <= 436 && size < need)
            throw LumaException("Malformed Luma HDRv meta data (attachment too short)");
        switch (id) {
        case 430: memcpy(&m_params.ptfBitDepth, buffer, sizeof(unsigned int)); have430 = true; break;
        case 431: memcpy(&m_params.colorBitDepth, buffer, sizeof(unsigned int)); have431 = true; break;
        case 432: { int v; memcpy(&v, buffer, sizeof v); m_params.ptf = (LumaQuantizer::ptf_t)v; have432 = true; break; }
        case 433: { int v; memcpy(&v, buffer, sizeof v); m_params.colorSpace = (LumaQuantizer::colorSpace_t)v; have433 = true; break; }
        case 434: mapping = (const float *)buffer; mappingBytes = size; break;
        case 435: memcpy(&m_params.preScaling, buffer, sizeof(float)); break;
        case 436: memcpy(&m_params.maxLum, buffer, sizeof(float)); memcpy(&m_params.minLum, buffer + sizeof(float), sizeof(float)); break;
        default: break;
        }
    }
    if (!(have430 && have431 && have432 && have433 && mapping)) {
        std::string msg = "Failed to locate Luma HDRv meta data in '" + std::string(inputFile) + "'";
        throw LumaException(msg.c_str());
    }

    // Rebuild the table from the parameters, then lay the attachment's floats over its head: the attachment
    // holds getSize() = maxVal entries, so the last entry always comes from this side's recomputation.
    m_quant.setQuantizer(m_params.ptf, m_params.ptfBitDepth, m_params.colorSpace, m_params.colorBitDepth, m_params.maxLum,
                         m_params.minLum);
    const size_t tableBytes = ((size_t)m_quant.getSize() + 1) * sizeof(float);
    memcpy(const_cast<float *>(m_quant.getMapping()), mapping, mappingBytes < tableBytes ? mappingBytes : tableBytes);
    m_quant.syncMapping();

    fprintf(stderr, "\nDecoding options:\n");
    fprintf(stderr, "-------------------------------------------------------------------\n");
    fprintf(stderr, "Transfer function (PTF):   %s\n", LumaQuantizer::name(m_params.ptf).c_str());
    if (m_params.ptf == LumaQuantizer::PTF_PQ || m_params.ptf == LumaQuantizer::PTF_LOG || m_params.ptf == LumaQuantizer::PTF_LINEAR)
        fprintf(stderr, "Encoding luminance range:  %.4f-%.2f\n", m_quant.getMinLum(), m_quant.getMaxLum());
    fprintf(stderr, "Color space:               %s\n", LumaQuantizer::name(m_params.colorSpace).c_str());
    fprintf(stderr, "PTF bit depth:             %d\n", m_params.ptfBitDepth);
    fprintf(stderr, "Color bit depth:           %d\n", m_params.colorBitDepth);
    fprintf(stderr, "Transform:                 HIP / gfx950 (lumahip ABI %d)\n", lumahip_abi_version());
    fprintf(stderr, "-------------------------------------------------------------------\n\n");

    // pre-fetch the first frame to learn the plane geometry (the reference pre-decodes frame 0 the same way)
    m_firstFrame = false;
    m_initialized = true;
    if (!run()) {
        m_initialized = false;
        return false;
    }
    m_firstFrame = true;

    m_params.highBitDepth = m_vpxFrame->highBitDepth;
    for (int p = 0; p < 3; p++)
        m_stride[p] = m_vpxFrame->stride[p];
    m_params.stride = m_stride;
    m_params.profile = m_vpxFrame->profile();
    for (int p = 0; p < 3; p++) {
        m_params.width[p] = (int)m_vpxFrame->planeWidth(p);
        m_params.height[p] = (int)m_vpxFrame->planeHeight(p);
    }
    return true;
}

bool LumaDecoder::run()
{
    if (!m_initialized) {
        if (!initialize(m_input))
            return false;
    }
    if (m_firstFrame) {  // frame 0 was fetched by initialize()
        m_firstFrame = false;
        return true;
    }
    m_vpxFrame = NULL;
    if (!m_source->readFrame(&m_vpxFrame))  // probably end of stream
        return false;
    for (int p = 0; p < 3; p++)
        m_planePtrs[p] = m_vpxFrame->planes[p];
    return true;
}

LumaFrame *LumaDecoder::decode()
{
    if (m_pipelined) {
        // keep two frames started, complete and return the older one (luma_decoder.h: setPipelined)
        while (lumahip_decode_stream_pending(m_quant.context()) < 2 && pushNext()) {
        }
        const int pending = lumahip_decode_stream_pending(m_quant.context());
        if (pending <= 0)
            return NULL;
        const unsigned int seq = m_pushed - (unsigned int)pending;
        if (lumahip_decode_stream_pop(m_quant.context()) != LUMAHIP_OK)
            throw LumaException(lumahip_last_error(m_quant.context()));
        return (seq & 1) ? &m_frame2 : &m_frame;
    }
    if (!run())
        return NULL;
    allocateFrame(m_frame, m_vpxFrame->d_w, m_vpxFrame->d_h);
    const unsigned char *pl[3] = {m_vpxFrame->planes[0], m_vpxFrame->planes[1], m_vpxFrame->planes[2]};
    const int rc = lumahip_decode_frame_host(m_quant.context(), pl, m_vpxFrame->stride, m_vpxFrame->d_w, m_vpxFrame->d_h,
                                             m_vpxFrame->profile(), m_params.preScaling, m_frame.buffer);
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_quant.context()));
    return &m_frame;
}

// include/luma/luma_decoder.h:89-92 of the reference: the base forwards to its reader.  Here the upstream stage seeks by
// frame; the raw plane stream is constant-rate: time -> frame index at the stream's fps.
void LumaDecoderBase::seekToTime(float tm, bool absolute)
{
    beforeSeek();   // (pipelined mode: what was read ahead belongs to the old position)
    m_time = absolute ? tm : m_time + tm;
    if (m_time < 0.0f)
        m_time = 0.0f;
    LumaPlaneSource *src = getReader();
    const float fd = src->getFrameDuration();
    float fps = fd > 0.0f ? 1.0f / fd : 25.0f;
    if (src == &m_rawReader && m_rawReader.fps() > 0.0f)
        fps = m_rawReader.fps();   // (the stream's own figure, not the reciprocal of its reciprocal)
    src->seekToFrame((unsigned int)(m_time * fps));
    m_firstFrame = false;
}
