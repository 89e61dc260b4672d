// Facade LumaBatchEncoder: LumaEncoder's parameter handling and metadata, the hot path through the many-GPU layer of
// the C ABI (lumahip_multi_*).
#include "../../../include/luma/luma_batch_encoder.h"

#include <string>

#include "encoder_common.h"

LumaBatchEncoder::LumaBatchEncoder()
    : m_multi(NULL), m_sink(NULL), m_maxVal(0), m_frameCount(0), m_w(0), m_h(0), m_lastMean(0.0f), m_initialized(false)
{
}

LumaBatchEncoder::~LumaBatchEncoder()
{
    if (m_multi)
        lumahip_multi_destroy(m_multi);
}

unsigned int LumaBatchEncoder::shards() const { return m_multi ? (unsigned int)lumahip_multi_shards(m_multi) : 0; }

bool LumaBatchEncoder::quantizerCameOverRccl() const { return m_multi && lumahip_multi_used_rccl(m_multi) != 0; }

bool LumaBatchEncoder::initialize(const char *outputFile, const unsigned int w, const unsigned int h, bool verbose,
                                  const int *devices, int nshards)
{
    if (!m_sink)
        m_sink = &m_rawWriter;
    luma_detail::checkGeometry(m_params, w, h);
    if (m_params.ptfBitDepth < 1 || m_params.ptfBitDepth > 16 || m_params.colorBitDepth < 1 || m_params.colorBitDepth > 16)
        throw LumaException("PTF / colour bit depth must be in 1..16");
    if (m_multi) {
        lumahip_multi_destroy(m_multi);
        m_multi = NULL;
    }
    if (lumahip_multi_create(&m_multi, devices, nshards) != LUMAHIP_OK)
        throw LumaException("No usable HIP device for the Luma HDRv quantizer (there is no CPU fallback)");

    m_sink->open(outputFile, w, h, (int)m_params.profile, m_params.fps);
    // the table: built once, on the host, exactly as LumaQuantizer::setQuantizer builds it; RCCL carries it to the GPUs
    m_maxVal = (1u << m_params.ptfBitDepth) - 1;
    m_mapping.assign((size_t)m_maxVal + 1, 0.0f);
    int rc = lumahip_build_lut((int)m_params.ptf, m_params.ptfBitDepth, m_params.maxLum, m_params.minLum, m_mapping.data(),
                               m_mapping.size());
    if (rc == LUMAHIP_ERR_UNSUPPORTED)
        throw LumaException("PSI / JND-HDR-VDP tables exist for at most 12 bits");
    if (rc != LUMAHIP_OK)
        throw LumaException("Cannot build the transfer function table (missing lumahdrv_amd/data/ptf_*.f32?)");
    rc = lumahip_multi_set_quantizer(m_multi, (int)m_params.ptf, m_params.ptfBitDepth, (int)m_params.colorSpace,
                                     m_params.colorBitDepth, m_params.maxLum, m_params.minLum, m_mapping.data(), m_mapping.size());
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_multi_last_error(m_multi));
    luma_detail::writeAttachments(m_sink, m_params, m_mapping.data(), m_maxVal);

    m_w = w;
    m_h = h;
    m_planes.clear();   // plane buffers of an earlier initialize() belong to its geometry / profile
    const std::string how = "HIP / gfx950, " + std::to_string(shards()) + " shard(s), table over " +
                            (quantizerCameOverRccl() ? "RCCL" : "host copies");
    luma_detail::printBanner(m_params, outputFile, how.c_str());
    (void)verbose;
    m_frameCount = 0;
    m_initialized = true;
    return true;
}

bool LumaBatchEncoder::encode(LumaFrame *const *frames, unsigned int n)
{
    if (!m_initialized)
        throw LumaException("Encoder not initialized");
    if (n == 0)
        return true;
    std::vector<const float *> rgb(n);
    for (unsigned int i = 0; i < n; i++) {
        if (!frames[i] || !frames[i]->buffer || frames[i]->width != m_w || frames[i]->height != m_h || frames[i]->channels < 3)
            throw LumaException("Frame size differs from the size the encoder was initialized with");
        rgb[i] = frames[i]->buffer;
    }
    while (m_planes.size() < n) {
        m_planes.emplace_back();
        m_planes.back().allocate(m_w, m_h, (int)m_params.profile);
    }
    std::vector<unsigned char *> planes(3 * (size_t)n);
    for (unsigned int i = 0; i < n; i++)
        for (int p = 0; p < 3; p++)
            planes[3 * (size_t)i + p] = m_planes[i].image().planes[p];
    std::vector<float> means(n, 0.0f);
    const int rc = lumahip_multi_encode_frames_host(m_multi, rgb.data(), n, m_w, m_h, m_params.preScaling, (int)m_params.profile,
                                                    planes.data(), m_planes[0].image().stride, means.data());
    if (rc != LUMAHIP_OK)
        throw LumaException(lumahip_multi_last_error(m_multi));
    // downstream is sequential: frame order
    for (unsigned int i = 0; i < n; i++) {
        m_lastMean = means[i];
        luma_detail::warnMean(means[i]);
        m_frameCount++;
        if (!m_sink->addFrame(m_planes[i].image()))
            fprintf(stderr, "Failed to encode frame\n");
    }
    return true;
}

void LumaBatchEncoder::finish()
{
    if (m_sink)
        m_sink->close();
}
