// tools/integration/vpx_mkv_source.h -- INTEGRATION.md, way A, decode side: the upstream stages of the reference (its
// MkvInterface Matroska reader + the libvpx VP9 decoder), unchanged, attached to this repo's LumaDecoder as a LumaPlaneSource.
//
// What the reference's LumaDecoder does around its hot loop (src/luma_decoder.cpp:63-75 open + attachments, :137-141 the
// decoder with 4 threads, :168-202 readFrame -> vpx_codec_decode -> vpx_codec_get_frame) happens here; the planes of the
// vpx_image_t the decoder returns -- with the strides libvpx chose -- are handed to the fused decode kernel as they are.
// Needs libvpx (--enable-vp9-highbitdepth) and the reference's include/luma/mkv_interface.h + lib/ebml + lib/matroska on the
// include / link line; `make -C oracle ref_full` links tests/cpp/source_decode.cpp with it in the build container.
//
//     VpxMkvSource src;
//     LumaDecoder dec;  dec.setSource(&src);
//     dec.initialize("video.mkv");
//     while (LumaFrame *f = dec.decode()) ...
#ifndef LUMA_HIP_VPX_MKV_SOURCE_H
#define LUMA_HIP_VPX_MKV_SOURCE_H

#include <cstring>

#include "luma/luma_decoder.h"    // this repo: LumaPlaneSource, LumaPlanes
#include "mkv_interface.h"        // the reference's Matroska reader
#include "vp8dx.h"                // libvpx
#include "vpx_decoder.h"

class VpxMkvSource : public LumaPlaneSource {
public:
    VpxMkvSource() : m_open(false) { memset(&m_planes, 0, sizeof m_planes); }
    ~VpxMkvSource()
    {
        if (m_open)
            vpx_codec_destroy(&m_codec);
    }

    void open(const char *file)
    {
        m_reader.openRead(file);
        vpx_codec_dec_cfg_t cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.threads = 4;
        if (vpx_codec_dec_init(&m_codec, vpx_codec_vp9_dx(), &cfg, 0))
            throw LumaException("Failed to initialize decoder");
        m_open = true;
    }

    bool getAttachment(unsigned int index, unsigned char **buffer, unsigned int &id, unsigned int &size)
    {
        binary *b = NULL;
        if (!m_reader.getAttachment(index, &b, id, size))
            return false;
        *buffer = (unsigned char *)b;
        return true;
    }

    bool readFrame(const LumaPlanes **img)
    {
        if (!m_reader.readFrame())   // probably the end of the stream
            return false;
        unsigned int size = 0;
        const uint8 *packet = m_reader.getFrame(size);
        if (vpx_codec_decode(&m_codec, packet, size, NULL, 0))
            throw LumaException("Failed to decode frame");
        vpx_codec_iter_t it = NULL;
        const vpx_image_t *v = vpx_codec_get_frame(&m_codec, &it);
        if (v == NULL)
            throw LumaException("Failed to get decoded frame");
        for (int p = 0; p < 3; p++) {
            m_planes.planes[p] = v->planes[p];
            m_planes.stride[p] = v->stride[p];
        }
        m_planes.d_w = v->d_w;
        m_planes.d_h = v->d_h;
        m_planes.x_chroma_shift = v->x_chroma_shift;
        m_planes.y_chroma_shift = v->y_chroma_shift;
        m_planes.highBitDepth = (v->fmt & VPX_IMG_FMT_HIGHBITDEPTH) != 0;
        *img = &m_planes;
        return true;
    }

    // MkvInterface counts time in Matroska timecode units of 1 ms (TIMECODE_SCALE, include/luma/mkv_interface.h in the
    // reference): its cue times are frame index x frame duration (src/mkv_interface.cpp:335), which is what an absolute
    // seekToTime() is compared with (:657-669)
    bool seekToFrame(unsigned int index) { return m_reader.seekToTime((float)index * (float)m_reader.getFrameDuration(), true); }
    float getDuration() { return (float)m_reader.getDuration() / 1000.0f; }
    float getFrameDuration() { return (float)m_reader.getFrameDuration() / 1000.0f; }

private:
    MkvInterface m_reader;
    vpx_codec_ctx_t m_codec;
    LumaPlanes m_planes;
    bool m_open;
};

#endif
