// tools/integration/vpx_mkv_sink.h -- INTEGRATION.md, way A: the downstream stages of the reference (libvpx VP9 encoder +
// its MkvInterface Matroska writer), unchanged, attached to this repo's LumaEncoder as a LumaPlaneSink.
//
// Needs libvpx (built with --enable-vp9-highbitdepth) and the reference's include/luma/mkv_interface.h + lib/ebml +
// lib/matroska on the include / link line -- none of which ships here (VP9 and Matroska are out of scope), so this header
// is an integration example; tests/test_host_side.py compiles it against the reference tree and libvpx's public headers
// in the build container to keep it honest.  Usage:
//
//     LumaEncoder enc;  enc.setParams(p);
//     VpxMkvSink sink(p);                 // bitrate, quantizerScale, bitDepth, lossLess, colorSpace, maxLum / minLum
//     enc.setSink(&sink);
//     enc.initialize("video.mkv", w, h);  // -> sink.open + the attachments 430..436 + writeAttachments
//     enc.encode(&frame); ... enc.finish();
//
// The planes handed to addFrame() ARE a vpx_image_t's planes: LumaPlaneBuffer lays them out as vpx_img_alloc(fmt, w, h,
// 32) does, so the image below only borrows pointers and strides.
#ifndef LUMA_HIP_VPX_MKV_SINK_H
#define LUMA_HIP_VPX_MKV_SINK_H

#include <cstring>

#include "luma/luma_encoder.h"    // this repo: LumaPlaneSink, LumaEncoderParams
#include "mkv_interface.h"        // the reference's Matroska writer
#include "vp8cx.h"                // libvpx
#include "vpx_encoder.h"

class VpxMkvSink : public LumaPlaneSink {
public:
    explicit VpxMkvSink(const LumaEncoderParams &p) : m_p(p), m_open(false), m_index(0) { memset(&m_img, 0, sizeof m_img); }
    ~VpxMkvSink() { close(); }

    void open(const char *file, unsigned int w, unsigned int h, int profile, float fps)
    {
        m_writer.openWrite(file, w, h, m_p.maxLum, m_p.minLum);
        m_writer.setFramerate(fps);
        // the image descriptor vpx_img_alloc would produce, minus the allocation
        const vpx_img_fmt_t fmt[4] = {VPX_IMG_FMT_I420, VPX_IMG_FMT_I444, VPX_IMG_FMT_I42016, VPX_IMG_FMT_I44416};
        m_img.fmt = fmt[profile];
        m_img.w = m_img.d_w = w;
        m_img.h = m_img.d_h = h;
        m_img.x_chroma_shift = m_img.y_chroma_shift = (profile == 0 || profile == 2) ? 1 : 0;
        m_img.bit_depth = profile > 1 ? 16 : 8;
        m_img.bps = (profile > 1 ? 16 : 8) * ((profile == 0 || profile == 2) ? 3 : 6) / 2;

        vpx_codec_enc_cfg_t cfg;
        if (vpx_codec_enc_config_default(vpx_codec_vp9_cx(), &cfg, 0))
            throw LumaException("Failed to get default codec config");
        cfg.g_w = w;
        cfg.g_h = h;
        cfg.g_profile = (unsigned int)profile;
        cfg.g_threads = 6;
        cfg.g_timebase.num = 1;
        cfg.g_timebase.den = 25;
        cfg.g_error_resilient = 0;
        cfg.g_pass = VPX_RC_ONE_PASS;
        cfg.g_lag_in_frames = 0;
        cfg.rc_end_usage = VPX_Q;
        cfg.rc_min_quantizer = cfg.rc_max_quantizer = m_p.quantizerScale;
        cfg.rc_target_bitrate = m_p.bitrate;
        cfg.kf_mode = VPX_KF_AUTO;
        cfg.kf_max_dist = 25;
        cfg.g_bit_depth = (profile < 2 || m_p.bitDepth == 8) ? VPX_BITS_8 : (m_p.bitDepth == 10 ? VPX_BITS_10 : VPX_BITS_12);
        if (vpx_codec_enc_init(&m_codec, vpx_codec_vp9_cx(), &cfg, profile < 2 ? 0 : VPX_CODEC_USE_HIGHBITDEPTH))
            throw LumaException("Failed to initialize vpxEncoder");
        m_open = true;
        if (m_p.colorSpace == LumaQuantizer::CS_YCBCR)
            vpx_codec_control(&m_codec, VP9E_SET_COLOR_SPACE, 5);   // BT.2020, for third-party decoders
        if (m_p.lossLess && vpx_codec_control(&m_codec, VP9E_SET_LOSSLESS, 1))
            throw LumaException("Failed to use lossless mode");
    }

    void addAttachment(unsigned int id, const void *data, size_t size, const char *description)
    {
        // MkvInterface keeps the pointer until writeAttachments(): give it its own copy
        binary *copy = new binary[size];
        memcpy(copy, data, size);
        m_writer.addAttachment(id, copy, (unsigned int)size, description);
    }
    void writeAttachments() { m_writer.writeAttachments(); }

    bool addFrame(const LumaPlanes &planes)
    {
        for (int p = 0; p < 3; p++) {
            m_img.planes[p] = planes.planes[p];
            m_img.stride[p] = planes.stride[p];
        }
        int flags = 0;
        if (m_p.keyframeInterval > 0 && m_index % m_p.keyframeInterval == 0)
            flags = VPX_EFLAG_FORCE_KF;
        return submit(&m_img, (int)m_index++, flags) >= 0;
    }

    void close()
    {
        if (!m_open)
            return;
        while (submit(NULL, -1, 0) > 0) {   // flush the encoder
        }
        m_writer.close();
        vpx_codec_destroy(&m_codec);
        m_open = false;
    }

private:
    // one vpx_codec_encode call + its packet loop; returns the number of frame packets written, -1 on error
    int submit(vpx_image_t *img, int pts, int flags)
    {
        if (vpx_codec_encode(&m_codec, img, pts, 1, flags, VPX_DL_GOOD_QUALITY) != VPX_CODEC_OK)
            return -1;
        int packets = 0;
        vpx_codec_iter_t it = NULL;
        while (const vpx_codec_cx_pkt_t *pkt = vpx_codec_get_cx_data(&m_codec, &it)) {
            if (pkt->kind != VPX_CODEC_CX_FRAME_PKT)
                continue;
            m_writer.addFrame((const uint8 *)pkt->data.frame.buf, (unsigned int)pkt->data.frame.sz,
                              (pkt->data.frame.flags & VPX_FRAME_IS_KEY) != 0);
            packets++;
        }
        return packets;
    }

    LumaEncoderParams m_p;
    MkvInterface m_writer;
    vpx_codec_ctx_t m_codec;
    vpx_image_t m_img;
    bool m_open;
    unsigned int m_index;
};

#endif
