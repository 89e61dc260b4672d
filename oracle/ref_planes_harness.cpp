/*
 * ref_planes_harness.cpp -- drives the REAL reference plane loops, LumaEncoder::setChannels / setVpxChannel
 * (/root/reference/src/luma_encoder.cpp:196-201,260-317) and LumaDecoder::getVpxChannels
 * (/root/reference/src/luma_decoder.cpp:205-240), on caller-supplied data.
 * TEST INFRASTRUCTURE ONLY; built only where /root/reference exists (this container), by `make -C oracle ref_planes`.
 *
 * The reference's translation units are compiled unmodified, where they lie, and linked COMPLETELY (oracle/Makefile,
 * `ref_planes`): against the vendored libvpx built generic-gnu by its own configure + make, and the vendored libebml /
 * libmatroska sources.  Nothing is left unresolved and nothing stands in for a libvpx function.  The harness itself never
 * calls initialize() / run(): the two plane loops only read / write a vpx_image_t's public data fields (planes, stride,
 * d_w, d_h, chroma shifts, fmt), which it fills with the caller's buffers exactly as vpx_img_alloc(fmt, w, h, 32) / the VP9
 * decoder would describe them -- so that the loops can be driven with chosen strides, garbage codes and ragged sizes.
 * (The complete applications, lumaenc / lumadec through VP9 and Matroska, are oracle/_ref/full/lumaenc_ref / lumadec_ref.)
 *
 *   ref_planes_tool enc ptf bits cs bitsC maxLum minLum profile w h s0 s1 s2 xform sc in.f32 out.planes [lut.f32]
 *       in: 3*w*h floats (LumaFrame layout).  xform=1: m_quant.transformColorSpace(frame,true,sc) first, i.e. the body
 *       of LumaEncoder::encode minus run().  out: plane 0 (rows x s0 bytes), 1, 2 -- pre-filled with 0xA5 so that bytes
 *       the reference does not write are visible.  The "Mean luminance" warning of setVpxChannel goes to stderr as is.
 *   ref_planes_tool dec ptf bits cs bitsC maxLum minLum profile w h s0 s1 s2 xform sc in.planes out.f32 [lut.f32]
 *       getVpxChannels (+ transformColorSpace(frame,false,sc) when xform=1) -> 3*w*h floats.
 *   lut.f32 (optional): getSize() floats copied over getMapping(), what LumaDecoder::initialize does with MKV
 *   attachment 434 (src/luma_decoder.cpp:121-122).
 */
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#define private public
#define protected public
#include "luma_decoder.h"
#include "luma_encoder.h"
#undef private
#undef protected

static std::vector<unsigned char> slurp(const char *p)
{
    std::ifstream f(p, std::ios::binary);
    if (!f) {
        fprintf(stderr, "cannot read %s\n", p);
        exit(2);
    }
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
    if (argc < 18) {
        fprintf(stderr, "usage: see the header of oracle/ref_planes_harness.cpp\n");
        return 2;
    }
    const bool enc = !strcmp(argv[1], "enc");
    const int ptf = atoi(argv[2]);
    const unsigned bits = atoi(argv[3]);
    const int cs = atoi(argv[4]);
    const unsigned bitsC = atoi(argv[5]);
    const float maxLum = (float)atof(argv[6]), minLum = (float)atof(argv[7]);
    const int profile = atoi(argv[8]);
    const unsigned w = atoi(argv[9]), h = atoi(argv[10]);
    int stride[4] = {atoi(argv[11]), atoi(argv[12]), atoi(argv[13]), 0};
    const int xform = atoi(argv[14]);
    const float sc = (float)atof(argv[15]);
    const char *inp = argv[16], *outp = argv[17];
    const bool sub = (profile == 0 || profile == 2);
    const unsigned cw = sub ? (w + 1) / 2 : w, ch = sub ? (h + 1) / 2 : h;
    const size_t psz[3] = {(size_t)h * stride[0], (size_t)ch * stride[1], (size_t)ch * stride[2]};

    // the vpx_image_t the encoder's vpx_img_alloc / the VP9 decoder would hand to the loops (src/luma_encoder.cpp:121-128)
    vpx_image_t img;
    memset(&img, 0, sizeof img);
    img.fmt = profile == 0 ? VPX_IMG_FMT_I420 : profile == 1 ? VPX_IMG_FMT_I444 : profile == 2 ? VPX_IMG_FMT_I42016 : VPX_IMG_FMT_I44416;
    img.w = img.d_w = w;
    img.h = img.d_h = h;
    img.x_chroma_shift = img.y_chroma_shift = sub ? 1 : 0;
    img.bit_depth = profile > 1 ? 16 : 8;
    std::vector<unsigned char> planes(psz[0] + psz[1] + psz[2], 0xA5);
    img.planes[0] = planes.data();
    img.planes[1] = planes.data() + psz[0];
    img.planes[2] = planes.data() + psz[0] + psz[1];
    for (int p = 0; p < 3; p++)
        img.stride[p] = stride[p];

    if (enc) {
        LumaEncoder e;                       // constructor only: no codec, no file
        LumaEncoderParams prm = e.getParams();
        prm.profile = profile;
        prm.preScaling = sc;
        e.setParams(prm);
        e.m_quant.setQuantizer((LumaQuantizer::ptf_t)ptf, bits, (LumaQuantizer::colorSpace_t)cs, bitsC, maxLum, minLum);
        if (argc > 18) {
            std::vector<unsigned char> l = slurp(argv[18]);
            memcpy((void *)e.m_quant.getMapping(), l.data(), std::min(l.size(), (size_t)e.m_quant.getSize() * 4));
        }
        e.m_rawFrame = img;
        std::vector<unsigned char> in = slurp(inp);
        if (in.size() != (size_t)3 * w * h * 4) {
            fprintf(stderr, "input size mismatch\n");
            return 2;
        }
        LumaFrame f;
        f.width = w;
        f.height = h;
        f.channels = 3;
        f.buffer = new float[(size_t)3 * w * h];
        memcpy(f.buffer, in.data(), in.size());
        if (xform)
            e.m_quant.transformColorSpace(&f, true, prm.preScaling);   // LumaEncoder::encode, luma_encoder.h:142-148
        e.setChannels(&f);
        std::ofstream o(outp, std::ios::binary);
        o.write((const char *)planes.data(), planes.size());
    } else {
        LumaDecoder d(NULL);                 // no input file: initialize() is not run
        LumaDecoderParams prm = d.getParams();
        prm.profile = profile;
        prm.preScaling = sc;
        prm.stride = stride;
        prm.width[0] = w; prm.height[0] = h;
        prm.width[1] = prm.width[2] = cw; prm.height[1] = prm.height[2] = ch;   // src/luma_decoder.cpp:150-160
        d.setParams(prm);
        d.m_quant.setQuantizer((LumaQuantizer::ptf_t)ptf, bits, (LumaQuantizer::colorSpace_t)cs, bitsC, maxLum, minLum);
        if (argc > 18) {
            std::vector<unsigned char> l = slurp(argv[18]);
            memcpy((void *)d.m_quant.getMapping(), l.data(), std::min(l.size(), (size_t)d.m_quant.getSize() * 4));
        }
        std::vector<unsigned char> in = slurp(inp);
        if (in.size() != planes.size()) {
            fprintf(stderr, "input size mismatch\n");
            return 2;
        }
        memcpy(planes.data(), in.data(), in.size());
        d.m_vpxFrame = &img;
        d.m_frame.width = w;                 // LumaDecoder::decode, luma_decoder.h:148-154
        d.m_frame.height = h;
        d.m_frame.channels = 3;
        d.m_frame.init();
        d.getVpxChannels();
        if (xform)
            d.m_quant.transformColorSpace(&d.m_frame, false, prm.preScaling);
        std::ofstream o(outp, std::ios::binary);
        o.write((const char *)d.m_frame.buffer, (size_t)3 * w * h * 4);
    }
    fflush(stderr);
    return 0;
}
