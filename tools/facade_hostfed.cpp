// tools/facade_hostfed.cpp -- what the DROP-IN call costs end to end: LumaEncoder::encode(LumaFrame*) on host frames
// (include/luma/luma_encoder.h:142-148 of the reference; its callers: lumaenc.cpp:238, test/test_simple_enc.cpp:66), i.e.
// H2D of 12 B/pixel + the fused kernel + D2H of the planes, one frame per call, synchronous.  PCIe-bound by construction;
// bench.py prints these figures as `facade_hostfed` next to (never as) the device-resident `value`.
//   facade_hostfed [w h frames]          -> one JSON line on stdout
// Rows: a pageable LumaFrame (plain new float[], what the reference's LumaFrame is), the same loop in the facade's pipelined
// mode, the same frame pinned with
// lumahip_host_register, the batched pinned C-ABI entry point (3-slot pipeline), and LumaDecoder-side decode into a
// pageable frame (lumahip_decode_frame_host, what LumaDecoder::decode calls).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "luma/luma_encoder.h"
#include "luma/luma_test_pattern.h"
#include "lumahip.h"

namespace {
// float -> nearest binary16 -> float (round to nearest even; the finite, non-tiny values of the test pattern): what a frame
// looks like after the reference's EXR reader (Imf::Rgba, src/exr_interface.cpp:77-146)
float to_half_and_back(float v)
{
    uint32_t b;
    memcpy(&b, &v, 4);
    const uint32_t e = (b >> 23) & 0xff;
    if (e < 113 || e > 142)   // below the normal halves / beyond 65504: not in the pattern; leave as it is
        return v;
    const uint32_t rem = b & 0x1fff, lsb = (b >> 13) & 1;
    b &= ~0x1fffu;
    if (rem > 0x1000 || (rem == 0x1000 && lsb))
        b += 0x2000;
    float r;
    memcpy(&r, &b, 4);
    return r;
}
struct NullSink : LumaPlaneSink {
    void open(const char *, unsigned int, unsigned int, int, float) {}
    void addAttachment(unsigned int, const void *, size_t, const char *) {}
    void writeAttachments() {}
    bool addFrame(const LumaPlanes &) { return true; }
    void close() {}
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char **argv)
{
    const unsigned w = argc > 1 ? std::atoi(argv[1]) : 3840, h = argc > 2 ? std::atoi(argv[2]) : 2160;
    const int n = argc > 3 ? std::atoi(argv[3]) : 24;
    try {
        NullSink sink;
        LumaEncoder enc;
        enc.setSink(&sink);
        FILE *saved = stderr;
        (void)saved;
        enc.initialize("null", w, h);
        // distinct frames (a transcoder never encodes the same memory twice): 4 buffers, cycled
        std::vector<std::unique_ptr<LumaFrame>> fr;
        for (int i = 0; i < 4; i++) {
            fr.emplace_back(new LumaFrame());
            lumaTestFrame(*fr.back(), w, h);
            for (size_t j = 0; j < fr.back()->pixelCount(); j += 997)
                fr.back()->buffer[j] *= 1.0f + 0.01f * i;
        }
        const double px = (double)w * h;
        enc.encode(fr[0].get());  // warm-up: allocations, code object
        enc.encode(fr[1].get());
        double t0 = now();
        for (int i = 0; i < n; i++)
            enc.encode(fr[i % 4].get());
        const double pageable = n * px / (now() - t0) / 1e6;

        // the same loop with LumaEncoder::setPipelined(true): frame i+1 goes up before frame i is completed (one frame of latency)
        double pipelined = 0.0;
        {
            NullSink psink;
            LumaEncoder penc;
            penc.setSink(&psink);
            penc.setPipelined(true);
            penc.initialize("null", w, h);
            penc.encode(fr[0].get());
            penc.encode(fr[1].get());
            const double tp = now();
            for (int i = 0; i < n; i++)
                penc.encode(fr[i % 4].get());
            pipelined = n * px / (now() - tp) / 1e6;
            penc.finish();
        }

        // the same three loops on frames that hold binary16 values -- what ExrInterface::readFrame hands the encoder: such
        // frames cross PCIe as halves (include/lumahip.h, "Half upload")
        std::vector<std::unique_ptr<LumaFrame>> hfr;
        for (int i = 0; i < 4; i++) {
            hfr.emplace_back(new LumaFrame());
            lumaTestFrame(*hfr.back(), w, h);
            float *b = hfr.back()->buffer;
            for (size_t j = 0; j < hfr.back()->pixelCount(); j++)   // (pixelCount() counts all three channels)
                b[j] = to_half_and_back(b[j] * (1.0f + 0.01f * i));
        }
        double half_sync = 0.0, half_pipe = 0.0, half_batch = 0.0;
        long hi[3] = {0, 0, 0};   // lumahip_half_upload_info of the batched half-valued call's context
        {
            NullSink hsink;
            LumaEncoder henc;
            henc.setSink(&hsink);
            henc.initialize("null", w, h);
            henc.encode(hfr[0].get());
            henc.encode(hfr[1].get());
            double th = now();
            for (int i = 0; i < n; i++)
                henc.encode(hfr[i % 4].get());
            half_sync = n * px / (now() - th) / 1e6;
            NullSink psink;
            LumaEncoder penc;
            penc.setSink(&psink);
            penc.setPipelined(true);
            penc.initialize("null", w, h);
            penc.encode(hfr[0].get());
            penc.encode(hfr[1].get());
            th = now();
            for (int i = 0; i < n; i++)
                penc.encode(hfr[i % 4].get());
            half_pipe = n * px / (now() - th) / 1e6;
            penc.finish();
            // batched, pageable frames and planes
            const LumaPlanes &him = henc.getRawFrame();
            const int hst[3] = {him.stride[0], him.stride[1], him.stride[2]};
            std::vector<std::vector<unsigned char>> hp(3 * 4);
            for (int k = 0; k < 4; k++)
                for (int p = 0; p < 3; p++)
                    hp[3 * k + p].assign((size_t)him.planeHeight(p) * hst[p] + 4096, 0);
            std::vector<const float *> hrgb(n);
            std::vector<unsigned char *> hpl(3 * (size_t)n);
            for (int i = 0; i < n; i++) {
                hrgb[i] = hfr[i % 4]->buffer;
                for (int p = 0; p < 3; p++)
                    hpl[3 * (size_t)i + p] = hp[3 * (i % 4) + p].data();
            }
            lumahip_ctx *hctx = henc.getQuantizer()->context();
            const LumaEncoderParams hprm = henc.getParams();
            (void)lumahip_encode_frames_host(hctx, hrgb.data(), 4, w, h, hprm.preScaling, (int)hprm.profile, hpl.data(), hst, nullptr);
            th = now();
            if (lumahip_encode_frames_host(hctx, hrgb.data(), n, w, h, hprm.preScaling, (int)hprm.profile, hpl.data(), hst, nullptr) != LUMAHIP_OK)
                throw LumaException(lumahip_last_error(hctx));
            half_batch = n * px / (now() - th) / 1e6;
            (void)lumahip_half_upload_info(hctx, hi);
            if (hi[0] == 0)
                std::fprintf(stderr, "facade_hostfed: note: no frame went up as halves (no F16C on this host?)\n");
        }

        lumahip_ctx *ctx = enc.getQuantizer()->context();
        for (auto &f : fr)
            if (lumahip_host_register(ctx, f->buffer, f->pixelCount() * sizeof(float)) != LUMAHIP_OK)
                throw LumaException(lumahip_last_error(ctx));
        enc.encode(fr[0].get());
        t0 = now();
        for (int i = 0; i < n; i++)
            enc.encode(fr[i % 4].get());
        const double registered = n * px / (now() - t0) / 1e6;

        // batched C-ABI entry point, frames and planes pinned
        const LumaPlanes &im = enc.getRawFrame();
        const int st[3] = {im.stride[0], im.stride[1], im.stride[2]};
        const size_t psz[3] = {(size_t)im.planeHeight(0) * st[0], (size_t)im.planeHeight(1) * st[1], (size_t)im.planeHeight(2) * st[2]};
        std::vector<std::vector<unsigned char>> pl(3 * 4);
        std::vector<unsigned char *> plp(3 * (size_t)n);
        for (int k = 0; k < 4; k++)
            for (int p = 0; p < 3; p++) {
                pl[3 * k + p].assign(psz[p] + 4096, 0);
                (void)lumahip_host_register(ctx, pl[3 * k + p].data(), pl[3 * k + p].size());
            }
        std::vector<const float *> rgb(n);
        for (int i = 0; i < n; i++) {
            rgb[i] = fr[i % 4]->buffer;
            for (int p = 0; p < 3; p++)
                plp[3 * (size_t)i + p] = pl[3 * (i % 4) + p].data();
        }
        const LumaEncoderParams prm = enc.getParams();
        (void)lumahip_encode_frames_host(ctx, rgb.data(), 4, w, h, prm.preScaling, (int)prm.profile, plp.data(), st, nullptr);
        t0 = now();
        if (lumahip_encode_frames_host(ctx, rgb.data(), n, w, h, prm.preScaling, (int)prm.profile, plp.data(), st, nullptr) != LUMAHIP_OK)
            throw LumaException(lumahip_last_error(ctx));
        const double batch = n * px / (now() - t0) / 1e6;

        // the same batched entry point on PAGEABLE frames and planes (what LumaBatchEncoder gets from plain LumaFrames)
        std::vector<std::vector<float>> pfr(4);
        std::vector<std::vector<unsigned char>> ppl(3 * 4);
        for (int k = 0; k < 4; k++) {
            pfr[k].assign(fr[k]->buffer, fr[k]->buffer + fr[k]->pixelCount());
            for (int p = 0; p < 3; p++)
                ppl[3 * k + p].assign(psz[p] + 4096, 0);
        }
        std::vector<const float *> prgb(n);
        std::vector<unsigned char *> pplp(3 * (size_t)n);
        for (int i = 0; i < n; i++) {
            prgb[i] = pfr[i % 4].data();
            for (int p = 0; p < 3; p++)
                pplp[3 * (size_t)i + p] = ppl[3 * (i % 4) + p].data();
        }
        (void)lumahip_encode_frames_host(ctx, prgb.data(), 4, w, h, prm.preScaling, (int)prm.profile, pplp.data(), st, nullptr);
        t0 = now();
        if (lumahip_encode_frames_host(ctx, prgb.data(), n, w, h, prm.preScaling, (int)prm.profile, pplp.data(), st, nullptr) != LUMAHIP_OK)
            throw LumaException(lumahip_last_error(ctx));
        const double batch_pageable = n * px / (now() - t0) / 1e6;

        // decode into a pageable frame (what LumaDecoder::decode hands back)
        std::vector<float> out((size_t)3 * w * h);
        const unsigned char *cpl[3] = {im.planes[0], im.planes[1], im.planes[2]};
        (void)lumahip_decode_frame_host(ctx, cpl, st, w, h, (int)prm.profile, prm.preScaling, out.data());
        t0 = now();
        for (int i = 0; i < n; i++)
            if (lumahip_decode_frame_host(ctx, cpl, st, w, h, (int)prm.profile, prm.preScaling, out.data()) != LUMAHIP_OK)
                throw LumaException(lumahip_last_error(ctx));
        const double dec = n * px / (now() - t0) / 1e6;
        // the same per-frame decode through the streaming entry points (what LumaDecoder::setPipelined(true) calls): frame i+1 is
        // uploaded and launched before frame i is completed
        std::vector<float> out2((size_t)3 * w * h);
        double dec_pipe = 0.0;
        {
            float *ob[2] = {out.data(), out2.data()};
            (void)lumahip_decode_stream_push(ctx, cpl, st, w, h, (int)prm.profile, prm.preScaling, ob[0]);
            t0 = now();
            for (int i = 1; i <= n; i++) {
                if (lumahip_decode_stream_push(ctx, cpl, st, w, h, (int)prm.profile, prm.preScaling, ob[i & 1]) != LUMAHIP_OK ||
                    lumahip_decode_stream_pop(ctx) != LUMAHIP_OK)
                    throw LumaException(lumahip_last_error(ctx));
            }
            dec_pipe = n * px / (now() - t0) / 1e6;
            (void)lumahip_decode_stream_pop(ctx);
        }
        // batched decode into pageable frames (4 distinct outputs)
        std::vector<std::vector<float>> outs(4, std::vector<float>((size_t)3 * w * h));
        std::vector<const unsigned char *> dpl(3 * (size_t)n);
        std::vector<float *> dout(n);
        for (int i = 0; i < n; i++) {
            dout[i] = outs[i % 4].data();
            for (int p = 0; p < 3; p++)
                dpl[3 * (size_t)i + p] = ppl[3 * (i % 4) + p].data();
        }
        (void)lumahip_decode_frames_host(ctx, dpl.data(), st, 4, w, h, (int)prm.profile, prm.preScaling, dout.data());
        t0 = now();
        if (lumahip_decode_frames_host(ctx, dpl.data(), st, n, w, h, (int)prm.profile, prm.preScaling, dout.data()) != LUMAHIP_OK)
            throw LumaException(lumahip_last_error(ctx));
        const double dec_batch = n * px / (now() - t0) / 1e6;
        for (auto &f : fr)
            (void)lumahip_host_unregister(ctx, f->buffer);
        for (auto &v : pl)
            (void)lumahip_host_unregister(ctx, v.data());
        // LumaQuantizer::quantize / dequantize per value (src/luma_quantizer.cpp:215-264): scalar host calls, as in the reference
        double q_ns = 0.0, dq_ns = 0.0;
        {
            LumaQuantizer *q = enc.getQuantizer();
            const int m = 2000000;
            float acc = 0.0f, v = 0.003f;
            t0 = now();
            for (int i = 0; i < m; i++) {
                acc += q->quantize(v, 0) + q->quantize(v * 1e-4f, 1);
                v = v * 1.00001f + 1e-6f;
                if (v > 9000.0f)
                    v = 0.003f;
            }
            q_ns = (now() - t0) / (2.0 * m) * 1e9;
            t0 = now();
            for (int i = 0; i < m; i++)
                acc += q->dequantize((float)(i & 2047), 0) + q->dequantize((float)(i & 255), 1);
            dq_ns = (now() - t0) / (2.0 * m) * 1e9;
            if (acc == 12345.0f)
                std::fprintf(stderr, "%g\n", acc);   // keeps the loops alive
        }
        std::printf("{\"width\": %u, \"height\": %u, \"frames\": %d, \"unit\": \"Mpixels/s\", "
                    "\"LumaQuantizer_quantize_ns_per_call\": %.1f, \"LumaQuantizer_dequantize_ns_per_call\": %.1f, "
                    "\"LumaEncoder_encode_pageable_half_valued_frame\": %.1f, \"LumaEncoder_pipelined_encode_pageable_half_valued_frame\": %.1f, "
                    "\"lumahip_encode_frames_host_pageable_half_valued\": %.1f, "
                    "\"LumaEncoder_encode_pageable_frame\": %.1f, \"LumaEncoder_pipelined_encode_pageable_frame\": %.1f, "
                    "\"LumaEncoder_encode_registered_frame\": %.1f, "
                    "\"lumahip_encode_frames_host_pinned\": %.1f, \"lumahip_encode_frames_host_pageable\": %.1f, "
                    "\"decode_frame_host_pageable\": %.1f, \"decode_stream_pageable\": %.1f, \"lumahip_decode_frames_host_pageable\": %.1f, "
                    "\"half_upload_info\": [%ld, %ld, %ld]}\n",
                    w, h, n, q_ns, dq_ns, half_sync, half_pipe, half_batch, pageable, pipelined, registered, batch, batch_pageable, dec, dec_pipe, dec_batch,
                    hi[0], hi[1], hi[2]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "facade_hostfed: %s\n", e.what());
        return 1;
    }
    return 0;
}
