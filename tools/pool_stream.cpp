// tools/pool_stream.cpp -- a resident stream carved from the HBM chunk pool, from plain C++ over the C ABI (no Python, no
// torch): what a long-running caller of lumahip_encode_frames_device does to get the rate bench.py reports as `value`.
//
//   pool_stream [batches frames_per_batch [max_chunks]]      (default 12 x 20 frames of 3840x2160, PQ-11 Lu'v', profile 2)
//
// max_chunks > 0 = the SMALL pool: the pool may take at most that many 2 GiB chunks for its probes instead of
// all free memory, keeps only what the `batches` need (2 batches: 2 float + 1 Y + 1 U/V chunk = 8 GB) and returns the rest; the
// decode part is skipped.  Reported next to the two-lane rate: `pooled_frac_ordered` / `plain_frac_ordered`, the same launches
// back to back on one stream (what roofline.frac of bench.py is).
//
// 1. one context, the quantizer of BASELINE configs[1];
// 2. lumahip_pool_create: the free device memory in 2 GiB chunks, sorted into HBM region groups by traffic-only launches;
//    float chunks for the input, Y chunks (another group) and U / V chunks (a third group) for the coded planes;
// 3. the same stream once more in plain lumahip_malloc buffers;
// 4. both streams encoded alternately (passes interleaved, so clocks and temperature are shared), each pass = every batch
//    once inside one unordered section; wall clock around lumahip_sync.
// 5. the stream decoded into PACKED LumaFrames whose buffers come from lumahip_pool_alloc(LUMAHIP_POOL_ROTATING) in stream
//    order, against plain buffers.
// Prints one JSON line: Mpixel/s and fraction of the 8 TB/s roofline (15 B/pixel) for the pooled and the plain stream, both
// directions, and whether planes and decoded floats of the two are byte-identical (they must be: the pool only decides addresses).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define LUMAHIP_EXPERIMENTAL   /* the traffic probe and the synthetic frames are measurement hooks */
#include "lumahip.h"

namespace {
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Stream {
    std::vector<float *> rgb;                       // one pointer per batch
    std::vector<unsigned char *> y, u, v;           // planes of each batch
};

#define OK(expr)                                                                                    \
    do {                                                                                            \
        int rc_ = (expr);                                                                           \
        if (rc_ != LUMAHIP_OK) {                                                                    \
            std::fprintf(stderr, "pool_stream: %s -> %d (%s)\n", #expr, rc_, lumahip_last_error(ctx)); \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)
}  // namespace

int main(int argc, char **argv)
{
    const int nb = argc > 1 ? std::atoi(argv[1]) : 12, B = argc > 2 ? std::atoi(argv[2]) : 20;
    const int max_chunks = argc > 3 ? std::atoi(argv[3]) : 0;
    const unsigned w = 3840, h = 2160;
    const size_t n1 = (size_t)w * h, n3 = 3 * n1;
    const int stride[3] = {(int)(((w + 31) & ~31u) * 2), (int)(((w + 31) & ~31u)), (int)(((w + 31) & ~31u))};   // vpx_img_alloc(I42016, w, h, 32)
    const size_t psz[3] = {(size_t)stride[0] * h, (size_t)stride[1] * (h / 2), (size_t)stride[2] * (h / 2)};
    const size_t CH = (size_t)2 << 30;
    lumahip_ctx *ctx = nullptr;
    if (lumahip_create(&ctx, 0) != LUMAHIP_OK) {
        std::fprintf(stderr, "pool_stream: no HIP device (there is no CPU path)\n");
        return 1;
    }
    std::vector<float> lut(2048);
    OK(lumahip_build_lut(LUMAHIP_PTF_PQ, 11, 1e4f, 0.005f, lut.data(), lut.size()));
    OK(lumahip_set_quantizer(ctx, LUMAHIP_PTF_PQ, 11, LUMAHIP_CS_LUV, 8, 1e4f, 0.005f, lut.data(), lut.size()));

    const size_t fbytes = (size_t)B * n3 * sizeof(float), ybytes = (size_t)B * psz[0];
    const size_t ubytes = (((size_t)B * psz[1]) + ((1u << 20) - 1)) & ~(size_t)((1u << 20) - 1), uvbytes = 2 * ubytes;
    if (fbytes > CH || ybytes > CH || uvbytes > CH) {
        std::fprintf(stderr, "pool_stream: a batch of %d frames does not fit a 2 GiB chunk\n", B);
        return 1;
    }
    const int ypc = (int)(CH / ybytes), uvpc = (int)(CH / uvbytes);
    const int nring = max_chunks > 0 ? 0 : std::min(nb, 6);   // packed decoded frames: a ring of output batches (>> the 256 MB MALL)
    lumahip_pool_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.n_float = nb;
    cfg.n_y = (nb + ypc - 1) / ypc;
    cfg.n_uv = (nb + uvpc - 1) / uvpc;
    cfg.n_striped = (nring + 2) / 3;     // per region group: what the ROTATING allocations below draw from
    cfg.max_chunks = max_chunks;
    lumahip_pool *pool = nullptr;
    const double tp0 = now();
    OK(lumahip_pool_create(ctx, &cfg, &pool));
    const double pool_s = now() - tp0;

    Stream placed, plain;
    std::vector<void *> ychunks, uvchunks;
    for (int k = 0; k < cfg.n_y; k++) {
        void *p = nullptr;
        OK(lumahip_pool_alloc(pool, LUMAHIP_POOL_Y, -1, &p));
        ychunks.push_back(p);
    }
    for (int k = 0; k < cfg.n_uv; k++) {
        void *p = nullptr;
        OK(lumahip_pool_alloc(pool, LUMAHIP_POOL_UV, -1, &p));
        uvchunks.push_back(p);
    }
    for (int b = 0; b < nb; b++) {
        void *p = nullptr;
        OK(lumahip_pool_alloc(pool, LUMAHIP_POOL_FLOAT, -1, &p));
        placed.rgb.push_back((float *)p);
        placed.y.push_back((unsigned char *)ychunks[b / ypc] + (size_t)(b % ypc) * ybytes);
        unsigned char *uv = (unsigned char *)uvchunks[b / uvpc] + (size_t)(b % uvpc) * uvbytes;
        placed.u.push_back(uv);
        placed.v.push_back(uv + ubytes);
        void *q = nullptr;
        OK(lumahip_malloc(ctx, &q, fbytes));
        plain.rgb.push_back((float *)q);
        OK(lumahip_malloc(ctx, &q, ybytes));
        plain.y.push_back((unsigned char *)q);
        OK(lumahip_malloc(ctx, &q, ubytes));
        plain.u.push_back((unsigned char *)q);
        OK(lumahip_malloc(ctx, &q, ubytes));
        plain.v.push_back((unsigned char *)q);
        for (Stream *s : {&placed, &plain})
            OK(lumahip_synth_frames_device(ctx, s->rgb[b], n3, B, w, h, 20250929u, (uint64_t)b * B));
    }
    OK(lumahip_sync(ctx));

    auto pass = [&](Stream &s) -> int {
        int rc = lumahip_begin_unordered(ctx, 0);
        for (int b = 0; b < nb && rc == LUMAHIP_OK; b++) {
            unsigned char *pl[3] = {s.y[b], s.u[b], s.v[b]};
            rc = lumahip_encode_frames_device(ctx, s.rgb[b], n3, B, w, h, 1.0f, 2, pl, stride, psz, nullptr);
        }
        const int rc2 = lumahip_end_unordered(ctx);
        const int rc3 = lumahip_sync(ctx);
        return rc ? rc : (rc2 ? rc2 : rc3);
    };
    for (Stream *s : {&placed, &plain})
        OK(pass(*s));   // warm-up
    std::vector<double> tp, tq;
    const double tstart = now();
    while (now() - tstart < 3.0 || tp.size() < 5) {
        double t0 = now();
        OK(pass(placed));
        tp.push_back(now() - t0);
        t0 = now();
        OK(pass(plain));
        tq.push_back(now() - t0);
    }
    std::sort(tp.begin(), tp.end());
    std::sort(tq.begin(), tq.end());
    // the same launches back to back on ONE stream (no unordered section), 8 rounds of the batches per timing
    auto ordered = [&](Stream &s, double &sec) -> int {
        const int R = 8;
        const double t0 = now();
        for (int r = 0; r < R; r++)
            for (int b = 0; b < nb; b++) {
                unsigned char *pl[3] = {s.y[b], s.u[b], s.v[b]};
                const int rc = lumahip_encode_frames_device(ctx, s.rgb[b], n3, B, w, h, 1.0f, 2, pl, stride, psz, nullptr);
                if (rc)
                    return rc;
            }
        const int rc = lumahip_sync(ctx);
        sec = (now() - t0) / R;
        return rc;
    };
    std::vector<double> op, oq;
    for (int k = 0; k < 9; k++) {
        double t = 0;
        OK(ordered(placed, t));
        op.push_back(t);
        OK(ordered(plain, t));
        oq.push_back(t);
    }
    std::sort(op.begin(), op.end());
    std::sort(oq.begin(), oq.end());
    const double px = (double)nb * B * n1;
    const double fop = px * 15.0 / op[op.size() / 2] / 8e12, foq = px * 15.0 / oq[oq.size() / 2] / 8e12;
    const double rp = px / tp[tp.size() / 2] / 1e6, rq = px / tq[tq.size() / 2] / 1e6;

    // same bytes either way: compare the Y and U planes of the first and the last batch on the host
    bool same = true;
    std::vector<unsigned char> a(ybytes), c(ybytes);
    for (int b : {0, nb - 1}) {
        OK(lumahip_memcpy_d2h(ctx, a.data(), placed.y[b], ybytes));
        OK(lumahip_memcpy_d2h(ctx, c.data(), plain.y[b], ybytes));
        same = same && std::memcmp(a.data(), c.data(), ybytes) == 0;
        OK(lumahip_memcpy_d2h(ctx, a.data(), placed.u[b], (size_t)B * psz[1]));
        OK(lumahip_memcpy_d2h(ctx, c.data(), plain.u[b], (size_t)B * psz[1]));
        same = same && std::memcmp(a.data(), c.data(), (size_t)B * psz[1]) == 0;
    }
    // 5. decode, into PACKED LumaFrames (include/luma/luma_frame.h:84-87: what LumaDecoder::decode() returns): the output buffer of
    //    every batch of a ring is allocated in stream order in the pool's ROTATING mode -- the caller does no group arithmetic --
    //    against the same ring in plain lumahip_malloc buffers; passes interleaved as above.  Same floats either way.
    if (nring == 0) {   // the small pool: encode only
        std::printf("{\"tool\": \"pool_stream\", \"small_pool_max_chunks\": %d, \"batches\": %d, \"frames_per_batch\": %d, "
                    "\"pool_create_s\": %.2f, \"pooled_mpix_s\": %.0f, \"pooled_frac_of_8TBs\": %.4f, \"pooled_frac_ordered\": %.4f, "
                    "\"plain_mpix_s\": %.0f, \"plain_frac_of_8TBs\": %.4f, \"plain_frac_ordered\": %.4f, \"planes_identical\": %s, \"pool\": %s}\n",
                    max_chunks, nb, B, pool_s, rp, rp * 15e6 / 8e12, fop, rq, rq * 15e6 / 8e12, foq, same ? "true" : "false",
                    lumahip_pool_stats_json(pool));
        for (int b = 0; b < nb; b++) {
            (void)lumahip_free(ctx, plain.rgb[b]);
            (void)lumahip_free(ctx, plain.y[b]);
            (void)lumahip_free(ctx, plain.u[b]);
            (void)lumahip_free(ctx, plain.v[b]);
        }
        lumahip_pool_destroy(pool);
        lumahip_destroy(ctx);
        return same ? 0 : 2;
    }
    std::vector<float *> rot(nring), pln(nring);
    for (int k = 0; k < nring; k++) {
        void *p = nullptr;
        OK(lumahip_pool_alloc(pool, LUMAHIP_POOL_ROTATING, k == 0 ? 0 : -1, &p));
        rot[k] = (float *)p;
        OK(lumahip_malloc(ctx, &p, fbytes));
        pln[k] = (float *)p;
    }
    auto dpass = [&](std::vector<float *> &out) -> int {
        int rc = lumahip_begin_unordered(ctx, 0);
        for (int b = 0; b < nb && rc == LUMAHIP_OK; b++) {
            const unsigned char *pl[3] = {placed.y[b], placed.u[b], placed.v[b]};
            rc = lumahip_decode_frames_device(ctx, pl, stride, psz, B, w, h, 2, 1.0f, out[b % nring], n3);
        }
        const int rc2 = lumahip_end_unordered(ctx);
        const int rc3 = lumahip_sync(ctx);
        return rc ? rc : (rc2 ? rc2 : rc3);
    };
    OK(dpass(rot));
    OK(dpass(pln));
    std::vector<double> dr, dq;
    const double dstart = now();
    while (now() - dstart < 2.0 || dr.size() < 5) {
        double t0 = now();
        OK(dpass(rot));
        dr.push_back(now() - t0);
        t0 = now();
        OK(dpass(pln));
        dq.push_back(now() - t0);
    }
    std::sort(dr.begin(), dr.end());
    std::sort(dq.begin(), dq.end());
    const double drr = px / dr[dr.size() / 2] / 1e6, drq = px / dq[dq.size() / 2] / 1e6;
    bool dsame = true;
    {
        std::vector<float> fa(n3), fb(n3);   // the first frame of the last batch written into each ring slot... of slot 0
        OK(lumahip_memcpy_d2h(ctx, fa.data(), rot[(nb - 1) % nring], n3 * sizeof(float)));
        OK(lumahip_memcpy_d2h(ctx, fb.data(), pln[(nb - 1) % nring], n3 * sizeof(float)));
        dsame = std::memcmp(fa.data(), fb.data(), n3 * sizeof(float)) == 0;
    }
    int groups[8];
    for (int k = 0; k < nring && k < 8; k++)
        groups[k] = lumahip_pool_group_of(pool, rot[k]);
    char gbuf[64] = {0};
    for (int k = 0, o = 0; k < nring && k < 8; k++)
        o += std::snprintf(gbuf + o, sizeof gbuf - (size_t)o, "%s%d", k ? ", " : "", groups[k]);
    std::printf("{\"tool\": \"pool_stream\", \"batches\": %d, \"frames_per_batch\": %d, \"passes\": %zu, "
                "\"pool_create_s\": %.2f, \"pooled_mpix_s\": %.0f, \"pooled_frac_of_8TBs\": %.4f, "
                "\"plain_mpix_s\": %.0f, \"plain_frac_of_8TBs\": %.4f, \"pooled_frac_ordered\": %.4f, \"plain_frac_ordered\": %.4f, \"planes_identical\": %s, "
                "\"decode_packed_rotating_mpix_s\": %.0f, \"decode_packed_rotating_frac_of_8TBs\": %.4f, "
                "\"decode_packed_plain_mpix_s\": %.0f, \"decode_packed_plain_frac_of_8TBs\": %.4f, \"decode_ring_groups\": [%s], "
                "\"decoded_identical\": %s, \"pool\": %s}\n",
                nb, B, tp.size(), pool_s, rp, rp * 15e6 / 8e12, rq, rq * 15e6 / 8e12, fop, foq, same ? "true" : "false",
                drr, drr * 15e6 / 8e12, drq, drq * 15e6 / 8e12, gbuf, dsame ? "true" : "false", lumahip_pool_stats_json(pool));
    same = same && dsame;
    for (int k = 0; k < nring; k++)
        (void)lumahip_free(ctx, pln[k]);
    for (int b = 0; b < nb; b++) {
        (void)lumahip_free(ctx, plain.rgb[b]);
        (void)lumahip_free(ctx, plain.y[b]);
        (void)lumahip_free(ctx, plain.u[b]);
        (void)lumahip_free(ctx, plain.v[b]);
    }
    lumahip_pool_destroy(pool);
    lumahip_destroy(ctx);
    return same ? 0 : 2;
}
