// lumahip_pool.hip -- HBM chunk pool (include/lumahip.h, "HBM chunk pool"): WHICH device-memory regions the frames and the
// coded planes of a resident stream live in.
//
// Why this exists (profiles/r02_placement.txt): on MI355X (SPX / NPS1, ROCm 7.2) device memory falls into a few GROUPS of
// multi-GiB regions, and the rate of a launch depends on which groups the streams it reads and writes CONCURRENTLY live in.
// One 20-frame 4K launch of the encode traffic (12 B/pixel read from the float frames, 2 B/pixel written to Y, 1 B/pixel to
// U and V), same box, same minute:
//     input, Y, U, V all in regions of one group      0.464 ms
//     input in group A, Y U V together in group B     0.432 ms
//     input in A, Y in B, U V in A or in a third C     0.397 ms
// The relation is symmetric in the read / write roles, reproducible to three digits, independent of offsets inside a
// region and not a property of a single region; the mechanism is not visible from user space, the groups are.  A plain
// 50 GB allocation pairs its buffers at random (the 4-8 % run-to-run spread of the round-1 bench).
//
// Steps: (1) take the free memory in chunks; (2) find the groups: round k takes the first unclassified chunk r and times
// every other unclassified chunk against it (read chunk i, write chunk r) -- the slow ones share r's group; (3) Y candidates =
// the smallest group that is large enough, U / V candidates = another group, reference float chunk = first chunk of the
// largest remaining group; keep the Y and U / V candidates that run fastest with the reference; (4) rank ALL remaining chunks
// as float chunks by their time with the chosen planes chunks and keep the fastest -- so an imperfect grouping costs probes,
// not bandwidth; (5) optionally keep n_striped more chunks of each of the first three groups (channel-strided decode
// output); everything else goes back to the driver.  Every measurement is lumahip_probe_encode_traffic_device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#define LUMAHIP_EXPERIMENTAL   /* the pool finds the region groups with the traffic probe */
#include "../../include/lumahip.h"

namespace {

constexpr unsigned PROBE_W = 3840, PROBE_H = 2160, PROBE_FRAMES = 20;  // 1.99 GB read + 0.50 GB written per probe launch
// a pair counts as "same group" when it is this much slower than the fastest pair seen (measured: +7 %; repeatability of
// one measurement: 0.5 %)
constexpr double SAME_GROUP_PENALTY = 1.035;

struct Chunk {
    unsigned char *p = nullptr;
    int group = -1;
    int kind = -1;      // lumahip_pool_kind, -1 = not kept
    double t = 0.0;     // the probe time it was ranked by (ms)
    bool out = false;   // handed to the caller
};

double median(std::vector<double> v)
{
    if (v.empty())
        return 0.0;
    std::sort(v.begin(), v.end());
    return v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]);
}

// Partition chunks 0..n-1 into groups; probe(i, r) = time of a launch that reads chunk i and writes chunk r.  Round k takes
// the first unclassified chunk r as reference and times every other unclassified chunk against it: the slow ones share r's
// group.  Returns false when the first round shows no contrast (one group, or a machine without the effect).
bool find_groups(int n, const std::function<double(int, int)> &probe, std::vector<std::vector<int>> &groups, double &fast,
                 size_t max_groups = 16)
{
    std::vector<int> todo(n);
    for (int i = 0; i < n; i++)
        todo[i] = i;
    bool first_round = true;
    groups.clear();
    fast = 0.0;
    while (!todo.empty() && groups.size() < max_groups) {
        const int r = todo[0];
        std::vector<int> others(todo.begin() + 1, todo.end());
        if (others.empty()) {
            groups.push_back({r});
            todo.clear();
            break;
        }
        std::map<int, double> t;
        double lo = 1e30, hi = 0.0;
        for (int i : others) {
            t[i] = probe(i, r);
            lo = std::min(lo, t[i]);
            hi = std::max(hi, t[i]);
        }
        if (first_round) {
            first_round = false;
            fast = lo;
            if (hi <= lo * SAME_GROUP_PENALTY)
                return false;
        }
        fast = std::min(fast, lo);  // (a round whose chunks all share r's group has no fast pair: min keeps `fast`)
        std::vector<int> same = {r}, rest;
        for (int i : others)
            (t[i] > fast * SAME_GROUP_PENALTY ? same : rest).push_back(i);
        groups.push_back(same);
        todo = rest;
    }
    if (!todo.empty())
        groups.push_back(todo);  // more groups than max_groups: the remainder becomes one last group
    return true;
}

}  // namespace

// host-only (no GPU): the grouping step with a caller-supplied probe, probe(i, r, user) = time of reading chunk i while
// writing chunk r; group_of[i] receives the group of chunk i, *ngroups the count (0: no contrast).  For the CPU tests and the
// measurement tools.
extern "C" int lumahip_pool_find_groups(int n, double (*probe)(int, int, void *), void *user, int *group_of, int *ngroups,
                                        double *fastest, int *nprobes)
{
    if (n < 1 || !probe || !group_of || !ngroups)
        return LUMAHIP_ERR_ARG;
    std::vector<std::vector<int>> groups;
    double fast = 0.0;
    int calls = 0;
    const bool ok = find_groups(n, [&](int i, int r) { calls++; return probe(i, r, user); }, groups, fast);
    for (int i = 0; i < n; i++)
        group_of[i] = -1;
    *ngroups = 0;
    if (fastest)
        *fastest = fast;
    if (nprobes)
        *nprobes = calls;
    if (!ok)
        return LUMAHIP_OK;
    *ngroups = (int)groups.size();
    for (size_t g = 0; g < groups.size(); g++)
        for (int i : groups[g])
            group_of[i] = (int)g;
    return LUMAHIP_OK;
}

struct lumahip_pool {
    int device = 0;
    size_t chunk_bytes = 0;
    std::vector<Chunk> chunks;  // kept chunks only, in hand-out order per kind
    std::string json;
    unsigned rot_next = 0;      // LUMAHIP_POOL_ROTATING: the group the next chunk should come from
    bool grouped = false;       // region groups were found and the chunks chosen by them
    int ngroups = 0;            // how many
};

extern "C" void lumahip_pool_destroy(lumahip_pool *pool)
{
    if (!pool)
        return;
    (void)hipSetDevice(pool->device);
    for (Chunk &c : pool->chunks)
        (void)hipFree(c.p);
    delete pool;
}

extern "C" int lumahip_pool_create(lumahip_ctx *ctx, const lumahip_pool_config *cfg, lumahip_pool **out)
{
    if (!ctx || !cfg || !out)
        return LUMAHIP_ERR_ARG;
    *out = nullptr;
    const size_t CB = cfg->chunk_bytes ? cfg->chunk_bytes : ((size_t)2 << 30);
    const size_t keep_free = cfg->keep_free_bytes ? cfg->keep_free_bytes : ((size_t)6 << 30);
    const int iters = cfg->probe_iters > 0 ? cfg->probe_iters : 2;
    const int n_float = std::max(0, cfg->n_float), n_y = std::max(0, cfg->n_y), n_uv = std::max(0, cfg->n_uv);
    const int n_striped = std::max(0, cfg->n_striped);
    const unsigned w = PROBE_W, h = PROBE_H, B = PROBE_FRAMES;
    const size_t n3 = (size_t)3 * w * h;
    // plane geometry of vpx_img_alloc(I42016, w, h, 32): stride = 32-aligned width x 2 bytes, chroma half
    const int st[3] = {(int)((w + 31) / 32 * 32 * 2), (int)((w + 31) / 32 * 32), (int)((w + 31) / 32 * 32)};
    const size_t psz[3] = {(size_t)h * st[0], (size_t)(h / 2) * st[1], (size_t)(h / 2) * st[2]};
    size_t offs[3], o = 0;
    for (int p = 0; p < 3; p++) {
        offs[p] = o;
        o = (o + B * psz[p] + ((size_t)1 << 20) - 1) >> 20 << 20;
    }
    if (B * n3 * 4 > CB || o > CB)
        return LUMAHIP_ERR_ARG;  // a chunk must hold one probe batch

    const int device = lumahip_device(ctx);
    if (hipSetDevice(device) != hipSuccess)
        return LUMAHIP_ERR_HIP;

    lumahip_pool *pool = new lumahip_pool();
    pool->device = device;
    pool->chunk_bytes = CB;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    long want = free_b > keep_free ? (long)((free_b - keep_free) / CB) : 0;
    if (cfg->max_chunks > 0 && want > cfg->max_chunks)
        want = cfg->max_chunks;
    std::vector<Chunk> all;
    for (long i = 0; i < want; i++) {
        Chunk c;
        if (hipMalloc(&c.p, CB) != hipSuccess) {
            (void)hipGetLastError();  // out of memory: use what we have
            break;
        }
        all.push_back(c);
    }
    const int n = (int)all.size();
    const int need = n_float + n_y + n_uv + 3 * n_striped;
    int nprobe = 0;
    bool failed = false;
    auto probe4 = [&](int i, int y, int u, int v) -> double {
        nprobe++;
        unsigned char *pl[3] = {all[y].p + offs[0], all[u].p + offs[1], all[v].p + offs[2]};
        float ms = 0.0f;
        if (lumahip_probe_encode_traffic_device(ctx, reinterpret_cast<const float *>(all[i].p), n3, B, w, h, pl, st, psz, iters, &ms) !=
            LUMAHIP_OK)
            failed = true;
        return ms;
    };
    char buf[1024];
    std::string js = "{";
    snprintf(buf, sizeof buf, "\"chunk_GiB\": %.3f, \"chunks\": %d, \"float_chunks\": %d, \"y_chunks\": %d, \"uv_chunks\": %d, "
                              "\"striped_chunks_per_group\": %d",
             CB / 1073741824.0, n, n_float, n_y, n_uv, n_striped);
    js += buf;

    auto keep = [&](int i, int kind, double t) {
        all[i].kind = kind;
        all[i].t = t;
    };
    bool grouped = false;
    if (n < need || n < 4) {
        js += ", \"grouped\": false, \"note\": \"not enough device memory for the chunk pool\"";
        for (Chunk &c : all)
            (void)hipFree(c.p);
        all.clear();
    } else {
        (void)probe4(1, 0, 0, 0);  // warm-up (first touch of the code object)
        // (2) groups
        std::vector<std::vector<int>> groups;
        double fast = 0.0;
        const bool contrast = find_groups(n, [&](int i, int r) { return failed ? 0.0 : probe4(i, r, r, r); }, groups, fast);
        if (contrast && !failed) {
            js += ", \"groups\": [";
            for (size_t g = 0; g < groups.size(); g++) {
                snprintf(buf, sizeof buf, "%s%zu", g ? ", " : "", groups[g].size());
                js += buf;
                for (int i : groups[g])
                    all[i].group = (int)g;
            }
            js += "]";
        }
        // (3) which group holds what
        int gy = -1, g_in = -1, g_uv = -1;
        if (contrast && !failed && groups.size() >= 2) {
            std::vector<int> order(groups.size());
            for (size_t g = 0; g < groups.size(); g++)
                order[g] = (int)g;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return groups[a].size() < groups[b].size(); });
            for (int g : order)
                if ((int)groups[g].size() >= n_y + 1) {
                    gy = g;
                    break;
                }
            if (gy >= 0) {
                std::vector<int> rest;
                for (auto it = order.rbegin(); it != order.rend(); ++it)
                    if (*it != gy)
                        rest.push_back(*it);  // largest first
                g_in = rest.front();
                g_uv = (int)groups[rest.back()].size() >= n_uv + 1 ? rest.back() : g_in;
            }
        }
        std::vector<int> ycand, uvcand;
        int cref = -1;
        if (gy >= 0) {
            cref = groups[g_in][0];
            const size_t ny = (size_t)std::max(n_y + 6, 12), nuv = (size_t)std::max(n_uv + 5, 8);
            for (int i : groups[gy])
                if (ycand.size() < ny)
                    ycand.push_back(i);
            for (int i : groups[g_uv])
                if (i != cref && uvcand.size() < nuv)
                    uvcand.push_back(i);
        }
        if (gy >= 0 && (int)ycand.size() >= std::max(n_y, 1) && (int)uvcand.size() >= std::max(n_uv, 1) && !failed) {
            grouped = true;
            std::map<int, double> ty, tu, tf;
            for (int k : ycand)
                ty[k] = probe4(cref, k, uvcand[0], uvcand[0]);
            std::sort(ycand.begin(), ycand.end(), [&](int a, int b) { return ty[a] < ty[b]; });
            std::vector<int> ysel(ycand.begin(), ycand.begin() + n_y);
            const int yref = ycand[0];
            for (int k : uvcand)
                tu[k] = probe4(cref, yref, k, k);
            std::sort(uvcand.begin(), uvcand.end(), [&](int a, int b) { return tu[a] < tu[b]; });
            std::vector<int> uvsel(uvcand.begin(), uvcand.begin() + n_uv);
            const int uvref = uvcand[0];
            // (4) every other chunk as a float chunk against the chosen planes chunks
            std::vector<int> cand;
            for (int i = 0; i < n; i++)
                if (std::find(ysel.begin(), ysel.end(), i) == ysel.end() && std::find(uvsel.begin(), uvsel.end(), i) == uvsel.end())
                    cand.push_back(i);
            for (int i : cand)
                tf[i] = probe4(i, yref, uvref, uvref);
            std::sort(cand.begin(), cand.end(), [&](int a, int b) { return tf[a] < tf[b]; });
            const int nf = std::min<int>(n_float, (int)cand.size());
            for (int k : ysel)
                keep(k, LUMAHIP_POOL_Y, ty[k]);
            for (int k : uvsel)
                keep(k, LUMAHIP_POOL_UV, tu[k]);
            std::vector<double> tkept, trej;
            for (int j = 0; j < (int)cand.size(); j++) {
                if (j < nf) {
                    keep(cand[j], LUMAHIP_POOL_FLOAT, tf[cand[j]]);
                    tkept.push_back(tf[cand[j]]);
                } else {
                    trej.push_back(tf[cand[j]]);
                }
            }
            // (5) striped chunks: n_striped unassigned chunks from each of the first three groups
            if (n_striped > 0)
                for (int g = 0; g < 3 && g < (int)groups.size(); g++) {
                    int got = 0;
                    for (int i : groups[g])
                        if (all[i].kind < 0 && got < n_striped) {
                            keep(i, LUMAHIP_POOL_STRIPED, 0.0);
                            got++;
                        }
                }
            const double same_t = groups[gy].size() >= 2 ? probe4(groups[gy][0], groups[gy][1], groups[gy][1], groups[gy][1]) : 0.0;
            double ymax = 0.0, uvmax = 0.0;
            for (int k : ysel)
                ymax = std::max(ymax, ty[k]);
            for (int k : uvsel)
                uvmax = std::max(uvmax, tu[k]);
            snprintf(buf, sizeof buf,
                     ", \"grouped\": true, \"y_group\": %d, \"uv_group\": %d, \"probe_ms\": {\"input_and_planes_in_one_group\": %.4f, "
                     "\"planes_together_in_another_group\": %.4f, \"float_chunks_kept_fastest\": %.4f, \"float_chunks_kept_median\": %.4f, "
                     "\"float_chunks_kept_slowest\": %.4f, \"float_chunks_rejected_median\": %.4f, \"y_chunks_kept_slowest\": %.4f, "
                     "\"uv_chunks_kept_slowest\": %.4f}",
                     gy, g_uv, same_t, fast, tkept.empty() ? 0.0 : tkept.front(), median(tkept), tkept.empty() ? 0.0 : tkept.back(),
                     median(trej), ymax, uvmax);
            js += buf;
        } else {
            // no contrast, no usable group or a failed probe: plain choice (the first chunks), reported as such
            js += ", \"grouped\": false";
            if (failed)
                js += ", \"note\": \"a traffic probe failed\"";
            int i = 0;
            for (int k = 0; k < n_float && i < n; k++)
                keep(i++, LUMAHIP_POOL_FLOAT, 0.0);
            for (int k = 0; k < n_y && i < n; k++)
                keep(i++, LUMAHIP_POOL_Y, 0.0);
            for (int k = 0; k < n_uv && i < n; k++)
                keep(i++, LUMAHIP_POOL_UV, 0.0);
            for (int k = 0; k < 3 * n_striped && i < n; k++) {
                all[i].group = k % 3;  // nominal: alloc(group) still finds n_striped per "group"
                keep(i++, LUMAHIP_POOL_STRIPED, 0.0);
            }
        }
        (void)lumahip_sync(ctx);
        (void)hipDeviceSynchronize();
        // hand-out order: fastest first within a kind
        std::vector<Chunk> kept;
        for (int kind = 0; kind <= LUMAHIP_POOL_STRIPED; kind++) {
            std::vector<Chunk> k;
            for (Chunk &c : all)
                if (c.kind == kind)
                    k.push_back(c);
            std::stable_sort(k.begin(), k.end(), [](const Chunk &a, const Chunk &b) { return a.t < b.t; });
            kept.insert(kept.end(), k.begin(), k.end());
        }
        for (Chunk &c : all)
            if (c.kind < 0)
                (void)hipFree(c.p);
        pool->chunks = kept;
    }
    pool->grouped = grouped;
    for (const Chunk &c : pool->chunks)
        if (grouped && c.group + 1 > pool->ngroups)
            pool->ngroups = c.group + 1;
    snprintf(buf, sizeof buf, ", \"probes\": %d}", nprobe);
    js += buf;
    pool->json = js;
    *out = pool;
    return LUMAHIP_OK;
}

// The SMALL pool (round 6; profiles/r06_small_pool.txt).  Finding the groups does not need all of the device's memory: eight 2 GiB
// chunks taken side by side already fall into two of them, and a read stream in one with its write streams in the other is what
// the placed rate needs (0.734 - 0.737 of the roofline ordered for a 40-frame 4K stream in 2 + 1 + 1 chunks, against 0.67 - 0.68
// in plain allocations, same process; 30 ms and 27 probes instead of 5.7 s and 427).  At six chunks the first round sometimes
// sees one group only, hence the floor of eight and one retry with twice as many.  STRIPED chunks (n_striped > 0: decoded output
// spread over THREE groups) need the third group, which side-by-side allocations reach only after a few dozen chunks: 48 then.
extern "C" int lumahip_pool_create_small(lumahip_ctx *ctx, int n_float, int n_y, int n_uv, int n_striped, lumahip_pool **out)
{
    if (!ctx || !out || n_float < 0 || n_y < 0 || n_uv < 0 || n_striped < 0)
        return LUMAHIP_ERR_ARG;
    const int need = n_float + n_y + n_uv + 3 * n_striped;
    const int first = n_striped > 0 ? std::max(48, need + 8) : std::max(8, need + 4);
    int rc = LUMAHIP_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        lumahip_pool_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.n_float = n_float;
        cfg.n_y = n_y;
        cfg.n_uv = n_uv;
        cfg.n_striped = n_striped;
        cfg.max_chunks = first << attempt;
        rc = lumahip_pool_create(ctx, &cfg, out);
        if (rc != LUMAHIP_OK || attempt == 1 || ((*out)->grouped && (n_striped == 0 || (*out)->ngroups >= 3)))
            break;
        lumahip_pool_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

// ---- decoded batches in buffers the LIBRARY places (include/lumahip.h "lumahip_decoded_ring") --------------------------------
// One decode launch writes 12 of its 15 bytes per pixel.  A batch of packed LumaFrames in ONE caller-owned buffer puts all of
// that into one region group (0.69 of the roofline, whatever the buffer); the frames of a batch rotating over three buffers of
// three groups reach 0.74 (lumahip_decode_frames_device_rotating).  A caller that lets the library allocate gets exactly that:
// the ring takes three-chunk sets from a small pool (one chunk per group), carves every batch's three thirds out of them and
// hands out frame pointers; every frame is a packed LumaFrame.  No pool (tiny GPU share, no contrast): three plain allocations
// per batch, same layout, same results.
struct lumahip_decoded_ring {
    lumahip_ctx *ctx = nullptr;
    lumahip_pool *pool = nullptr;
    unsigned nbatches = 0, nframes = 0, w = 0, h = 0;
    size_t frame_stride = 0;                  // floats
    std::vector<float *> base[3];             // per batch
    std::vector<void *> plain;                // plain allocations (fallback), freed on destroy
    bool placed = false;
};

extern "C" void lumahip_decoded_ring_destroy(lumahip_decoded_ring *r)
{
    if (!r)
        return;
    for (void *p : r->plain)
        (void)lumahip_free(r->ctx, p);
    lumahip_pool_destroy(r->pool);
    delete r;
}

extern "C" int lumahip_decoded_ring_create(lumahip_ctx *ctx, unsigned nbatches, unsigned nframes, unsigned w, unsigned h,
                                           lumahip_decoded_ring **out)
{
    if (!ctx || !out || nbatches == 0 || nframes == 0 || w == 0 || h == 0 || (w & 1) || (h & 1))
        return LUMAHIP_ERR_ARG;
    *out = nullptr;
    lumahip_decoded_ring *r = new lumahip_decoded_ring();
    r->ctx = ctx;
    r->nbatches = nbatches;
    r->nframes = nframes;
    r->w = w;
    r->h = h;
    r->frame_stride = (size_t)3 * w * h;       // w, h even: a multiple of 4 floats, frames stay 16-byte aligned
    const size_t CB = (size_t)2 << 30;
    const size_t third = (((size_t)(nframes + 2) / 3) * r->frame_stride * sizeof(float) + ((size_t)1 << 20) - 1) >> 20 << 20;
    for (int k = 0; k < 3; k++)
        r->base[k].assign(nbatches, nullptr);
    if (third <= CB) {
        const unsigned per_chunk = (unsigned)(CB / third);
        const int sets = (int)((nbatches + per_chunk - 1) / per_chunk);
        lumahip_pool *pool = nullptr;
        if (lumahip_pool_create_small(ctx, 0, 0, 0, sets, &pool) == LUMAHIP_OK && pool && pool->grouped && pool->ngroups >= 3 &&
            lumahip_pool_available(pool, LUMAHIP_POOL_STRIPED, 0) >= sets && lumahip_pool_available(pool, LUMAHIP_POOL_STRIPED, 1) >= sets &&
            lumahip_pool_available(pool, LUMAHIP_POOL_STRIPED, 2) >= sets) {
            std::vector<unsigned char *> chunk[3];
            bool ok = true;
            for (int g = 0; g < 3 && ok; g++)
                for (int s = 0; s < sets && ok; s++) {
                    void *p = nullptr;
                    ok = lumahip_pool_alloc(pool, LUMAHIP_POOL_STRIPED, g, &p) == LUMAHIP_OK;
                    chunk[g].push_back((unsigned char *)p);
                }
            if (ok) {
                // batch b: its three thirds in the three groups, the group of buffer 0 walking with b so that consecutive
                // batches in flight start in different groups
                for (unsigned b = 0; b < nbatches; b++)
                    for (int k = 0; k < 3; k++)
                        r->base[k][b] = (float *)(chunk[(k + b) % 3][b / per_chunk] + (size_t)(b % per_chunk) * third);
                r->pool = pool;
                r->placed = true;
            }
        }
        if (!r->placed)
            lumahip_pool_destroy(pool);
    }
    if (!r->placed) {
        for (unsigned b = 0; b < nbatches; b++)
            for (int k = 0; k < 3; k++) {
                void *p = nullptr;
                if (lumahip_malloc(ctx, &p, third) != LUMAHIP_OK) {
                    lumahip_decoded_ring_destroy(r);
                    return LUMAHIP_ERR_HIP;
                }
                r->plain.push_back(p);
                r->base[k][b] = (float *)p;
            }
    }
    *out = r;
    return LUMAHIP_OK;
}

extern "C" int lumahip_decoded_ring_info(const lumahip_decoded_ring *r, int info[4], size_t *frame_stride)
{
    if (!r || !info)
        return LUMAHIP_ERR_ARG;
    info[0] = r->placed ? 1 : 0;
    info[1] = (int)r->nbatches;
    info[2] = (int)r->nframes;
    info[3] = r->pool ? r->pool->ngroups : 0;
    if (frame_stride)
        *frame_stride = r->frame_stride;
    return LUMAHIP_OK;
}

extern "C" float *lumahip_decoded_ring_frame(const lumahip_decoded_ring *r, unsigned batch, unsigned frame)
{
    if (!r || batch >= r->nbatches || frame >= r->nframes)
        return nullptr;
    return r->base[frame % 3][batch] + (size_t)(frame / 3) * r->frame_stride;
}

extern "C" int lumahip_decode_frames_device_ring(lumahip_ctx *ctx, const unsigned char *const planes_dev[3], const int stride[3],
                                                 const size_t plane_frame_stride[3], unsigned nframes, int profile, float sc,
                                                 lumahip_decoded_ring *r, unsigned batch)
{
    if (!ctx || !r || batch >= r->nbatches || nframes == 0 || nframes > r->nframes)
        return LUMAHIP_ERR_ARG;
    float *const bases[3] = {r->base[0][batch], r->base[1][batch], r->base[2][batch]};
    return lumahip_decode_frames_device_rotating(ctx, planes_dev, stride, plane_frame_stride, nframes, r->w, r->h, profile, sc, bases,
                                                 r->frame_stride);
}

extern "C" int lumahip_pool_alloc(lumahip_pool *pool, int kind, int group, void **chunk)
{
    if (!pool || !chunk)
        return LUMAHIP_ERR_ARG;
    *chunk = nullptr;
    if (kind == LUMAHIP_POOL_ROTATING) {
        // consecutive allocations walk the region groups 0, 1, 2, 0, ... (the chunks kept for LUMAHIP_POOL_STRIPED): a caller that
        // gives consecutive batches of PACKED frames consecutive allocations has launches in flight that write different groups
        // without knowing about groups.  `group` >= 0 restarts the walk at that group.  A group that has run out is skipped.
        if (group >= 0)
            pool->rot_next = (unsigned)group;
        for (int tries = 0; tries < 3; tries++) {
            const int g = (int)(pool->rot_next++ % 3u);
            for (Chunk &c : pool->chunks)
                if (!c.out && c.kind == LUMAHIP_POOL_STRIPED && c.group == g) {
                    c.out = true;
                    *chunk = c.p;
                    return LUMAHIP_OK;
                }
        }
        return LUMAHIP_ERR_STATE;
    }
    for (Chunk &c : pool->chunks)
        if (!c.out && c.kind == kind && (group < 0 || c.group == group)) {
            c.out = true;
            *chunk = c.p;
            return LUMAHIP_OK;
        }
    return LUMAHIP_ERR_STATE;
}

extern "C" int lumahip_pool_release(lumahip_pool *pool, void *chunk)
{
    if (!pool || !chunk)
        return LUMAHIP_ERR_ARG;
    for (Chunk &c : pool->chunks)
        if (c.p == chunk && c.out) {
            c.out = false;
            return LUMAHIP_OK;
        }
    return LUMAHIP_ERR_ARG;
}

extern "C" int lumahip_pool_available(const lumahip_pool *pool, int kind, int group)
{
    if (!pool)
        return 0;
    int k = 0;
    if (kind == LUMAHIP_POOL_ROTATING) {   // (they are the striped chunks; `group` does not restrict what a rotating allocation may fall back to)
        kind = LUMAHIP_POOL_STRIPED;
        group = -1;
    }
    for (const Chunk &c : pool->chunks)
        if (!c.out && c.kind == kind && (group < 0 || c.group == group))
            k++;
    return k;
}

extern "C" int lumahip_pool_group_of(const lumahip_pool *pool, const void *chunk)
{
    if (!pool)
        return -1;
    for (const Chunk &c : pool->chunks)
        if (c.p == chunk)
            return c.group;
    return -1;
}

extern "C" const char *lumahip_pool_stats_json(const lumahip_pool *pool) { return pool ? pool->json.c_str() : "{}"; }
