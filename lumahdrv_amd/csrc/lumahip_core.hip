// lumahip_core.hip -- context life cycle, quantizer upload, launch-geometry rules, memory helpers of include/lumahip.h.
// No kernels here; the arithmetic is in luma_device.hpp / luma_kernels.hpp.
#include "lumahip_internal.hpp"

using namespace lh;
using namespace lhost;

int lumahip_fail(lumahip_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

extern "C" int lumahip_abi_version(void) { return LUMAHIP_ABI_VERSION; }

extern "C" int lumahip_device_count(int *count)
{
    if (!count)
        return LUMAHIP_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return LUMAHIP_ERR_HIP;
    }
    *count = n;
    return LUMAHIP_OK;
}

extern "C" int lumahip_create(lumahip_ctx **out, int device)
{
    if (!out)
        return LUMAHIP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return LUMAHIP_ERR_HIP;  // no CPU fallback: the path needs a HIP device
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess)
            return LUMAHIP_ERR_HIP;
    }
    if (device >= n)
        return LUMAHIP_ERR_ARG;
    lumahip_ctx *c = new lumahip_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return LUMAHIP_ERR_HIP;
    }
    c->stream = c->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        c->num_cu = prop.multiProcessorCount;
    // Measurement overrides (launch geometry, LDS limit, literal search) are lumahip_tune() calls; the LUMAHIP_* environment
    // variables of the same names are honoured only under LUMAHIP_TUNING=1, so that a stray variable cannot change how a
    // production process launches.
    if (const char *g = getenv("LUMAHIP_TUNING"); g && atoi(g) != 0) {
        static const char *const keys[][2] = {{"LUMAHIP_BLOCK", "block"}, {"LUMAHIP_BLOCKS_PER_CU", "blocks_per_cu"},
                                              {"LUMAHIP_GRID_ENC", "grid_enc"}, {"LUMAHIP_GRID_DEC", "grid_dec"},
                                              {"LUMAHIP_ALLOW_ALIASED_FRAMES", "allow_aliased_frames"},
                                              {"LUMAHIP_LDS_TABLE_MAX_KB", "lds_table_max_kb"},
                                              {"LUMAHIP_FORCE_LITERAL", "force_literal"}, {"LUMAHIP_LANES", "lanes"},
                                              {"LUMAHIP_LANE_GRID_ENC", "lane_grid_enc"}, {"LUMAHIP_LANE_GRID_DEC", "lane_grid_dec"},
                                              {"LUMAHIP_COPY_THREADS", "copy_threads"}, {"LUMAHIP_HOST_BANDS", "host_bands"}, {"LUMAHIP_BAND_TAPER", "band_taper"}, {"LUMAHIP_COPY_SPIN", "copy_spin"},
                                              {"LUMAHIP_YCBCR_TABLES", "ycbcr_tables"}, {"LUMAHIP_HALF_TABLE", "half_table"}, {"LUMAHIP_NUMA", "numa"}, {"LUMAHIP_NUMA_NODE", "numa_node"}, {"LUMAHIP_HALF_UPLOAD", "half_upload"}};
        for (const auto &k : keys)
            if (const char *e = getenv(k[0]))
                (void)lumahip_tune(c, k[1], atol(e));
    }
    *out = c;
    return LUMAHIP_OK;
}

extern "C" void lumahip_destroy(lumahip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_rec);
    (void)hipFree(c->d_rec_y);
    (void)hipFree(c->d_ytab);
    for (auto &t : c->half_tabs)
        (void)hipFree(t.d);
    for (auto &t : c->rb_tabs)
        (void)hipFree(t.d);
    lag_policy_destroy(c->half_pol);
    lag_policy_destroy(c->rb_pol);
    (void)hipFree(c->d_frame);
    (void)hipFree(c->d_planes);
    (void)hipFree(c->d_stats);
    (void)hipFree(c->d_stats_part);
    (void)hipFree(c->d_arr);
    for (auto &sl : c->slot) {
        (void)hipFree(sl.d_frame);
        (void)hipFree(sl.d_planes);
        (void)hipFree(sl.d_stats);
        if (sl.h2d) (void)hipEventDestroy(sl.h2d);
        if (sl.kern) (void)hipEventDestroy(sl.kern);
        if (sl.d2h) (void)hipEventDestroy(sl.d2h);
    }
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    for (int i = 0; i < lumahip_ctx::MAX_BANDS; i++) {
        if (c->band_h2d[i]) (void)hipEventDestroy(c->band_h2d[i]);
        if (c->band_kern[i]) (void)hipEventDestroy(c->band_kern[i]);
    }
    (void)hipFree(c->d_band_stats);
    auto drop_stage = [](lumahip_ctx::Stage *st) {
        if (st->ev) (void)hipEventSynchronize(st->ev);
        if (st->h) (void)hipHostFree(st->h);
        if (st->ev) (void)hipEventDestroy(st->ev);
    };
    for (int i = 0; i < lumahip_ctx::N_STAGE; i++)
        drop_stage(&c->stage_up[i]);
    for (int i = 0; i < lumahip_ctx::N_STAGE_DN; i++)
        drop_stage(&c->stage_dn[i]);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_es_stats) (void)hipHostFree(c->h_es_stats);
    lumahip_copy_pool_destroy(c->copy_pool);
    if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d);
    if (c->s_kern) (void)hipStreamDestroy(c->s_kern);
    if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h);
    for (int i = 0; i < LUMAHIP_MAX_LANES; i++) {
        if (c->lane_stream[i]) (void)hipStreamDestroy(c->lane_stream[i]);
        if (c->lane_done[i]) (void)hipEventDestroy(c->lane_done[i]);
    }
    if (c->lane_fork) (void)hipEventDestroy(c->lane_fork);
    if (c->own_stream)
        (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char *lumahip_last_error(const lumahip_ctx *c) { return c ? c->err.c_str() : "null context"; }

extern "C" int lumahip_device(const lumahip_ctx *c) { return c ? c->device : -1; }

extern "C" int lumahip_set_stream(lumahip_ctx *c, void *s)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (c->lanes_active)
        return fail(c, LUMAHIP_ERR_STATE, "close the unordered section before changing the stream");
    c->stream = (hipStream_t)s;  // NULL is a valid handle: the device's default (null) stream
    return LUMAHIP_OK;
}

extern "C" int lumahip_reset_stream(lumahip_ctx *c)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (c->lanes_active)
        return fail(c, LUMAHIP_ERR_STATE, "close the unordered section before changing the stream");
    c->stream = c->own_stream;
    return LUMAHIP_OK;
}

extern "C" int lumahip_sync(lumahip_ctx *c)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->lanes_active; i++)   // inside an unordered section: its lanes too (include/lumahip.h)
        HIPCHK(c, hipStreamSynchronize(c->lane_stream[i]));
    return LUMAHIP_OK;
}

// ---- unordered sections ------------------------------------------------------------------------------------------
// One long launch leaves the memory system under-used while it ramps up (table staging, the first loads) and while its
// last workgroups finish; back-to-back launches on one stream pay that at every boundary.  Batches of frames are
// independent, so inside a section successive launches go to different streams and one batch's ramp / tail overlaps its
// neighbours' steady state (profiles/r02_concurrent_launches.txt: +4 % encode, +7 % decode).
extern "C" int lumahip_begin_unordered(lumahip_ctx *c, int lanes)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (c->lanes_active)
        return fail(c, LUMAHIP_ERR_STATE, "lumahip_begin_unordered: a section is already open");
    if (lanes == 0)
        lanes = c->lanes_default > 0 ? c->lanes_default : 2;
    if (lanes < 1 || lanes > LUMAHIP_MAX_LANES)
        return fail(c, LUMAHIP_ERR_ARG, "lanes must be 1..%d (0 = default)", LUMAHIP_MAX_LANES);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->lane_fork)
        HIPCHK(c, hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming));
    for (int i = 0; i < lanes; i++) {
        if (!c->lane_stream[i]) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->lane_stream[i], hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->lane_done[i], hipEventDisableTiming));
        }
    }
    // everything queued on the context's stream so far happens before the section
    HIPCHK(c, hipEventRecord(c->lane_fork, c->stream));
    for (int i = 0; i < lanes; i++)
        HIPCHK(c, hipStreamWaitEvent(c->lane_stream[i], c->lane_fork, 0));
    c->lanes_active = lanes;
    c->lane_next = 0;
    return LUMAHIP_OK;
}

extern "C" int lumahip_end_unordered(lumahip_ctx *c)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!c->lanes_active)
        return fail(c, LUMAHIP_ERR_STATE, "lumahip_end_unordered: no section is open");
    HIPCHK(c, hipSetDevice(c->device));
    const int lanes = c->lanes_active;
    c->lanes_active = 0;
    // everything queued on the context's stream from now on happens after the section
    for (int i = 0; i < lanes; i++) {
        HIPCHK(c, hipEventRecord(c->lane_done[i], c->lane_stream[i]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->lane_done[i], 0));
    }
    return LUMAHIP_OK;
}

static int requantize(lumahip_ctx *c);

extern "C" int lumahip_tune(lumahip_ctx *c, const char *key, long v)
{
    if (!c || !key)
        return LUMAHIP_ERR_ARG;
    const std::string k(key);
    // the keys that rebuild the device tables: not under frames pushed with the stream entry points (as lumahip_set_quantizer)
    if ((k == "lds_table_max_kb" || k == "force_literal" || k == "ycbcr_tables" || k == "lin_index") && c->es_head != c->es_tail)
        return fail(c, LUMAHIP_ERR_STATE, "frames pushed with lumahip_encode_stream_push / lumahip_decode_stream_push are still in flight: pop them before '%s'", key);
    if (k == "block") {
        if (v == 0) {
            c->block_threads = 256;
            c->block_forced = false;
        } else if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) {
            c->block_threads = (int)v;
            c->block_forced = true;
        } else {
            return fail(c, LUMAHIP_ERR_ARG, "block must be 0 (default), 64, 128, 256, 512 or 1024");
        }
    } else if (k == "blocks_per_cu") {
        c->blocks_per_cu = v > 0 ? (int)v : 0;
    } else if (k == "grid_enc") {
        c->grid_override[0] = v > 0 ? v : 0;
    } else if (k == "grid_dec") {
        c->grid_override[1] = v > 0 ? v : 0;
    } else if (k == "allow_aliased_frames") {
        c->allow_alias = v != 0;
    } else if (k == "lds_table_max_kb") {
        if (v < 0)
            c->lds_table_max = LUMAHIP_LDS_TABLE_MAX_DEFAULT;
        else if (v <= 152)
            c->lds_table_max = (size_t)v * 1024;
        else
            return fail(c, LUMAHIP_ERR_ARG, "lds_table_max_kb must be 0..152 (or -1 for the default)");
        if (c->have_quant)
            return requantize(c);
    } else if (k == "force_literal") {
        c->force_literal = v != 0;
        if (c->have_quant)
            return requantize(c);
    } else if (k == "ycbcr_tables") {
        c->use_ycbcr_tables = v != 0;
        if (c->have_quant)
            return requantize(c);
    } else if (k == "ycbcr_rb_tables") {
        if (v < 0 || v > 2)
            return fail(c, LUMAHIP_ERR_ARG, "ycbcr_rb_tables must be 0 (six powf per pixel), 1 (red and blue from the per-stream tables where a wave's codes are close) or 2 (always)");
        c->rb_mode = (int)v;
        lag_policy_reset(c->rb_pol);
    } else if (k == "test_fail_rb_alloc") {   // tests: the next allocation of the red / blue tables behaves as if hipMalloc had failed
        c->test_fail_rb_alloc = v != 0;
    } else if (k == "dec_vw") {
        c->dec_vw = v == 2 ? 2 : 0;
    } else if (k == "rb_near_y") {
        c->rb_near_y = (int)std::max<long>(0, std::min<long>(v, 1 << 24));
    } else if (k == "rb_near_c") {
        c->rb_near_c = (int)std::max<long>(0, std::min<long>(v, 1 << 24));
    } else if (k == "lin_index") {   // 0: tables whose float-bit records miss LDS keep them in global memory, as before round 5 (A/B, tests)
        c->use_lin_index = v != 0;
        if (c->have_quant)
            return requantize(c);
    } else if (k == "numa" || k == "numa_node") {
        // takes effect for what is allocated / started from now on: set it before the first host call of a context
        if (k == "numa")
            c->numa_mode = (v >= 0 && v <= 3) ? (int)v : 2;   // 1: rings and threads, 2: the rings only (default), 3: the threads only
        else
            c->numa_force_node = v >= 0 ? (int)std::min<long>(v, 1023) : -1;
        c->numa_resolved = false;
        lumahip_copy_pool_destroy(c->copy_pool);
        c->copy_pool = nullptr;
    } else if (k == "half_upload") {
        if (v < 0 || v > 2)
            return fail(c, LUMAHIP_ERR_ARG, "half_upload must be 0 (never), 1 (while the frames hold binary16 values) or 2 (always try)");
        c->in16_mode = (int)v;
        c->in16_backoff = c->in16_backoff_len = 0;
    } else if (k == "half_table") {
        if (v < 0 || v > 2)
            return fail(c, LUMAHIP_ERR_ARG, "half_table must be 0 (off), 1 (while the stream is binary16 data) or 2 (always)");
        c->half_mode = (int)v;
        lag_policy_reset(c->half_pol);
    } else if (k == "host_bands") {
        if (v < 1 || v > lumahip_ctx::MAX_BANDS)
            return fail(c, LUMAHIP_ERR_ARG, "host_bands must be 1..%d", lumahip_ctx::MAX_BANDS);
        c->host_bands = (int)v;
    } else if (k == "band_taper") {
        if (v < 10 || v > 100)
            return fail(c, LUMAHIP_ERR_ARG, "band_taper must be 10..100 (per cent)");
        c->band_taper = (int)v;
    } else if (k == "copy_spin") {
        c->copy_spin = v > 0 ? (int)std::min<long>(v, 10000000) : 0;
        lumahip_copy_pool_destroy(c->copy_pool);
        c->copy_pool = nullptr;
    } else if (k == "copy_threads") {
        if (v < 0 || v > 32)
            return fail(c, LUMAHIP_ERR_ARG, "copy_threads must be 0..32");
        lumahip_copy_pool_destroy(c->copy_pool);
        c->copy_pool = nullptr;
        c->copy_threads = (int)v;
    } else if (k == "lane_grid_enc") {
        c->lane_grid[0] = v > 0 ? v : 0;
    } else if (k == "lane_grid_dec") {
        c->lane_grid[1] = v > 0 ? v : 0;
    } else if (k == "lanes") {
        if (v < 0 || v > LUMAHIP_MAX_LANES)
            return fail(c, LUMAHIP_ERR_ARG, "lanes must be 0..%d", LUMAHIP_MAX_LANES);
        c->lanes_default = (int)v;
    } else {
        return fail(c, LUMAHIP_ERR_ARG, "unknown tuning key '%s'", key);
    }
    return LUMAHIP_OK;
}

// ---------------------------------------------------------------------------------------- quantizer

// The search index of a table is a pure function of the table, and several contexts of one process usually hold the same
// table (one per GPU in the multi-device layer, encoder + decoder of a transcoder): built once, shared.
namespace {
struct IndexCacheEntry {
    std::vector<float> lut;
    std::shared_ptr<const ThreshIndex> ix;
};
std::mutex g_index_mutex;
std::vector<IndexCacheEntry> g_index_cache;  // a handful of entries, most recent last

std::shared_ptr<const ThreshIndex> cached_thresh_index(const std::vector<float> &lut)
{
    std::lock_guard<std::mutex> lk(g_index_mutex);
    for (auto &e : g_index_cache)
        if (e.lut.size() == lut.size() && memcmp(e.lut.data(), lut.data(), lut.size() * sizeof(float)) == 0)
            return e.ix;
    IndexCacheEntry e;
    e.lut = lut;
    e.ix = std::make_shared<const ThreshIndex>(build_thresh_index(lut.data(), (int)lut.size(), 1 << 19));
    if (g_index_cache.size() >= 8)
        g_index_cache.erase(g_index_cache.begin());
    g_index_cache.push_back(e);
    return e.ix;
}
struct LinCacheEntry {
    std::vector<float> lut;
    std::shared_ptr<const LinIndex> ix;
};
std::vector<LinCacheEntry> g_lin_cache;

// value-keyed records (lut_index.hpp LinIndex) of a table whose float-bit records do not fit LDS
std::shared_ptr<const LinIndex> cached_lin_index(const std::vector<float> &lut)
{
    std::lock_guard<std::mutex> lk(g_index_mutex);
    for (auto &e : g_lin_cache)
        if (e.lut.size() == lut.size() && memcmp(e.lut.data(), lut.data(), lut.size() * sizeof(float)) == 0)
            return e.ix;
    LinCacheEntry e;
    e.lut = lut;
    e.ix = std::make_shared<const LinIndex>(build_lin_index(lut.data(), (int)lut.size(), 1 << 15));
    if (g_lin_cache.size() >= 8)
        g_lin_cache.erase(g_lin_cache.begin());
    g_lin_cache.push_back(e);
    return e.ix;
}
struct YIndexCacheEntry {
    std::vector<float> lut;
    float Lmax;
    std::shared_ptr<const ThreshIndex> ix;
};
std::vector<YIndexCacheEntry> g_yindex_cache;

// records of the YCbCr composite  t -> search(PQdec(t / 255)), t = 219 y + 16  (host_lut.cpp ycbcr_luma_code_host), per (table, Lmax)
std::shared_ptr<const ThreshIndex> cached_ycbcr_index(const std::vector<float> &lut, float Lmax)
{
    std::lock_guard<std::mutex> lk(g_index_mutex);
    for (auto &e : g_yindex_cache)
        if (e.lut.size() == lut.size() && memcmp(&e.Lmax, &Lmax, sizeof(float)) == 0 &&
            memcmp(e.lut.data(), lut.data(), lut.size() * sizeof(float)) == 0)
            return e.ix;
    YIndexCacheEntry e;
    e.lut = lut;
    e.Lmax = Lmax;
    const int maxVal = (int)lut.size() - 1;
    e.ix = std::make_shared<const ThreshIndex>(build_thresh_index_fn(
        [&](float t) { return ycbcr_luma_code_host(t, lut.data(), maxVal, Lmax); }, maxVal, 1 << 16, true));
    if (g_yindex_cache.size() >= 8)
        g_yindex_cache.erase(g_yindex_cache.begin());
    g_yindex_cache.push_back(e);
    return e.ix;
}
}  // namespace

// device copy of the table + the decode-side decisions: everything a decoder needs
static int upload_table(lumahip_ctx *c)
{
    const size_t n = c->h_lut.size();
    const size_t powf_b = (c->q.cs == CS_YCBCR) ? sizeof(PowfTablesWide) : 0;
    // decode side: luminance table (+ Lu'v' chroma table, + the powf tables for YCbCr) staged in LDS
    c->lut_in_lds = c->bitdepthC <= 12 && (n + 4) * sizeof(float) <= std::max<size_t>(c->lds_table_max, 16 * 1024 + 16) &&
                    (n + 4) * sizeof(float) + ((size_t)4 << c->bitdepthC) + 64 + powf_b <= LUMAHIP_LDS_PER_WORKGROUP;
    const size_t lut_floats = (n + 1 + 3) & ~(size_t)3;  // NaN padding up to a multiple of 16 bytes
    std::vector<float> padded(lut_floats, __builtin_nanf(""));
    memcpy(padded.data(), c->h_lut.data(), n * sizeof(float));
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_rec);
    (void)hipFree(c->d_rec_y);
    (void)hipFree(c->d_ytab);
    c->d_lut = nullptr;
    c->d_rec = nullptr;
    c->d_rec_y = nullptr;
    c->d_ytab = nullptr;
    for (auto &t : c->rb_tabs)     // (the red / blue tables of the YCbCr decode kernels were built from the old y table)
        (void)hipFree(t.d);
    c->rb_tabs.clear();
    c->rb_unavailable = false;
    c->tix_y.reset();
    HIPCHK(c, hipMalloc(&c->d_lut, lut_floats * sizeof(float)));
    HIPCHK(c, hipMemcpy(c->d_lut, padded.data(), lut_floats * sizeof(float), hipMemcpyHostToDevice));
    QuantDev &q = c->q;
    q.lut = c->d_lut;
    q.rec = nullptr;
    q.ytab = nullptr;
    if (c->q.cs == CS_YCBCR && c->use_ycbcr_tables && c->lut_in_lds) {
        // YCbCr decode: the first PQ evaluation of a pixel depends on its luminance code only -- one table per stream, built
        // with the host libm (the function the reference calls).  Only for tables of finite non-negative values: the
        // kernels' range analysis of what follows (luma_device.hpp ycbcr_inv) assumes them.
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++)
            ok = c->h_lut[i] >= 0.0f && c->h_lut[i] <= 3.0e38f;
        const size_t both = 2 * (((n + 4) * sizeof(float) + 15) & ~(size_t)15) + ((size_t)8 << c->bitdepthC) + 64 + powf_b;   // + the two chroma-term tables
        if (ok && both <= LUMAHIP_LDS_PER_WORKGROUP) {
            std::vector<float> yt(lut_floats, 0.0f);
            ycbcr_ytab_host(c->h_lut.data(), n, c->q.Lmax, yt.data());
            HIPCHK(c, hipMalloc(&c->d_ytab, lut_floats * sizeof(float)));
            HIPCHK(c, hipMemcpy(c->d_ytab, yt.data(), lut_floats * sizeof(float), hipMemcpyHostToDevice));
            q.ytab = c->d_ytab;
        }
    }
    q.lut_len = (int)n;
    q.pad = (int)(lut_floats - n);
    q.maxVal = (int)n - 1;                                   // (int)pow(2,bitdepth)-1, src/luma_quantizer.cpp:180
    q.mode = n <= 4096 ? LUT_LITERAL_LDS : LUT_LITERAL_GLOBAL;
    q.shift = q.kmin = q.nbuckets = 0;
    c->tix.reset();
    c->index_ready = false;
    return LUMAHIP_OK;
}

// The encode-side search index, built (or fetched from the cache) when the first encode-side launch needs it: a context
// that only decodes never pays for it (for a 16-bit table it is tens of millions of host table probes).
// Every monotone finite table gets threshold records (lut_index.hpp): in LDS when they fit lds_table_max, else in global
// memory (L2-resident).  Anything else (NaNs, decreasing entries -- a decoder may be handed any attachment-434 table) runs
// the reference's bisection literally; so does everything after lumahip_tune(ctx, "force_literal", 1) (the tests' hook).
int lhost::ensure_search_index(lumahip_ctx *c)
{
    if (c->index_ready)
        return LUMAHIP_OK;
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set (call lumahip_set_quantizer first)");
    HIPCHK(c, hipSetDevice(c->device));
    QuantDev &q = c->q;
    if (!c->force_literal) {
        c->tix = cached_thresh_index(c->h_lut);
        if (c->tix->ok) {
            const size_t powf_b = (q.cs == CS_YCBCR) ? sizeof(PowfTablesWide) : 0;
            const ThreshIndex &ix = *c->tix;
            std::vector<uint32_t> r((ix.rec.size() + 3) & ~(size_t)3, 0u);
            memcpy(r.data(), ix.rec.data(), ix.rec.size() * sizeof(uint32_t));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_rec);
            c->d_rec = nullptr;
            HIPCHK(c, hipMalloc(&c->d_rec, r.size() * sizeof(uint32_t)));
            HIPCHK(c, hipMemcpy(c->d_rec, r.data(), r.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            q.rec = c->d_rec;
            q.mode = (ix.rec.size() * 4 <= c->lds_table_max && ix.rec.size() * 4 + 16 + powf_b <= LUMAHIP_LDS_PER_WORKGROUP)
                         ? LUT_THRESH_LDS
                         : LUT_THRESH_GLOBAL;
            q.shift = ix.shift;
            q.kmin = ix.kmin;
            q.nbuckets = ix.nbuckets;
            q.kscale = 0.0f;
            if (q.mode == LUT_THRESH_GLOBAL && c->use_lin_index) {
                // float-bit records too large for LDS: an evenly spaced table (PTF_LINEAR) fits when its records are keyed by
                // VALUE instead (lut_index.hpp LinIndex: LINEAR-12 32 KiB against 229 KiB)
                c->lix = cached_lin_index(c->h_lut);
                const LinIndex &lx = *c->lix;
                if (lx.ok && lx.rec.size() * 4 <= c->lds_table_max && lx.rec.size() * 4 + 16 + powf_b <= LUMAHIP_LDS_PER_WORKGROUP) {
                    std::vector<uint32_t> rl((lx.rec.size() + 3) & ~(size_t)3, 0u);
                    memcpy(rl.data(), lx.rec.data(), lx.rec.size() * sizeof(uint32_t));
                    (void)hipFree(c->d_rec);
                    c->d_rec = nullptr;
                    HIPCHK(c, hipMalloc(&c->d_rec, rl.size() * sizeof(uint32_t)));
                    HIPCHK(c, hipMemcpy(c->d_rec, rl.data(), rl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                    q.rec = c->d_rec;
                    q.mode = LUT_LINKEY_LDS;
                    q.shift = 0;
                    q.kmin = 0;
                    q.nbuckets = lx.nbuckets;
                    q.kscale = lx.kscale;
                }
            }
            // YCbCr encode: the luminance code straight from the luma y (composite records; luma_device.hpp ycbcr_fwd<., YCODE>)
            if (q.cs == CS_YCBCR && c->use_ycbcr_tables && q.mode == LUT_THRESH_LDS) {
                c->tix_y = cached_ycbcr_index(c->h_lut, q.Lmax);
                const ThreshIndex &iy = *c->tix_y;
                if (iy.ok && iy.rec.size() * 4 <= c->lds_table_max && iy.rec.size() * 4 + 16 + powf_b <= LUMAHIP_LDS_PER_WORKGROUP) {
                    std::vector<uint32_t> ry((iy.rec.size() + 3) & ~(size_t)3, 0u);
                    memcpy(ry.data(), iy.rec.data(), iy.rec.size() * sizeof(uint32_t));
                    (void)hipFree(c->d_rec_y);
                    c->d_rec_y = nullptr;
                    HIPCHK(c, hipMalloc(&c->d_rec_y, ry.size() * sizeof(uint32_t)));
                    HIPCHK(c, hipMemcpy(c->d_rec_y, ry.data(), ry.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                    c->q_y = q;
                    c->q_y.rec = c->d_rec_y;
                    c->q_y.shift = iy.shift;
                    c->q_y.kmin = iy.kmin;
                    c->q_y.nbuckets = iy.nbuckets;
                }
            }
        }
    }
    c->index_ready = true;
    return LUMAHIP_OK;
}

// after a tuning change that moves a table between LDS and global memory / switches the search
static int requantize(lumahip_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return upload_table(c);
}

extern "C" int lumahip_set_quantizer(lumahip_ctx *c, int ptf, unsigned bitdepth, int cs, unsigned bitdepthC,
                                     float maxLum, float minLum, const float *lut, size_t n)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (bitdepth < 1 || bitdepth > 16 || bitdepthC < 1 || bitdepthC > 16)
        return fail(c, LUMAHIP_ERR_ARG, "bit depths must be 1..16 (got %u / %u)", bitdepth, bitdepthC);
    if (!lut || n != ((size_t)1 << bitdepth))
        return fail(c, LUMAHIP_ERR_ARG, "LUT must hold 2^bitdepth = %zu floats (got %zu)", (size_t)1 << bitdepth, n);
    if (ptf < 0 || ptf > 4)
        return fail(c, LUMAHIP_ERR_ARG, "unknown transfer function %d", ptf);
    if (c->es_head != c->es_tail)
        return fail(c, LUMAHIP_ERR_STATE, "frames pushed with lumahip_encode_stream_push are still in flight: pop them before changing the quantizer");
    // an unknown colour space is accepted here, as in the reference (setQuantizer stores it blindly,
    // src/luma_quantizer.cpp:181); the transform entry points then fail the way transformColorSpace does.
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    lag_policy_reset(c->half_pol);   // a new stream: the kernel-choice policies start afresh
    lag_policy_reset(c->rb_pol);
    c->have_quant = false;
    c->h_lut.assign(lut, lut + n);
    c->q.maxC = (float)(((unsigned)1 << bitdepthC) - 1);     // src/luma_quantizer.cpp:183
    c->q.cs = cs;
    c->q.Lmax = maxLum;
    c->ptf = ptf;
    c->bitdepth = bitdepth;
    c->bitdepthC = bitdepthC;
    c->minLum = minLum;
    const int rc = upload_table(c);
    if (rc)
        return rc;
    c->have_quant = true;
    return LUMAHIP_OK;
}

// host-only view of the threshold records (no GPU, no context): info = {ok, mant_bits, shift, kmin, nbuckets}
extern "C" int lumahip_thresh_index_host(const float *lut, size_t n, int info[5], uint32_t *rec_out, size_t rec_cap)
{
    if (!lut || !info || n < 2 || n > 65536)
        return LUMAHIP_ERR_ARG;
    const ThreshIndex ix = build_thresh_index(lut, (int)n, 1 << 19);
    info[0] = ix.ok ? 1 : 0;
    info[1] = ix.mant_bits;
    info[2] = ix.shift;
    info[3] = ix.kmin;
    info[4] = ix.nbuckets;
    if (rec_out && ix.ok) {
        if (rec_cap < ix.rec.size())
            return LUMAHIP_ERR_ARG;
        memcpy(rec_out, ix.rec.data(), ix.rec.size() * sizeof(uint32_t));
    }
    return LUMAHIP_OK;
}

// host-only view of the value-keyed records (no GPU, no context): info = {ok, nbuckets, bits of kscale}; rec_out: 2 words per bucket
extern "C" int lumahip_lin_index_host(const float *lut, size_t n, int info[3], uint32_t *rec_out, size_t rec_cap)
{
    if (!lut || !info || n < 2 || n > 65536)
        return LUMAHIP_ERR_ARG;
    const LinIndex ix = build_lin_index(lut, (int)n, 1 << 15);
    info[0] = ix.ok ? 1 : 0;
    info[1] = ix.nbuckets;
    memcpy(&info[2], &ix.kscale, sizeof(float));
    if (rec_out && ix.ok) {
        if (rec_cap < ix.rec.size())
            return LUMAHIP_ERR_ARG;
        memcpy(rec_out, ix.rec.data(), ix.rec.size() * sizeof(uint32_t));
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_quantizer_info(const lumahip_ctx *c, int info[5])
{
    if (!c || !info)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return LUMAHIP_ERR_STATE;
    lumahip_ctx *m = const_cast<lumahip_ctx *>(c);  // builds the (lazily built) search index if nothing has yet
    const int rc = ensure_search_index(m);
    if (rc)
        return rc;
    const bool ok = c->tix && c->tix->ok;
    info[0] = c->q.mode;
    info[1] = (ok && c->q.mode != LUT_LINKEY_LDS) ? c->tix->mant_bits : 0;
    info[2] = c->q.nbuckets;
    info[3] = (ok && c->q.mode != LUT_LINKEY_LDS) ? c->tix->shift : 0;
    info[4] = (int)lds_bytes(c, true, c->q.cs);
    return LUMAHIP_OK;
}

namespace lhost {

// host-only views of the two per-stream YCbCr tables (no GPU, no context), for the CPU tests
extern "C" int lumahip_ycbcr_luma_index_host(const float *lut, size_t n, float Lmax, int info[5], uint32_t *rec_out, size_t rec_cap)
{
    if (!lut || !info || n < 2 || n > 65536)
        return LUMAHIP_ERR_ARG;
    const int maxVal = (int)n - 1;
    const ThreshIndex ix = build_thresh_index_fn([&](float t) { return ycbcr_luma_code_host(t, lut, maxVal, Lmax); }, maxVal, 1 << 16, true);
    info[0] = ix.ok ? 1 : 0;
    info[1] = ix.mant_bits;
    info[2] = ix.shift;
    info[3] = ix.kmin;
    info[4] = ix.nbuckets;
    if (rec_out && ix.ok) {
        if (rec_cap < ix.rec.size())
            return LUMAHIP_ERR_ARG;
        memcpy(rec_out, ix.rec.data(), ix.rec.size() * sizeof(uint32_t));
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_ycbcr_ytab_host(const float *lut, size_t n, float Lmax, float *out)
{
    if (!lut || !out || n < 1)
        return LUMAHIP_ERR_ARG;
    ycbcr_ytab_host(lut, n, Lmax, out);
    return LUMAHIP_OK;
}

// The half-input table of the YCbCr encode kernels for this call's preScaling: a pure function of (sc, Lmax), 124 KiB, built
// with the host libm in about a millisecond and kept -- per process on the host, per context on the device -- because a stream
// encodes every frame with the same pair.  A context holds up to four device copies; launches that read an older copy may
// still be queued on any stream or lane when a fifth pair turns up, so making room waits for the device first.
namespace {
struct HalfHostEntry {
    float sc, Lmax;
    bool ok;
    std::shared_ptr<const std::vector<float>> tab;
};
std::mutex g_half_mutex;
std::vector<HalfHostEntry> g_half_cache;

HalfHostEntry cached_half_table(float sc, float Lmax)
{
    std::lock_guard<std::mutex> lk(g_half_mutex);
    for (auto &e : g_half_cache)
        if (memcmp(&e.sc, &sc, 4) == 0 && memcmp(&e.Lmax, &Lmax, 4) == 0)
            return e;
    HalfHostEntry e;
    e.sc = sc;
    e.Lmax = Lmax;
    auto t = std::make_shared<std::vector<float>>((size_t)lds_half_bytes() / sizeof(float), 0.0f);
    e.ok = ycbcr_half_table_host(sc, Lmax, t->data());
    if (e.ok)
        e.tab = t;
    if (g_half_cache.size() >= 8)
        g_half_cache.erase(g_half_cache.begin());
    g_half_cache.push_back(e);
    return e;
}
}  // namespace

int half_table_for(lumahip_ctx *c, float sc, const float **tab)
{
    *tab = nullptr;
    const float Lmax = c->q.Lmax;
    for (auto &t : c->half_tabs)
        if (memcmp(&t.sc, &sc, 4) == 0 && memcmp(&t.Lmax, &Lmax, 4) == 0) {
            t.last_use = ++c->half_clock;
            *tab = t.d;
            return LUMAHIP_OK;
        }
    const HalfHostEntry h = cached_half_table(sc, Lmax);
    if (c->half_tabs.size() >= 4) {
        HIPCHK(c, hipDeviceSynchronize());
        size_t old = 0;
        for (size_t i = 1; i < c->half_tabs.size(); i++)
            if (c->half_tabs[i].last_use < c->half_tabs[old].last_use)
                old = i;
        (void)hipFree(c->half_tabs[old].d);
        c->half_tabs.erase(c->half_tabs.begin() + (long)old);
    }
    lumahip_ctx::HalfTab t;
    t.sc = sc;
    t.Lmax = Lmax;
    t.last_use = ++c->half_clock;
    if (h.ok) {
        HIPCHK(c, hipMalloc(&t.d, (size_t)lds_half_bytes()));
        // a blocking copy into memory nothing else knows yet: done when it returns, whatever stream the launch goes to
        if (hipMemcpy(t.d, h.tab->data(), (size_t)lds_half_bytes(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(t.d);
            return fail(c, LUMAHIP_ERR_HIP, "upload of the half-input table failed");
        }
    }
    c->half_tabs.push_back(t);
    *tab = t.d;
    return LUMAHIP_OK;
}

extern "C" int lumahip_ycbcr_half_table_host(float sc, float Lmax, float *out, size_t cap)
{
    if (!out || cap < (size_t)HALF_TABLE_LEN)
        return LUMAHIP_ERR_ARG;
    return ycbcr_half_table_host(sc, Lmax, out) ? LUMAHIP_OK : LUMAHIP_ERR_UNSUPPORTED;
}

extern "C" int lumahip_half_table_info(lumahip_ctx *c, float sc, int info[6])
{
    if (!c || !info)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_search_index(c))
        return rc;
    info[0] = info[1] = 0;
    info[3] = HALF_TABLE_LEN;
    if (ycbcr_composite_ready(c) && c->half_mode != 0 && lds_bytes(c, true, CS_YCBCR, true, true) <= LUMAHIP_LDS_PER_WORKGROUP) {
        const float *t = nullptr;
        if (int rc = half_table_for(c, sc, &t))
            return rc;
        info[0] = t != nullptr;
        info[1] = t ? (int)lds_bytes(c, true, CS_YCBCR, true, true) : 0;
    }
    info[2] = 0;
    for (const auto &t : c->half_tabs)
        info[2] += t.d != nullptr;
    info[4] = (int)std::min<unsigned long>(c->half_launches, 0x7fffffffUL);
    info[5] = (int)std::min<unsigned long>(c->half_pol.backoff_launches, 0x7fffffffUL);
    return LUMAHIP_OK;
}

// The NUMA node of the context's GPU and the CPUs of that node this process may run on.  Nothing on a one-node host.
void numa_resolve(lumahip_ctx *c)
{
    if (c->numa_resolved)
        return;
    c->numa_resolved = true;
    c->numa_node = -1;
    c->numa_cpus.clear();
    if (!c->numa_mode)
        return;
    // more than one memory node online?  (node numbers can be sparse -- "0,2" -- so the list is read, not node1 probed)
    {
        char buf[256] = {0};
        FILE *on = fopen("/sys/devices/system/node/online", "r");
        if (!on)
            return;
        const bool got = fgets(buf, sizeof buf, on) != nullptr;
        fclose(on);
        if (!got || (!strchr(buf, ',') && !strchr(buf, '-')))
            return;   // "0": one node
    }
    int node = c->numa_force_node;
    if (node < 0) {
        int v = -1;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeHostNumaId, c->device) == hipSuccess && v >= 0)
            node = v;
        else
            (void)hipGetLastError();
    }
    if (node < 0) {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) == hipSuccess)
            node = numa_node_of_pci("/sys", bus);
        else
            (void)hipGetLastError();
    }
    if (node < 0)
        return;
    std::vector<int> cpus;
    if (!numa_cpus_of_node("/sys", node, numa_allowed_cpus(), cpus))
        return;   // (e.g. a cpuset that excludes the whole node: leave the threads where the scheduler puts them)
    c->numa_node = node;
    c->numa_cpus = cpus;
}

extern "C" int lumahip_half_upload_info(const lumahip_ctx *c, long info[3])
{
    if (!c || !info)
        return LUMAHIP_ERR_ARG;
    info[0] = (long)c->in16_frames;
    info[1] = (long)c->in16_fallbacks;
    info[2] = c->in16_backoff;
    return LUMAHIP_OK;
}

extern "C" int lumahip_numa_info(lumahip_ctx *c, int info[3])
{
    if (!c || !info)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    numa_resolve(c);
    info[0] = c->numa_node;
    info[1] = (int)c->numa_cpus.size();
    info[2] = c->numa_cpus.empty() ? -1 : c->numa_cpus[0];
    return LUMAHIP_OK;
}

extern "C" int lumahip_numa_pin_current_thread(lumahip_ctx *c)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    numa_resolve(c);
    if (c->numa_node >= 0 && c->numa_mode != 2 && !numa_pin_thread(pthread_self(), c->numa_cpus))
        return fail(c, LUMAHIP_ERR_STATE, "pthread_setaffinity_np refused the CPUs of NUMA node %d", c->numa_node);
    return LUMAHIP_OK;
}

// Kernel choice from feedback (lumahip_internal.hpp LagPolicy says what and why).
void lag_policy_reset(LagPolicy &p)
{
    // words of launches still in flight belong to the stream that ends here: wait for them, clear them
    for (const auto &q : p.pending) {
        (void)hipEventSynchronize(p.ev[q.slot]);
        __atomic_store_n(&p.h_flag[q.slot], 0u, __ATOMIC_RELAXED);
    }
    p.pending.clear();
    p.state = LagPolicy::ON_FAST;
    p.backoff = p.backoff_len = 0;
}

void lag_policy_destroy(LagPolicy &p)
{
    if (!p.h_flag)
        return;
    for (auto &e : p.ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipHostFree(p.h_flag);
    p.h_flag = nullptr;
}

static void lag_policy_word(LagPolicy &p, const LagPolicy::Pending &q, bool bad)
{
    if (p.state == LagPolicy::ON_FAST) {
        if (bad) {
            p.bad_words++;
            p.backoff_len = 16;
            p.backoff = p.backoff_len;
            p.state = LagPolicy::BACKOFF;
        }
    } else if (p.state == LagPolicy::PROBE_WAIT && q.probe) {
        if (bad) {
            p.bad_words++;
            p.backoff_len = std::min(2 * std::max(p.backoff_len, 8), p.max_backoff);
            p.backoff = p.backoff_len;
            p.state = LagPolicy::BACKOFF;
        } else {
            p.backoff_len = 0;
            p.state = LagPolicy::ON_FAST;
        }
    }
    // (BACKOFF, or a stale non-probe launch: the launches that were in flight when the first bad word arrived say nothing new)
}

bool lag_policy_next(LagPolicy &p, uint32_t **flag)
{
    *flag = nullptr;
    if (!p.h_flag) {
        if (hipHostMalloc(reinterpret_cast<void **>(&p.h_flag), LagPolicy::RING * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            p.h_flag = nullptr;
            (void)hipGetLastError();
            return true;   // no feedback channel: always the fast kernel
        }
        for (int i = 0; i < LagPolicy::RING; i++) {
            p.h_flag[i] = 0;
            if (hipEventCreateWithFlags(&p.ev[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                for (int j = 0; j < i; j++) {
                    (void)hipEventDestroy(p.ev[j]);
                    p.ev[j] = nullptr;
                }
                (void)hipHostFree(p.h_flag);
                p.h_flag = nullptr;
                return true;
            }
        }
    }
    const unsigned long e = p.elig++;
    // the words that are due: fast launches issued LAG or more eligible launches ago, oldest first
    while (!p.pending.empty() && p.pending.front().issued_at + LagPolicy::LAG <= e) {
        const LagPolicy::Pending q = p.pending.front();
        p.pending.erase(p.pending.begin());
        (void)hipEventSynchronize(p.ev[q.slot]);
        const bool set = __atomic_load_n(&p.h_flag[q.slot], __ATOMIC_RELAXED) != 0;
        __atomic_store_n(&p.h_flag[q.slot], 0u, __ATOMIC_RELAXED);
        lag_policy_word(p, q, set == p.report_is_bad);
    }
    bool probe = false;
    if (p.state == LagPolicy::BACKOFF) {
        if (p.backoff > 0) {
            p.backoff--;
            p.backoff_launches++;
            return false;
        }
        p.state = LagPolicy::PROBE_WAIT;
        probe = true;
    } else if (p.state == LagPolicy::PROBE_WAIT) {
        p.backoff_launches++;
        return false;
    }
    const int slot = (int)(p.seq++ % LagPolicy::RING);   // free: at most LAG - 1 < RING launches are pending here
    p.pending.push_back({e, slot, probe});
    *flag = &p.h_flag[slot];
    return true;
}

// the launch that was handed the newest word never happened (an error return between lag_policy_next and the kernel launch): no
// event will be recorded for it and its word stays 0 -- which for the red / blue policy would read as bad news.  Take the entry
// back; the eligible-launch count keeps the attempt (it only spaces the reads), a probe that did not happen is asked for again.
void lag_policy_cancel(LagPolicy &p)
{
    if (p.pending.empty())
        return;
    const LagPolicy::Pending q = p.pending.back();
    p.pending.pop_back();
    p.seq--;
    if (q.probe && p.state == LagPolicy::PROBE_WAIT) {
        p.state = LagPolicy::BACKOFF;   // backoff == 0: the next eligible launch is the probe
        p.backoff = 0;
    }
}

// right behind a fast launch that was given a feedback word: the event that says its word is final
int lag_policy_launched(lumahip_ctx *c, LagPolicy &p, hipStream_t s)
{
    if (p.pending.empty())
        return LUMAHIP_OK;
    HIPCHK(c, hipEventRecord(p.ev[p.pending.back().slot], s));
    return LUMAHIP_OK;
}

// dynamic LDS of the encode-side kernels (search tables) and of the decode-side kernels (the table itself)
bool ycbcr_composite_ready(const lumahip_ctx *c) { return c->q.cs == CS_YCBCR && c->d_rec_y != nullptr && c->index_ready; }

int check_geom(lumahip_ctx *c, unsigned w, unsigned h, int profile, int cs_eff)
{
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set (call lumahip_set_quantizer first)");
    if (w == 0 || h == 0 || (w & 1) || (h & 1))
        return fail(c, LUMAHIP_ERR_ARG, "Invalid frame size %ux%u (must be even, non-zero)", w, h);
    if (profile < 0 || profile > 3)
        return fail(c, LUMAHIP_ERR_ARG, "profile must be 0..3 (got %d)", profile);
    if (cs_eff < 0 || cs_eff > CS_PACK)
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "Unrecognized color transformation (colour space %d)", cs_eff);
    return LUMAHIP_OK;
}

bool make_geom(FrameGeom &g, unsigned w, unsigned h, int vw, int nw, unsigned nframes)
{
    g.w = (int)w;
    g.h = (int)h;
    g.unitsX = (int)w / vw;
    g.unitsY = (int)h / 2;
    g.tilesX = (g.unitsX + 63) / 64;
    g.tilesY = (g.unitsY + nw - 1) / nw;
    g.tilesPerFrame = g.tilesX * g.tilesY;
    g.nframes = (int)nframes;
    g.interleave = 0;
    const long long total = (long long)g.tilesPerFrame * nframes;
    if (w > 0x7fffffffu / 4 || h > 0x7fffffffu / 4 || total > 0x7fffffffLL)
        return false;  // tile indices are 32-bit
    g.totalTiles = (int)total;
    return true;
}

// rows and bytes per row of plane p as vpx_img_alloc lays it out (src/luma_encoder.cpp:121-128)
hipStream_t launch_stream(lumahip_ctx *c, bool lanes)
{
    if (!lanes || c->lanes_active == 0)
        return c->stream;
    return c->lane_stream[c->lane_next++ % (unsigned)c->lanes_active];
}

void plane_dims(unsigned w, unsigned h, int profile, int p, int &rows, int &row_bytes)
{
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    rows = (p && sub) ? (int)(h + 1) / 2 : (int)h;
    row_bytes = ((p && sub) ? (int)(w + 1) / 2 : (int)w) * bps;
}

// the device entry points take caller-chosen strides: reject layouts in which rows or frames would overlap or the
// kernels would write outside a plane (negative / too small strides, frame strides smaller than a frame)
int check_layout(lumahip_ctx *c, unsigned w, unsigned h, int profile, unsigned nframes, const float *const rgb[3],
                 size_t frame_stride, const int stride[3], const size_t pfs[3])
{
    for (int p = 0; p < 3; p++) {
        int rows, row_bytes;
        plane_dims(w, h, profile, p, rows, row_bytes);
        if (stride[p] < row_bytes)
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: stride %d < row bytes %d", p, stride[p], row_bytes);
        if (nframes > 1 && !c->allow_alias && pfs[p] < (size_t)rows * (size_t)stride[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: frame stride %zu < plane size %zu", p, pfs[p], (size_t)rows * stride[p]);
    }
    if (!rgb || c->allow_alias)
        return LUMAHIP_OK;
    // float frames: colour plane k of frame f covers [rgb[k] + f*frame_stride, + w*h).  No two of the 3*nframes planes may
    // overlap.  Plane sequence k is the arithmetic progression rgb[k] + f*frame_stride; two sequences are fine when they are
    // disjoint as whole ranges (channel-major layouts) or when they interleave with room for each other inside one frame
    // stride (frame-major layouts, the reference's LumaFrame among them).
    const size_t n = (size_t)w * h;
    if (nframes > 1 && frame_stride < n)
        return fail(c, LUMAHIP_ERR_ARG, "frame stride %zu < w*h = %zu floats", frame_stride, n);
    const size_t span = (size_t)(nframes - 1) * frame_stride + n;  // floats one plane sequence covers
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++) {
            const float *lo = rgb[i] < rgb[j] ? rgb[i] : rgb[j], *hi = rgb[i] < rgb[j] ? rgb[j] : rgb[i];
            const size_t d = (size_t)(hi - lo);
            const bool disjoint = d >= span;
            const bool interleaved = d >= n && (nframes == 1 || d + n <= frame_stride);
            if (!disjoint && !interleaved)
                return fail(c, LUMAHIP_ERR_ARG, "colour planes %d and %d (%zu floats apart, frame stride %zu): planes of different "
                                                 "frames would overlap", i, j, d, frame_stride);
        }
    return LUMAHIP_OK;
}

}  // namespace lhost

// ---------------------------------------------------------------------------------------- memory helpers

extern "C" int lumahip_host_register(lumahip_ctx *c, void *p, size_t bytes)
{
    if (!c || !p || !bytes)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostRegister(p, bytes, hipHostRegisterDefault));
    return LUMAHIP_OK;
}

extern "C" int lumahip_host_unregister(lumahip_ctx *c, void *p)
{
    if (!c || !p)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostUnregister(p));
    return LUMAHIP_OK;
}

extern "C" int lumahip_malloc(lumahip_ctx *c, void **p, size_t bytes)
{
    if (!c || !p)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(p, bytes));
    return LUMAHIP_OK;
}

extern "C" int lumahip_free(lumahip_ctx *c, void *p)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipFree(p));
    return LUMAHIP_OK;
}

extern "C" int lumahip_memcpy_h2d(lumahip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = xfer_h2d(c, dst, src, bytes, c->stream);
    if (rc)
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

extern "C" int lumahip_memcpy_d2h(lumahip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = xfer_d2h(c, dst, src, bytes, c->stream);
    if (rc)
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}
