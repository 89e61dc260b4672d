// lumahip_multi.hip -- many GPUs in one process (include/lumahip.h, "many GPUs in one process").
//
// The reference's callers loop `encoder.encode(&frame)` over a sequence (lumaenc.cpp:205-243).  Frames are independent in
// the hot path, so a caller that holds a batch can split it into contiguous blocks, one per shard; each shard is an
// ordinary lumahip_ctx on its GPU driven by its own host thread.  Block (not round-robin) sharding keeps every shard's
// output in stream order for the sequential VP9 consumer downstream (src/luma_encoder.cpp:229-257).  The only shared
// state is the read-only quantizer: built once on the host, uploaded to the first device, and carried to the other
// devices by ONE RCCL broadcast over xGMI (parameter block + table, <= 256 KiB).  No data-path collective.
//
// RCCL is bound at run time (dlopen of librccl.so.1) the first time a quantizer is set: single-GPU users of the library
// never load it, and inside a PyTorch process the copy PyTorch already loaded is the one that is found.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lumahip.h"

namespace {

// the four RCCL entry points used, with the types of rccl/rccl.h (ncclResult_t = int enum, ncclSuccess = 0;
// ncclDataType_t: ncclUint32 = 3)
typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;

    bool tried = false, ok = false;

    // one attempt per process: a library that cannot be opened, or lacks an entry point, stays unusable (and unloaded)
    bool load()
    {
        if (tried)
            return ok;
        tried = true;
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib)
                break;
            why = dlerror();
        }
        if (!lib)
            return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast) {
            why = "librccl.so lacks ncclCommInitAll / ncclBroadcast / ncclGroupStart / ncclGroupEnd";
            CommInitAll = nullptr;
            CommDestroy = nullptr;
            GroupStart = GroupEnd = nullptr;
            Broadcast = nullptr;
            GetErrorString = nullptr;
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        ok = true;
        return true;
    }
};
Rccl g_rccl;
constexpr int NCCL_UINT32 = 3;

constexpr size_t PARAM_WORDS = 16;  // parameter block in front of the table: magic, ptf, bits, cs, bitsC, maxLum, minLum, n

}  // namespace

struct lumahip_multi {
    std::vector<int> dev;                 // device of every shard
    std::vector<lumahip_ctx *> ctx;       // one context per shard
    std::vector<int> udev;                // distinct devices, first = broadcast root
    std::vector<ncclComm_t> comm;         // one communicator rank per distinct device
    std::vector<hipStream_t> bstream;     // broadcast stream per distinct device
    std::vector<uint32_t *> bbuf;         // broadcast buffer per distinct device
    size_t bbuf_words = 0;
    bool used_rccl = false;
    int transport = 0;                    // lumahip_multi_set_transport: 0 auto, 1 RCCL always, 2 host copies always
    std::string note;                     // why the last table travelled the way it did
    std::string err;
};

static int mfail(lumahip_multi *m, int code, const char *fmt, ...)
{
    if (m) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        m->err = buf;
    }
    return code;
}

#define MHIP(m, expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return mfail((m), LUMAHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define MNCCL(m, expr)                                                                                             \
    do {                                                                                                           \
        int e_ = (expr);                                                                                           \
        if (e_ != 0)                                                                                               \
            return mfail((m), LUMAHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                                     \
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "RCCL error", __FILE__, __LINE__);     \
    } while (0)

extern "C" int lumahip_shard_range(unsigned nframes, int shard, int nshards, unsigned *first, unsigned *count)
{
    if (nshards <= 0 || shard < 0 || shard >= nshards || !first || !count)
        return LUMAHIP_ERR_ARG;
    const unsigned base = nframes / (unsigned)nshards, extra = nframes % (unsigned)nshards;
    const unsigned s = (unsigned)shard;
    *first = s * base + (s < extra ? s : extra);
    *count = base + (s < extra ? 1u : 0u);
    return LUMAHIP_OK;
}

extern "C" int lumahip_multi_create(lumahip_multi **out, const int *devices, int nshards)
{
    if (!out)
        return LUMAHIP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return LUMAHIP_ERR_HIP;  // no CPU fallback
    std::vector<int> dv;
    if (!devices) {
        if (nshards <= 0)
            nshards = ndev;
        for (int i = 0; i < nshards; i++)
            dv.push_back(i % ndev);
    } else {
        if (nshards <= 0)
            return LUMAHIP_ERR_ARG;
        dv.assign(devices, devices + nshards);
    }
    for (int d : dv)
        if (d < 0 || d >= ndev)
            return LUMAHIP_ERR_ARG;
    lumahip_multi *m = new lumahip_multi();
    m->dev = dv;
    for (int d : dv) {
        bool seen = false;
        for (int u : m->udev)
            seen = seen || u == d;
        if (!seen)
            m->udev.push_back(d);
        lumahip_ctx *c = nullptr;
        const int rc = lumahip_create(&c, d);
        if (rc != LUMAHIP_OK) {
            lumahip_multi_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
    }
    *out = m;
    return LUMAHIP_OK;
}

extern "C" void lumahip_multi_destroy(lumahip_multi *m)
{
    if (!m)
        return;
    for (lumahip_ctx *c : m->ctx)
        lumahip_destroy(c);
    for (size_t i = 0; i < m->udev.size(); i++) {
        (void)hipSetDevice(m->udev[i]);
        if (i < m->bbuf.size() && m->bbuf[i]) (void)hipFree(m->bbuf[i]);
        if (i < m->bstream.size() && m->bstream[i]) (void)hipStreamDestroy(m->bstream[i]);
    }
    for (ncclComm_t cm : m->comm)
        if (cm && g_rccl.CommDestroy)
            (void)g_rccl.CommDestroy(cm);
    delete m;
}

extern "C" int lumahip_multi_shards(const lumahip_multi *m) { return m ? (int)m->ctx.size() : 0; }

extern "C" lumahip_ctx *lumahip_multi_ctx(lumahip_multi *m, int shard)
{
    return (m && shard >= 0 && shard < (int)m->ctx.size()) ? m->ctx[shard] : nullptr;
}

extern "C" const char *lumahip_multi_last_error(const lumahip_multi *m) { return m ? m->err.c_str() : "null handle"; }

extern "C" int lumahip_multi_used_rccl(const lumahip_multi *m) { return (m && m->used_rccl) ? 1 : 0; }

extern "C" int lumahip_multi_set_transport(lumahip_multi *m, int mode)
{
    if (!m)
        return LUMAHIP_ERR_ARG;
    if (mode < 0 || mode > 2)
        return mfail(m, LUMAHIP_ERR_ARG, "transport must be 0 (auto), 1 (RCCL) or 2 (host copies)");
    m->transport = mode;
    return LUMAHIP_OK;
}

extern "C" const char *lumahip_multi_transport_note(const lumahip_multi *m) { return m ? m->note.c_str() : ""; }

// communicators, broadcast streams and buffers, created once per handle (the table may be replaced many times)
static int ensure_comms(lumahip_multi *m, size_t words)
{
    const int nu = (int)m->udev.size();
    if (m->comm.empty()) {
        if (!g_rccl.load())
            return mfail(m, LUMAHIP_ERR_HIP, "RCCL is needed to broadcast the quantizer and could not be loaded: %s", g_rccl.why.c_str());
        m->comm.assign(nu, nullptr);
        MNCCL(m, g_rccl.CommInitAll(m->comm.data(), nu, m->udev.data()));
        m->bstream.assign(nu, nullptr);
        m->bbuf.assign(nu, nullptr);
        for (int i = 0; i < nu; i++) {
            MHIP(m, hipSetDevice(m->udev[i]));
            MHIP(m, hipStreamCreateWithFlags(&m->bstream[i], hipStreamNonBlocking));
        }
    }
    if (m->bbuf_words < words) {
        for (int i = 0; i < nu; i++) {
            MHIP(m, hipSetDevice(m->udev[i]));
            (void)hipFree(m->bbuf[i]);
            m->bbuf[i] = nullptr;
            MHIP(m, hipMalloc(&m->bbuf[i], words * sizeof(uint32_t)));
            MHIP(m, hipMemset(m->bbuf[i], 0, words * sizeof(uint32_t)));
        }
        m->bbuf_words = words;
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_multi_set_quantizer(lumahip_multi *m, int ptf, unsigned bitdepth, int cs, unsigned bitdepthC,
                                           float maxLum, float minLum, const float *lut, size_t n)
{
    if (!m)
        return LUMAHIP_ERR_ARG;
    if (!lut || bitdepth < 1 || bitdepth > 16 || n != ((size_t)1 << bitdepth))
        return mfail(m, LUMAHIP_ERR_ARG, "LUT must hold 2^bitdepth floats, bitdepth 1..16");
    // How the table reaches the devices.  Several distinct devices: one RCCL broadcast from the first (below).  One distinct
    // device (any number of shards on it), or an RCCL that cannot be loaded: there is nothing to broadcast / nothing to
    // broadcast with -- every shard's context takes the table from the host, and lumahip_multi_used_rccl reports 0.
    // lumahip_multi_set_transport(m, 1) insists on RCCL (a one-rank communicator on a single device; fails without RCCL),
    // (m, 2) never uses it.
    bool direct = m->transport == 2 || (m->transport == 0 && m->udev.size() == 1);
    if (!direct && m->transport == 0 && !g_rccl.load()) {
        direct = true;
        m->note = "RCCL could not be loaded (" + g_rccl.why + "): the table went to every device from the host";
    } else {
        m->note = direct ? "one distinct device (or host copies requested): the table went to every shard from the host"
                         : "RCCL broadcast from the first device";
    }
    if (direct) {
        m->used_rccl = false;
        for (size_t s = 0; s < m->ctx.size(); s++) {
            const int rc = lumahip_set_quantizer(m->ctx[s], ptf, bitdepth, cs, bitdepthC, maxLum, minLum, lut, n);
            if (rc != LUMAHIP_OK)
                return mfail(m, rc, "shard %zu (device %d): %s", s, m->dev[s], lumahip_last_error(m->ctx[s]));
        }
        return LUMAHIP_OK;
    }
    const size_t words = PARAM_WORDS + n;
    int rc = ensure_comms(m, words);
    if (rc)
        return rc;
    // the root device's buffer: parameter block + table, exactly as handed in
    std::vector<uint32_t> blk(words, 0u);
    const uint32_t magic = 0x4c554d41u;  // "LUMA"
    blk[0] = magic;
    blk[1] = (uint32_t)ptf;
    blk[2] = bitdepth;
    blk[3] = (uint32_t)cs;
    blk[4] = bitdepthC;
    memcpy(&blk[5], &maxLum, 4);
    memcpy(&blk[6], &minLum, 4);
    blk[7] = (uint32_t)n;
    memcpy(&blk[PARAM_WORDS], lut, n * sizeof(float));
    MHIP(m, hipSetDevice(m->udev[0]));
    MHIP(m, hipMemcpy(m->bbuf[0], blk.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice));
    // one broadcast, root = the first device; in place on every rank
    const int nu = (int)m->udev.size();
    MNCCL(m, g_rccl.GroupStart());
    for (int i = 0; i < nu; i++) {
        (void)hipSetDevice(m->udev[i]);
        const int e = g_rccl.Broadcast(m->bbuf[i], m->bbuf[i], words, NCCL_UINT32, 0, m->comm[i], m->bstream[i]);
        if (e != 0) {
            (void)g_rccl.GroupEnd();
            return mfail(m, LUMAHIP_ERR_HIP, "ncclBroadcast failed on device %d: %s", m->udev[i],
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error");
        }
    }
    MNCCL(m, g_rccl.GroupEnd());
    for (int i = 0; i < nu; i++) {
        MHIP(m, hipSetDevice(m->udev[i]));
        MHIP(m, hipStreamSynchronize(m->bstream[i]));
    }
    m->used_rccl = true;
    // every shard configures its context from what arrived on ITS device
    std::vector<uint32_t> got(words);
    for (size_t s = 0; s < m->ctx.size(); s++) {
        int ui = 0;
        while (m->udev[ui] != m->dev[s])
            ui++;
        MHIP(m, hipSetDevice(m->dev[s]));
        MHIP(m, hipMemcpy(got.data(), m->bbuf[ui], words * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (got[0] != magic || got[7] != (uint32_t)n)
            return mfail(m, LUMAHIP_ERR_HIP, "shard %zu (device %d): the broadcast quantizer block did not arrive", s, m->dev[s]);
        float ma, mi;
        memcpy(&ma, &got[5], 4);
        memcpy(&mi, &got[6], 4);
        rc = lumahip_set_quantizer(m->ctx[s], (int)got[1], got[2], (int)got[3], got[4], ma, mi,
                                   reinterpret_cast<const float *>(&got[PARAM_WORDS]), got[7]);
        if (rc != LUMAHIP_OK)
            return mfail(m, rc, "shard %zu (device %d): %s", s, m->dev[s], lumahip_last_error(m->ctx[s]));
    }
    return LUMAHIP_OK;
}

// run fn(shard, first, count) on one host thread per shard that has frames; returns the first failure
template <typename F>
static int for_each_shard(lumahip_multi *m, unsigned nframes, F fn)
{
    const int ns = (int)m->ctx.size();
    std::vector<int> rcs(ns, LUMAHIP_OK);
    std::vector<std::thread> th;
    for (int s = 0; s < ns; s++) {
        unsigned first = 0, count = 0;
        (void)lumahip_shard_range(nframes, s, ns, &first, &count);
        if (count == 0)
            continue;
        th.emplace_back([&, s, first, count]() {
            (void)lumahip_numa_pin_current_thread(m->ctx[s]);   // this thread IS the shard's calling thread: next to its GPU (no-op on one-node hosts)
            rcs[s] = fn(s, first, count);
        });
    }
    for (auto &t : th)
        t.join();
    for (int s = 0; s < ns; s++)
        if (rcs[s] != LUMAHIP_OK)
            return mfail(m, rcs[s], "shard %d (device %d): %s", s, m->dev[s], lumahip_last_error(m->ctx[s]));
    return LUMAHIP_OK;
}

extern "C" int lumahip_multi_encode_frames_host(lumahip_multi *m, const float *const *rgb, unsigned nframes, unsigned w,
                                                unsigned h, float sc, int profile, unsigned char *const *planes,
                                                const int stride[3], float *mean_lum)
{
    if (!m || !rgb || !planes || !stride || nframes == 0)
        return mfail(m, LUMAHIP_ERR_ARG, "null argument");
    return for_each_shard(m, nframes, [&](int s, unsigned first, unsigned count) {
        return lumahip_encode_frames_host(m->ctx[s], rgb + first, count, w, h, sc, profile, planes + 3 * (size_t)first, stride,
                                          mean_lum ? mean_lum + first : nullptr);
    });
}

extern "C" int lumahip_multi_decode_frames_host(lumahip_multi *m, const unsigned char *const *planes, const int stride[3],
                                                unsigned nframes, unsigned w, unsigned h, int profile, float sc,
                                                float *const *rgb_out)
{
    if (!m || !rgb_out || !planes || !stride || nframes == 0)
        return mfail(m, LUMAHIP_ERR_ARG, "null argument");
    return for_each_shard(m, nframes, [&](int s, unsigned first, unsigned count) {
        return lumahip_decode_frames_host(m->ctx[s], planes + 3 * (size_t)first, stride, count, w, h, profile, sc, rgb_out + first);
    });
}

extern "C" int lumahip_multi_encode_frames_device(lumahip_multi *m, const float *const *rgb_dev, size_t frame_stride,
                                                  const unsigned *count, unsigned w, unsigned h, float sc, int profile,
                                                  unsigned char *const *planes_dev, const int stride[3], const size_t pfs[3])
{
    if (!m || !rgb_dev || !count || !planes_dev || !stride || !pfs)
        return mfail(m, LUMAHIP_ERR_ARG, "null argument");
    for (size_t s = 0; s < m->ctx.size(); s++) {
        if (count[s] == 0)
            continue;
        const int rc = lumahip_encode_frames_device(m->ctx[s], rgb_dev[s], frame_stride, count[s], w, h, sc, profile,
                                                    planes_dev + 3 * s, stride, pfs, nullptr);
        if (rc != LUMAHIP_OK)
            return mfail(m, rc, "shard %zu (device %d): %s", s, m->dev[s], lumahip_last_error(m->ctx[s]));
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_multi_decode_frames_device(lumahip_multi *m, const unsigned char *const *planes_dev, const int stride[3],
                                                  const size_t pfs[3], const unsigned *count, unsigned w, unsigned h,
                                                  int profile, float sc, float *const *rgb_dev, size_t frame_stride)
{
    if (!m || !rgb_dev || !count || !planes_dev || !stride || !pfs)
        return mfail(m, LUMAHIP_ERR_ARG, "null argument");
    for (size_t s = 0; s < m->ctx.size(); s++) {
        if (count[s] == 0)
            continue;
        const int rc = lumahip_decode_frames_device(m->ctx[s], planes_dev + 3 * s, stride, pfs, count[s], w, h, profile, sc,
                                                    rgb_dev[s], frame_stride);
        if (rc != LUMAHIP_OK)
            return mfail(m, rc, "shard %zu (device %d): %s", s, m->dev[s], lumahip_last_error(m->ctx[s]));
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_multi_sync(lumahip_multi *m)
{
    if (!m)
        return LUMAHIP_ERR_ARG;
    for (size_t s = 0; s < m->ctx.size(); s++) {
        const int rc = lumahip_sync(m->ctx[s]);
        if (rc != LUMAHIP_OK)
            return mfail(m, rc, "shard %zu (device %d): %s", s, m->dev[s], lumahip_last_error(m->ctx[s]));
    }
    return LUMAHIP_OK;
}
