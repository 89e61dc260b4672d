// lumahip_host.hip -- the _host entry points of include/lumahip.h: staging buffers, host <-> device transfers, the 3-slot
// pipeline of the batched forms.  No kernels here.
#include "lumahip_internal.hpp"
#include "half_stage.hpp"

#include <atomic>
#include <condition_variable>
#include <thread>

using namespace lh;
using namespace lhost;

// The fused kernels on packed frames for the entry points of this file: always on c->stream (which the callers point at their
// kernel stream), never on a lane of an open unordered section -- the uploads and downloads around them are ordered against
// that stream's events (include/lumahip.h: only the four _device encode / decode entry points take part in a section).
static int encode_packed(lumahip_ctx *c, const float *rgb, size_t frame_stride, unsigned nframes, unsigned w, unsigned h, float sc,
                         int profile, unsigned char *const planes[3], const int stride[3], const size_t pfs[3], float *stats)
{
    const size_t n = (size_t)w * h;
    const float *const pl[3] = {rgb, rgb + n, rgb + 2 * n};
    return encode_frames_device_impl(c, pl, frame_stride, nframes, w, h, sc, profile, planes, stride, pfs, stats, c->q.cs, false);
}
// the same for a frame that was uploaded as binary16 (xfer_h2d_f16): planes at the same element offsets behind a pointer to halves
static int encode_packed16(lumahip_ctx *c, const void *halves, size_t frame_stride, unsigned nframes, unsigned w, unsigned h, float sc,
                           int profile, unsigned char *const planes[3], const int stride[3], const size_t pfs[3], float *stats)
{
    const size_t n = (size_t)w * h;
    const uint16_t *b = static_cast<const uint16_t *>(halves);
    const float *const pl[3] = {reinterpret_cast<const float *>(b), reinterpret_cast<const float *>(b + n), reinterpret_cast<const float *>(b + 2 * n)};
    return encode_frames_device_impl(c, pl, frame_stride, nframes, w, h, sc, profile, planes, stride, pfs, stats, c->q.cs, false, true);
}
static int decode_packed(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3], const size_t pfs[3],
                         unsigned nframes, unsigned w, unsigned h, int profile, float sc, float *rgb, size_t frame_stride)
{
    const size_t n = (size_t)w * h;
    float *const pl[3] = {rgb, rgb + n, rgb + 2 * n};
    return decode_impl(c, planes, stride, pfs, nframes, w, h, profile, sc, pl, frame_stride, DisplayParams(), c->q.cs, false);
}

// ---- host <-> device transfers of the _host entry points --------------------------------------------------------
// Caller memory is pageable unless the caller pinned it (hipHostMalloc, hipHostRegister / lumahip_host_register).
// Pinned memory is handed to the copy engine directly (asynchronous, the fast path of the batched entry points).
// Pageable memory is NOT handed to hipMemcpy*Async: the runtime then pins the caller's pages on the fly and caches
// that pinning, and on this stack (ROCm 7.2, MI355X) the GPU occasionally faulted on such a range when host buffers are
// allocated and freed at a high rate ("Memory access fault by GPU ... on address <host heap page>", about one run of
// the GPU test suite in twenty).  Pageable data therefore moves through a ring of context-owned pinned chunks per direction:
// the CPU fills (empties) one chunk while the DMAs of the previous ones are in flight.
static constexpr size_t XFER_CHUNK = (size_t)8 << 20;

// ---- copy threads ---------------------------------------------------------------------------------------------------------
// One CPU thread copies pageable memory into a pinned chunk at ~21 GB/s on the GPU box's host, a third of what the PCIe
// link moves (profiles/r02_hostfed.txt: 1.4 Gpixel/s pageable against 4.2 pinned).  The staging copies are therefore split
// over a few persistent worker threads owned by the context (lumahip_tune "copy_threads", default 5; 0 = the calling thread
// alone): the CPU side then keeps up with the DMA of the previous chunk.
struct lumahip_copy_pool {
    struct Job {
        unsigned char *dst;
        const unsigned char *src;
        size_t width, rows, dst_pitch, src_pitch;  // rows x width bytes; rows == 1: one flat span
        bool to_half = false;                      // flat span of `width` bytes of floats -> width / 2 bytes of halves (half_stage.cpp)
    };
    std::atomic<bool> inexact{false};   // a to_half job met a value that is not a half (raised by any worker, read by copy_to_half)
    // A worker that has just finished a piece polls for the next one for ~0.5 ms before it goes to sleep on the condition
    // variable: while a frame streams through, the pieces follow each other within tens of microseconds, and waking a
    // sleeping thread costs 50-100 us on the GPU box's host -- as much as copying the piece (a 4K band is 8 MB).
    int spin = 2000;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go;
    std::vector<Job> jobs;     // one per worker for the current generation
    std::atomic<unsigned> generation{0};
    std::atomic<int> pending{0};
    std::atomic<bool> stop{false};

    // cpus: the CPUs of the GPU's NUMA node the workers are pinned to (empty, the default: wherever the scheduler puts them;
    // lumahip_tune "numa" 1) -- they write the pinned staging chunks, which live on that node, and read them back out of it
    lumahip_copy_pool(int n, int spin_, const std::vector<int> &cpus) : spin(spin_)   // (spin is set before the workers exist: they read it without synchronisation)
    {
        jobs.resize(n);
        for (int i = 0; i < n; i++) {
            workers.emplace_back([this, i]() { loop(i); });
            if (!cpus.empty())
                (void)lh::numa_pin_thread(workers.back().native_handle(), cpus);
        }
    }
    ~lumahip_copy_pool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop.store(true);
        }
        cv_go.notify_all();
        for (auto &t : workers)
            t.join();
    }
    void run_job(const Job &j)
    {
        if (j.to_half) {
            if (!lh::convert_f32_to_f16_checked(reinterpret_cast<const float *>(j.src), reinterpret_cast<uint16_t *>(j.dst), j.width / 4))
                inexact.store(true, std::memory_order_relaxed);
            return;
        }
        run(j);
    }
    static void run(const Job &j)
    {
        if (j.rows == 1) {
            memcpy(j.dst, j.src, j.width);
        } else {
            for (size_t r = 0; r < j.rows; r++)
                memcpy(j.dst + r * j.dst_pitch, j.src + r * j.src_pitch, j.width);
        }
    }
    void loop(int me)
    {
        unsigned seen = 0;
        for (;;) {
            int spins = 0;
            unsigned g;
            while ((g = generation.load(std::memory_order_acquire)) == seen) {
                if (stop.load(std::memory_order_relaxed))
                    return;
                if (++spins < spin) {
                    __builtin_ia32_pause();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return stop.load() || generation.load() != seen; });
            }
            seen = g;
            const Job j = jobs[me];
            if (j.width)
                run_job(j);
            pending.fetch_sub(1, std::memory_order_release);
        }
    }
    // rows x width bytes from src (pitch sp) to dst (pitch dp), split over the workers and the calling thread
    void copy(unsigned char *dst, size_t dp, const unsigned char *src, size_t sp, size_t width, size_t rows)
    {
        const size_t parts = workers.size() + 1;
        const bool flat = rows == 1;
        const size_t total = flat ? width : rows;
        if (total * (flat ? 1 : width) < ((size_t)256 << 10) || total < parts) {  // small: not worth handing out
            run(Job{dst, src, width, rows, dp, sp});
            return;
        }
        // flat spans are cut at 4 KiB boundaries so that no two threads share a page
        size_t per = (total + parts - 1) / parts;
        if (flat)
            per = (per + 4095) & ~(size_t)4095;
        auto part = [&](size_t k) -> Job {
            const size_t a = std::min(total, k * per), b = std::min(total, (k + 1) * per);
            if (a >= b)
                return Job{nullptr, nullptr, 0, 0, 0, 0};
            return flat ? Job{dst + a, src + a, b - a, 1, 0, 0} : Job{dst + a * dp, src + a * sp, width, b - a, dp, sp};
        };
        for (size_t k = 0; k < workers.size(); k++)
            jobs[k] = part(k + 1);
        pending.store((int)workers.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);   // (a worker about to sleep re-checks the generation under this lock)
            generation.fetch_add(1, std::memory_order_release);
        }
        cv_go.notify_all();
        const Job mine = part(0);
        if (mine.width)
            run(mine);
        while (pending.load(std::memory_order_acquire) != 0)
            __builtin_ia32_pause();
    }
    // n floats at src -> n halves at dst, split over the workers and the calling thread (spans cut at multiples of 2048 floats);
    // false when some value is not a half (dst is then useless)
    bool copy_to_half(uint16_t *dst, const float *src, size_t n)
    {
        inexact.store(false, std::memory_order_relaxed);
        const size_t parts = workers.size() + 1;
        if (n < ((size_t)64 << 10) || n < parts)
            return lh::convert_f32_to_f16_checked(src, dst, n);
        size_t per = ((n + parts - 1) / parts + 2047) & ~(size_t)2047;
        auto part = [&](size_t k) -> Job {
            const size_t a = std::min(n, k * per), b = std::min(n, (k + 1) * per);
            Job j{nullptr, nullptr, 0, 0, 0, 0};
            if (a < b) {
                j.dst = reinterpret_cast<unsigned char *>(dst + a);
                j.src = reinterpret_cast<const unsigned char *>(src + a);
                j.width = (b - a) * 4;
                j.rows = 1;
                j.to_half = true;
            }
            return j;
        };
        for (size_t k = 0; k < workers.size(); k++)
            jobs[k] = part(k + 1);
        pending.store((int)workers.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            generation.fetch_add(1, std::memory_order_release);
        }
        cv_go.notify_all();
        const Job mine = part(0);
        if (mine.width)
            run_job(mine);
        while (pending.load(std::memory_order_acquire) != 0)
            __builtin_ia32_pause();
        return !inexact.load(std::memory_order_relaxed);
    }
};

void lumahip_copy_pool_destroy(lumahip_copy_pool *p) { delete p; }

static void staged_copy(lumahip_ctx *c, unsigned char *dst, size_t dp, const unsigned char *src, size_t sp, size_t width, size_t rows)
{
    if (c->copy_threads > 0 && !c->copy_pool) {
        numa_resolve(c);
        c->copy_pool = new lumahip_copy_pool(c->copy_threads, c->copy_spin, c->numa_mode == 2 ? std::vector<int>() : c->numa_cpus);
    }
    if (c->copy_pool)
        c->copy_pool->copy(dst, dp, src, sp, width, rows);
    else
        lumahip_copy_pool::run(lumahip_copy_pool::Job{dst, src, width, rows, dp, sp});
}

static bool host_range_is_pinned(const void *p, size_t bytes)
{
    if (!p || !bytes)
        return false;
    const unsigned char *ends[2] = {(const unsigned char *)p, (const unsigned char *)p + bytes - 1};
    for (const unsigned char *q : ends) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();  // plain malloc memory: not an error of ours
            return false;
        }
        if (at.type != hipMemoryTypeHost)
            return false;
    }
    return true;
}

static int stage_alloc(lumahip_ctx *c, lumahip_ctx::Stage &st, size_t bytes = XFER_CHUNK)
{
    if (!st.h) {
        // on the GPU's NUMA node: the calling thread's memory policy says where, hipHostMallocNumaUser makes the runtime follow it
        numa_resolve(c);
        bool placed = false;
        if (c->numa_node >= 0 && c->numa_mode != 3 && numa_prefer_node(c->numa_node)) {
            placed = hipHostMalloc((void **)&st.h, bytes, hipHostMallocNumaUser) == hipSuccess;
            if (!numa_prefer_node(-1)) {
                // the caller's memory policy could not be put back: its later allocations would silently prefer the GPU's node.
                // Say so, and stop placing rings for this context (the rings already made stay where they are)
                c->numa_mode = 0;
                c->numa_node = -1;
                (void)fail(c, LUMAHIP_ERR_STATE, "set_mempolicy could not restore the calling thread's memory policy after placing a "
                                                 "staging ring; NUMA placement is off for this context from here on");
                if (placed)
                    (void)hipHostFree(st.h);
                st.h = nullptr;
                return LUMAHIP_ERR_STATE;
            }
            if (!placed) {
                st.h = nullptr;
                (void)hipGetLastError();
            }
        }
        if (!placed)
            HIPCHK(c, hipHostMalloc((void **)&st.h, bytes, hipHostMallocDefault));
        if (!st.ev)
            HIPCHK(c, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    }
    return LUMAHIP_OK;
}

// an upload chunk is free again once the DMA that read it has completed
static int stage_ready(lumahip_ctx *c, lumahip_ctx::Stage &st)
{
    if (int rc = stage_alloc(c, st))
        return rc;
    if (st.pending) {
        HIPCHK(c, hipEventSynchronize(st.ev));
        st.pending = false;
    }
    return LUMAHIP_OK;
}

// A download chunk is free again once its DMA has landed AND its bytes have been copied out to the caller's pageable
// memory; the copy happens here, i.e. lazily, when the ring comes round to the chunk again or when a call drains what is
// still in flight (d2h_flush).  The calling thread therefore never waits for a download it has only just queued: it goes
// on staging the next upload, which is what keeps the copy engine busy in the batched entry points (pageable frames:
// 3.1 -> 4.1 Gpixel/s once the fetch of frame i-1 stopped blocking the staging of frame i+1).  A context-owned thread that
// empties the chunks concurrently was built and measured as well: +2 % on that path, -5 % on the download-heavy decode
// calls (it copies single-threaded where this thread uses the copy threads), so it was not kept (profiles/r03_hostfed_sweep.txt).
static int stage_dn_ready(lumahip_ctx *c, lumahip_ctx::Stage &st)
{
    if (int rc = stage_alloc(c, st, c->dn_chunk))
        return rc;
    if (st.pending) {
        HIPCHK(c, hipEventSynchronize(st.ev));
        st.pending = false;
        staged_copy(c, st.out, st.out_pitch, st.h, st.chunk_pitch, st.width, st.rows);
        st.out = nullptr;
    }
    return LUMAHIP_OK;
}

// every device -> host chunk still in flight: wait for it and copy it out (oldest first)
static int d2h_flush(lumahip_ctx *c)
{
    for (int i = 0; i < lumahip_ctx::N_STAGE_DN; i++) {
        lumahip_ctx::Stage &st = c->stage_dn[(c->dn_next + i) % lumahip_ctx::N_STAGE_DN];
        if (st.h && st.pending)
            if (int rc = stage_dn_ready(c, st))
                return rc;
    }
    return LUMAHIP_OK;
}

// Error paths: download chunks that are still pending point into the CALLER's memory (st.out).  A call that fails must not
// leave them behind -- a later flush, or the ring coming round, would copy into buffers the caller may have freed by then.
// d2h_drop waits for the DMA of every such chunk (all of them, or those with one tag: 0 = the plain calls, sequence number + 1 =
// one pushed frame, so that a failing plain call leaves the chunks of frames pushed earlier alone) and forgets it without copying; DnGuard does that on every exit of a scope that has not been told the downloads were completed or handed over.
static void d2h_drop(lumahip_ctx *c, bool all, unsigned tag)
{
    for (auto &st : c->stage_dn)
        if (st.h && st.pending && (all || st.tag == tag)) {
            if (st.ev)
                (void)hipEventSynchronize(st.ev);
            st.pending = false;
            st.out = nullptr;
        }
}
struct DnGuard {
    lumahip_ctx *c;
    bool all;
    unsigned tag;
    bool armed = true;
    ~DnGuard()
    {
        if (!armed)
            return;
        if (c->s_d2h)
            (void)hipStreamSynchronize(c->s_d2h);
        d2h_drop(c, all, tag);
    }
};

// The pipelined encode paths queue a whole frame's planes for download and go back to staging the next upload; that only
// works while the ring of download chunks holds the frame (five chunks of 8 MiB for a 4K frame's 25 MB, eight are there).
// The planes of an 8K frame are 99.5 MB: thirteen such chunks -- the ring came round to chunks of the SAME frame, the host sat
// waiting for its kernel, and the pipelined paths were slower than the synchronous one (3.35 against 3.55 Gpixel/s).  So the
// download chunks grow with the frame (a fifth of the planes, at most 32 MiB each); they are re-allocated only when nothing is
// in flight.
static int dn_chunks_for(lumahip_ctx *c, size_t planes_bytes)
{
    size_t want = ((planes_bytes / 5 + ((size_t)1 << 20) - 1) >> 20) << 20;
    want = std::min(std::max(want, XFER_CHUNK), (size_t)32 << 20);
    if (want <= c->dn_chunk)
        return LUMAHIP_OK;
    if (int rc = d2h_flush(c))
        return rc;
    for (auto &st : c->stage_dn)
        if (st.h) {
            (void)hipHostFree(st.h);
            st.h = nullptr;   // (the event stays; stage_alloc makes the new buffer on first use)
        }
    c->dn_chunk = want;
    return LUMAHIP_OK;
}

// the chunks of pushed frames up to sequence number `seq` (stream entry points; chunks are in issue order, oldest first)
static int d2h_flush_upto(lumahip_ctx *c, unsigned seq)
{
    for (int i = 0; i < lumahip_ctx::N_STAGE_DN; i++) {
        lumahip_ctx::Stage &st = c->stage_dn[(c->dn_next + i) % lumahip_ctx::N_STAGE_DN];
        if (st.h && st.pending && (int)(st.tag - (seq + 1)) <= 0)   // (tags are sequence number + 1; 0 = a chunk of a plain call: always due)
            if (int rc = stage_dn_ready(c, st))
                return rc;
    }
    return LUMAHIP_OK;
}

// rows x width bytes, host pitch hp, device pitch dp.  Returns once the copies are queued on `s` (the caller's buffer
// is no longer needed if it was pageable: it has been copied into the staging chunks).
static int xfer_h2d_2d(lumahip_ctx *c, void *dst, size_t dp, const void *src, size_t hp, size_t width, size_t rows, hipStream_t s)
{
    if (!width || !rows)
        return LUMAHIP_OK;
    if (host_range_is_pinned(src, (rows - 1) * hp + width)) {
        if (dp == width && hp == width)
            HIPCHK(c, hipMemcpyAsync(dst, src, width * rows, hipMemcpyHostToDevice, s));
        else
            HIPCHK(c, hipMemcpy2DAsync(dst, dp, src, hp, width, rows, hipMemcpyHostToDevice, s));
        return LUMAHIP_OK;
    }
    // staged: the device side is written as whole rows of dp bytes (the padding between rows belongs to the context's
    // own buffers), so that one chunk is one contiguous DMA
    const bool flat = (dp == width && hp == width);
    if (!flat && dp > XFER_CHUNK)
        return fail(c, LUMAHIP_ERR_ARG, "row pitch %zu exceeds the staging chunk", dp);
    const size_t total = flat ? width * rows : rows;                 // bytes or rows
    for (size_t done = 0; done < total;) {
        lumahip_ctx::Stage &st = c->stage_up[c->up_next++ % lumahip_ctx::N_STAGE];
        int rc = stage_ready(c, st);
        if (rc)
            return rc;
        // The first chunks of a call are small (1, 2, 4 MiB, then whole chunks): the copy engine starts after 15 us of
        // staging instead of after the 110 us a whole chunk takes to fill, and nothing of that lead is lost later because the
        // DMA of a chunk (150 us) takes longer than filling the next one.
        size_t cap = XFER_CHUNK;
        if (c->up_ramp < 3)
            cap = std::min(cap, (size_t)1 << (20 + c->up_ramp++));
        const size_t per = flat ? cap : std::max<size_t>(1, cap / dp);   // bytes or rows per chunk
        const size_t n = total - done < per ? total - done : per;
        size_t bytes;
        if (flat) {
            staged_copy(c, st.h, 0, (const unsigned char *)src + done, 0, n, 1);
            bytes = n;
        } else {
            staged_copy(c, st.h, dp, (const unsigned char *)src + done * hp, hp, width, n);
            bytes = (n - 1) * dp + width;
        }
        HIPCHK(c, hipMemcpyAsync((unsigned char *)dst + done * (flat ? 1 : dp), st.h, bytes, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(st.ev, s));
        st.pending = true;
        done += n;
    }
    return LUMAHIP_OK;
}

namespace lhost {

int xfer_h2d(lumahip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    return xfer_h2d_2d(c, dst, bytes, src, bytes, bytes, 1, s);
}

}

// Host floats -> device halves: the half upload.  The reference's LumaFrame is float, but its EXR reader fills it with widened
// binary16 values (src/exr_interface.cpp:77-146); such a frame crosses PCIe in 6 instead of 12 bytes per pixel, and the encode
// kernels instantiated for binary16 input (k_encode<., ., 4, 3, IN16>) widen it back exactly.  The copy threads convert while
// they stage (reading 12 B and writing 6 B per pixel moves less memory than the plain staging copy) and check every value's
// round trip; *exact = false as soon as a chunk holds anything that is not a half -- nothing of that chunk has been queued then,
// and the caller uploads the frame (or band) as floats instead.  Pinned or pageable `src` alike: the CPU reads it either way.
static int xfer_h2d_f16(lumahip_ctx *c, void *dst_halves, const float *src, size_t nfloats, hipStream_t s, bool *exact)
{
    *exact = true;
    if (c->copy_threads > 0 && !c->copy_pool) {
        numa_resolve(c);
        c->copy_pool = new lumahip_copy_pool(c->copy_threads, c->copy_spin, c->numa_mode == 2 ? std::vector<int>() : c->numa_cpus);
    }
    for (size_t done = 0; done < nfloats;) {
        lumahip_ctx::Stage &st = c->stage_up[c->up_next++ % lumahip_ctx::N_STAGE];
        if (int rc = stage_ready(c, st))
            return rc;
        size_t cap = XFER_CHUNK;   // bytes of halves per chunk; the first chunks of a call are small (see xfer_h2d_2d)
        if (c->up_ramp < 3)
            cap = std::min(cap, (size_t)1 << (20 + c->up_ramp++));
        const size_t n = std::min(nfloats - done, cap / 2);
        const bool ok = c->copy_pool ? c->copy_pool->copy_to_half(reinterpret_cast<uint16_t *>(st.h), src + done, n)
                                     : lh::convert_f32_to_f16_checked(src + done, reinterpret_cast<uint16_t *>(st.h), n);
        if (!ok) {
            *exact = false;
            return LUMAHIP_OK;
        }
        HIPCHK(c, hipMemcpyAsync((unsigned char *)dst_halves + done * 2, st.h, n * 2, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(st.ev, s));
        st.pending = true;
        done += n;
    }
    return LUMAHIP_OK;
}

// Whether this call tries the half upload (lumahip_tune "half_upload": 0 never, 1 while the frames hold halves, 2 always try).
// A frame that turns out to hold other values costs the conversion of its first chunks for nothing, so after one such frame
// the next 16 go up as floats before one is tried again, the pause doubling up to 1024 frames while the misses continue.
static bool in16_try(lumahip_ctx *c, unsigned w, bool wants_float_frame)
{
    if (c->in16_mode == 0 || wants_float_frame || !lh::f16c_available() || !encode_supports_in16(c, w))
        return false;
    if (c->in16_mode == 2)
        return true;
    if (c->in16_backoff > 0) {
        c->in16_backoff--;
        return false;
    }
    return true;
}
static void in16_result(lumahip_ctx *c, bool exact)
{
    if (exact) {
        c->in16_frames++;
        c->in16_backoff_len = 0;
    } else {
        c->in16_fallbacks++;
        c->in16_backoff_len = c->in16_backoff_len ? std::min(2 * c->in16_backoff_len, 1024) : 16;
        c->in16_backoff = c->in16_backoff_len;
    }
}

// Device -> host.  Pinned destination: queued on `s`, the caller synchronises.  Pageable destination: the data goes through
// the ring of pinned chunks; with `deferred` false it is in `dst` when the call returns (everything queued on `s` before it has
// completed by then), with `deferred` true the last chunks may still be in flight and d2h_flush() completes them -- which lets
// the DMA of one piece overlap the copy-out of the previous one ACROSS calls (the row bands of the host entry points).
static int xfer_d2h_2d(lumahip_ctx *c, void *dst, size_t hp, const void *src, size_t dp, size_t width, size_t rows, hipStream_t s,
                       bool deferred = false)
{
    if (!width || !rows)
        return LUMAHIP_OK;
    if (host_range_is_pinned(dst, (rows - 1) * hp + width)) {
        if (dp == width && hp == width)
            HIPCHK(c, hipMemcpyAsync(dst, src, width * rows, hipMemcpyDeviceToHost, s));
        else
            HIPCHK(c, hipMemcpy2DAsync(dst, hp, src, dp, width, rows, hipMemcpyDeviceToHost, s));
        return LUMAHIP_OK;
    }
    const bool flat = (dp == width && hp == width);
    if (!flat && dp > c->dn_chunk)
        return fail(c, LUMAHIP_ERR_ARG, "row pitch %zu exceeds the staging chunk", dp);
    const size_t total = flat ? width * rows : rows;
    const size_t per = flat ? c->dn_chunk : c->dn_chunk / dp;
    for (size_t done = 0; done < total;) {
        lumahip_ctx::Stage &st = c->stage_dn[c->dn_next++ % lumahip_ctx::N_STAGE_DN];
        int rc = stage_dn_ready(c, st);   // the chunk this ring slot carried N_STAGE_DN chunks ago has been emptied
        if (rc)
            return rc;
        const size_t n = total - done < per ? total - done : per;
        const size_t bytes = flat ? n : (n - 1) * dp + width;
        HIPCHK(c, hipMemcpyAsync(st.h, (const unsigned char *)src + done * (flat ? 1 : dp), bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(st.ev, s));
        if (flat) {
            st.out = (unsigned char *)dst + done;
            st.out_pitch = st.chunk_pitch = 0;
            st.width = n;
            st.rows = 1;
        } else {
            st.out = (unsigned char *)dst + done * hp;
            st.out_pitch = hp;
            st.chunk_pitch = dp;
            st.width = width;
            st.rows = n;
        }
        st.tag = c->d2h_tag;
        st.pending = true;
        done += n;
    }
    return deferred ? LUMAHIP_OK : d2h_flush(c);
}

namespace lhost {

int xfer_d2h(lumahip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    return xfer_d2h_2d(c, dst, bytes, src, bytes, bytes, 1, s);
}

static int xfer_d2h_deferred(lumahip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    return xfer_d2h_2d(c, dst, bytes, src, bytes, bytes, 1, s, true);
}

// a few floats from the device: through the pinned scratch, synchronous
int read_small(lumahip_ctx *c, float *dst, const float *src_dev, int n, hipStream_t s)
{
    if (!c->h_small)
        HIPCHK(c, hipHostMalloc((void **)&c->h_small, 64 * sizeof(float), hipHostMallocDefault));
    HIPCHK(c, hipMemcpyAsync(c->h_small, src_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    memcpy(dst, c->h_small, (size_t)n * sizeof(float));
    return LUMAHIP_OK;
}

int ensure(lumahip_ctx *c, void **p, size_t *cap, size_t need)
{
    if (*cap >= need)
        return LUMAHIP_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIPCHK(c, hipMalloc(p, need));
    *cap = need;
    return LUMAHIP_OK;
}

}  // namespace lhost

struct PlaneLayout {
    int rows[3];
    int row_bytes[3];
    size_t off[3];
    size_t total;
};

static void plane_layout(PlaneLayout &L, unsigned w, unsigned h, int profile, const int stride[3])
{
    const bool sub = (profile == 0 || profile == 2);
    const int bps = profile > 1 ? 2 : 1;
    size_t off = 0;
    for (int p = 0; p < 3; p++) {
        const int pw = (p && sub) ? (int)(w + 1) / 2 : (int)w;
        const int ph = (p && sub) ? (int)(h + 1) / 2 : (int)h;
        L.rows[p] = ph;
        L.row_bytes[p] = pw * bps;
        L.off[p] = off;
        off += ((size_t)ph * stride[p] + 255) & ~(size_t)255;
    }
    L.total = off;
}

// The reference warns when its (sequentially summed) mean luminance is <= 1.  That fp32 sum is far from the true sum on
// large frames: once the running sum S is large, addends below ulp(S)/2 vanish and the rest are rounded to multiples of
// ulp(S) (measured: -0.2 % at 1080p, several % at 4K on wide-range content), whereas the kernels' statistic (per-wave
// partial sums) is accurate to ~1e-6.  For N <= 2^25 non-negative values (up to 8K frames) S stays below N * 4 around the
// threshold, i.e. ulp(S)/2 <= 2, so the two can only disagree about `<= 1` when the accurate mean lies in [0.25, 4]: inside
// that band the host entry points replace the statistic by the reference's exact value (k_seq_sum), outside it the decision
// is the same either way.  The argument needs both premises, so frames beyond 2^25 pixels and frames whose channel 0 has
// negative values (possible for CS_RGB and the pack-only entry points: cancellation, no bound) always take the exact sum.
static bool mean_needs_reference_sum(float mean, float minimum, unsigned w, unsigned h)
{
    if ((size_t)w * h > ((size_t)1 << 25) || !(minimum >= 0.0f))
        return true;
    return mean >= 0.25f && mean <= 4.0f;
}

static int pipe_streams(lumahip_ctx *c);

// ---- row bands ------------------------------------------------------------------------------------------------------------
// One host frame per call is PCIe time: 99.5 MB up at ~56 GB/s (1.76 ms), 30 us of kernel, 24.9 MB down (0.45 ms); done one
// after the other that is 2.2 ms pinned and more staged (profiles/r03_hostfed_lab.txt).  Rows are independent (pairs of rows
// in 4:2:0), so a large frame is cut into `host_bands` bands of rows: band k+1 goes up while band k is transformed and band
// k-1 comes down -- the link is full duplex -- and the call is bound by the upload plus whatever is left to do for the LAST
// band once its rows have arrived (its kernel, its download, the copy out of the staging chunks).  The bands therefore
// TAPER: each is `band_taper` % of the previous one (default 70: 39.5 / 27.6 / 19.3 / 13.5 % of the rows for four bands), so that
// tail is an eighth of the frame instead of a quarter (rocprofv3 timeline of the uniform split: 0.5 ms of 2.67 ms per
// pageable 4K frame after the last upload chunk; profiles/r03_hostfed_timeline.txt).  A download still fits under the next
// band's upload: it moves a quarter of the bytes.  Bands are multiples of 16 rows (whole tiles of every kernel variant); the
// pixel arithmetic does not depend on the split.
static int band_plan(const lumahip_ctx *c, unsigned w, unsigned h, unsigned r0[lumahip_ctx::MAX_BANDS + 1])
{
    int nb = c->host_bands;
    // small frames are latency, not bandwidth: up to 1920x1080 one piece is as fast or faster (registered frames +10 %, pageable
    // +-0; profiles/r03_hostfed_sweep.txt), from 2560x1440 on the bands win (pageable +12 %)
    if ((size_t)w * h < (size_t)3 << 20 || nb < 2)
        nb = 1;
    const double q = c->band_taper / 100.0;
    double wsum = 0.0, wk = 1.0;
    for (int k = 0; k < nb; k++, wk *= q)
        wsum += wk;
    unsigned row = 0;
    int k = 0;
    wk = 1.0;
    r0[0] = 0;
    for (int i = 0; i < nb && row < h; i++, wk *= q) {
        unsigned rows = (unsigned)(h * (wk / wsum) + 0.5);
        rows = (rows + 15) & ~15u;
        if (rows < 64)
            rows = 64;
        if (i == nb - 1 || row + rows > h || h - (row + rows) < 64)   // the last band takes what is left
            rows = h - row;
        row += rows;
        r0[++k] = row;
    }
    return k;   // number of bands; band i = rows [r0[i], r0[i+1])
}

static int band_events(lumahip_ctx *c, int nb)
{
    for (int k = 0; k < nb; k++)
        if (!c->band_h2d[k]) {
            HIPCHK(c, hipEventCreateWithFlags(&c->band_h2d[k], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->band_kern[k], hipEventDisableTiming));
        }
    if (!c->d_band_stats)
        HIPCHK(c, hipMalloc(&c->d_band_stats, lumahip_ctx::MAX_BANDS * 3 * sizeof(float)));
    return LUMAHIP_OK;
}

static int encode_frame_host_impl(lumahip_ctx *c, const float *rgb, unsigned w, unsigned h, float sc, int profile,
                                  unsigned char *const planes[3], const int stride[3], float *mean_lum,
                                  float *transformed_out, int cs_eff)
{
    if (!c || !rgb || !planes || !stride)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    c->up_ramp = 0;
    const int bps = profile > 1 ? 2 : 1;
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (int p = 0; p < 3; p++)
        if (!planes[p] || stride[p] < L.row_bytes[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: null or stride %d < row bytes %d", p, stride[p], L.row_bytes[p]);
    (void)bps;
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = ensure(c, (void **)&c->d_frame, &c->d_frame_cap, nfl * sizeof(float))))
        return rc;
    if ((rc = ensure(c, (void **)&c->d_planes, &c->d_planes_cap, L.total)))
        return rc;
    if (!c->d_stats)
        HIPCHK(c, hipMalloc(&c->d_stats, 3 * sizeof(float)));
    unsigned char *dp[3] = {c->d_planes + L.off[0], c->d_planes + L.off[1], c->d_planes + L.off[2]};
    const size_t pfs[3] = {0, 0, 0};
    const size_t n1 = (size_t)w * h;
    const bool sub = (profile == 0 || profile == 2);
    unsigned band0[lumahip_ctx::MAX_BANDS + 1];
    const int nb = band_plan(c, w, h, band0);
    float st[3] = {0.0f, __builtin_inff(), -__builtin_inff()};
    // Half upload (xfer_h2d_f16): tried unless the caller wants the transformed FLOAT frame back.  frame16: every band so far went
    // up as halves; mixed: some did and then a band held other values -- the device frame is then not usable as a whole.
    // (nor for frames that are already colour-transformed -- lumahip_pack_frame_host: such values are never halves)
    bool use16 = in16_try(c, w, transformed_out != nullptr || cs_eff == CS_PACK), frame16 = use16, mixed = false;
    const bool tried16 = use16;
    if (nb > 1) {
        if ((rc = pipe_streams(c)) || (rc = band_events(c, nb)))
            return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));   // the bands run on the pipeline streams: after everything queued so far
        hipStream_t saved = c->stream;
        DnGuard dn_guard{c, false, 0};   // a failing exit below drops the download chunks still pointing at the caller's planes
        auto fetch = [&](int k) -> int {              // planes rows of band k, after its kernel
            const unsigned r0 = band0[k], rows = band0[k + 1] - r0;
            HIPCHK(c, hipStreamWaitEvent(c->s_d2h, c->band_kern[k], 0));
            for (int p = 0; p < 3; p++) {
                const unsigned pr0 = (p && sub) ? r0 / 2 : r0, prow = (p && sub) ? rows / 2 : rows;
                const size_t off = (size_t)pr0 * stride[p];
                if (int r = xfer_d2h_2d(c, planes[p] + off, stride[p], dp[p] + off, stride[p], L.row_bytes[p], prow, c->s_d2h, true))
                    return r;
            }
            return LUMAHIP_OK;
        };
        for (int k = 0; k < nb && rc == LUMAHIP_OK; k++) {
            const unsigned r0 = band0[k], rows = band0[k + 1] - r0;
            const size_t roff = (size_t)r0 * w;
            uint16_t *const d16 = reinterpret_cast<uint16_t *>(c->d_frame);
            if (use16) {
                bool exact = true;
                for (int ch = 0; ch < 3 && rc == LUMAHIP_OK && exact; ch++)
                    rc = xfer_h2d_f16(c, d16 + ch * n1 + roff, rgb + ch * n1 + roff, (size_t)rows * w, c->s_h2d, &exact);
                if (rc)
                    break;
                if (!exact) {
                    // this band holds values that are not halves: it and the rest of the frame go up as floats -- into the same
                    // buffer, at float offsets, which the kernels of the earlier bands may still be reading as halves: wait for them
                    use16 = frame16 = false;
                    mixed = k > 0;
                    HIPCHK(c, hipStreamSynchronize(c->s_kern));
                }
            }
            if (!use16) {
                for (int ch = 0; ch < 3 && rc == LUMAHIP_OK; ch++)
                    rc = xfer_h2d(c, c->d_frame + ch * n1 + roff, rgb + ch * n1 + roff, (size_t)rows * w * sizeof(float), c->s_h2d);
                if (rc)
                    break;
            }
            HIPCHK(c, hipEventRecord(c->band_h2d[k], c->s_h2d));
            HIPCHK(c, hipStreamWaitEvent(c->s_kern, c->band_h2d[k], 0));
            // (binary16 planes: the same element offsets behind a pointer to halves)
            const float *const fp[3] = {use16 ? reinterpret_cast<const float *>(d16 + roff) : c->d_frame + roff,
                                        use16 ? reinterpret_cast<const float *>(d16 + n1 + roff) : c->d_frame + n1 + roff,
                                        use16 ? reinterpret_cast<const float *>(d16 + 2 * n1 + roff) : c->d_frame + 2 * n1 + roff};
            unsigned char *bp[3];
            for (int p = 0; p < 3; p++)
                bp[p] = dp[p] + (size_t)((p && sub) ? r0 / 2 : r0) * stride[p];
            c->stream = c->s_kern;
            rc = encode_frames_device_impl(c, fp, nfl, 1, w, rows, sc, profile, bp, stride, pfs, c->d_band_stats + 3 * k, cs_eff, false, use16);
            c->stream = saved;
            if (rc)
                break;
            HIPCHK(c, hipEventRecord(c->band_kern[k], c->s_kern));
            if (k >= 1)
                rc = fetch(k - 1);
        }
        if (rc == LUMAHIP_OK)
            rc = fetch(nb - 1);
        if (int r = d2h_flush(c))   // the chunks still in flight (also after an error: nothing may stay pending)
            rc = rc ? rc : r;
        else
            dn_guard.armed = false;
        c->stream = saved;
        HIPCHK(c, hipStreamSynchronize(c->s_h2d));
        HIPCHK(c, hipStreamSynchronize(c->s_kern));
        HIPCHK(c, hipStreamSynchronize(c->s_d2h));
        if (rc)
            return rc;
        float bs[lumahip_ctx::MAX_BANDS * 3];
        if ((rc = read_small(c, bs, c->d_band_stats, 3 * nb, c->stream)))
            return rc;
        for (int k = 0; k < nb; k++) {
            st[0] += bs[3 * k];
            st[1] = fminf(st[1], bs[3 * k + 1]);
            st[2] = fmaxf(st[2], bs[3 * k + 2]);
        }
    } else {
        uint16_t *const d16 = reinterpret_cast<uint16_t *>(c->d_frame);
        if (use16) {
            bool exact = true;
            if ((rc = xfer_h2d_f16(c, d16, rgb, nfl, c->stream, &exact)))
                return rc;
            if (!exact)
                use16 = frame16 = false;   // (nothing has been launched on the halves: the floats simply follow on the same stream)
        }
        if (!use16 && (rc = xfer_h2d(c, c->d_frame, rgb, nfl * sizeof(float), c->stream)))
            return rc;
        const float *const fp[3] = {use16 ? reinterpret_cast<const float *>(d16) : c->d_frame,
                                    use16 ? reinterpret_cast<const float *>(d16 + n1) : c->d_frame + n1,
                                    use16 ? reinterpret_cast<const float *>(d16 + 2 * n1) : c->d_frame + 2 * n1};
        if ((rc = encode_frames_device_impl(c, fp, nfl, 1, w, h, sc, profile, dp, stride, pfs, c->d_stats, cs_eff, false, use16)))
            return rc;
        for (int p = 0; p < 3; p++)
            if ((rc = xfer_d2h_2d(c, planes[p], stride[p], dp[p], stride[p], L.row_bytes[p], L.rows[p], c->stream)))
                return rc;
    }
    if (transformed_out) {
        rc = lumahip_transform_color_space_device(c, c->d_frame, nfl, 1, w, h, 1, sc);
        if (rc)
            return rc;
        if ((rc = xfer_d2h(c, transformed_out, c->d_frame, nfl * sizeof(float), c->stream)))
            return rc;
    }
    if (nb == 1 && (rc = read_small(c, st, c->d_stats, 3, c->stream)))  // synchronises the stream
        return rc;
    if (transformed_out)
        HIPCHK(c, hipStreamSynchronize(c->stream));
    if (tried16)
        in16_result(c, frame16);
    if (mean_lum) {
        *mean_lum = st[0] / (float)((int)w * (int)h);  // avg /= (w*h), src/luma_encoder.cpp:314
        if (mean_needs_reference_sum(*mean_lum, st[1], w, h)) {  // d_frame holds the caller's frame (as floats or as halves), or already its transformed version
            if (mixed && (rc = xfer_h2d(c, c->d_frame, rgb, nfl * sizeof(float), c->stream)))   // part halves, part floats: once more, whole
                return rc;
            return transformed_out ? seq_mean(c, c->d_frame, w, h, mean_lum)
                                   : mean_luminance_reference_impl(c, c->d_frame, w, h, sc, cs_eff, mean_lum, frame16);
        }
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_encode_frame_host(lumahip_ctx *c, const float *rgb, unsigned w, unsigned h, float sc, int profile,
                                         unsigned char *const planes[3], const int stride[3], float *mean_lum,
                                         float *transformed_out)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    return encode_frame_host_impl(c, rgb, w, h, sc, profile, planes, stride, mean_lum, transformed_out, c->q.cs);
}

static int decode_frame_host_impl(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3], unsigned w,
                                  unsigned h, int profile, float sc, float *rgb_out, int cs_eff)
{
    if (!c || !rgb_out || !planes || !stride)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, cs_eff);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    c->up_ramp = 0;
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (int p = 0; p < 3; p++)
        if (!planes[p] || stride[p] < L.row_bytes[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: null or stride %d < row bytes %d", p, stride[p], L.row_bytes[p]);
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = ensure(c, (void **)&c->d_frame, &c->d_frame_cap, nfl * sizeof(float))))
        return rc;
    if ((rc = ensure(c, (void **)&c->d_planes, &c->d_planes_cap, L.total)))
        return rc;
    unsigned char *dp[3] = {c->d_planes + L.off[0], c->d_planes + L.off[1], c->d_planes + L.off[2]};
    const size_t pfs[3] = {0, 0, 0};
    const size_t n1 = (size_t)w * h;
    const bool sub = (profile == 0 || profile == 2);
    unsigned band0[lumahip_ctx::MAX_BANDS + 1];
    const int nb = band_plan(c, w, h, band0);
    if (nb > 1) {
        if ((rc = pipe_streams(c)) || (rc = band_events(c, nb)))
            return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipStream_t saved = c->stream;
        DnGuard dn_guard{c, false, 0};   // a failing exit below drops the download chunks still pointing at the caller's planes
        auto fetch = [&](int k) -> int {              // float rows of band k, after its kernel
            const unsigned r0 = band0[k], rows = band0[k + 1] - r0;
            const size_t roff = (size_t)r0 * w;
            HIPCHK(c, hipStreamWaitEvent(c->s_d2h, c->band_kern[k], 0));
            for (int ch = 0; ch < 3; ch++)
                if (int r = xfer_d2h_deferred(c, rgb_out + ch * n1 + roff, c->d_frame + ch * n1 + roff, (size_t)rows * w * sizeof(float), c->s_d2h))
                    return r;
            return LUMAHIP_OK;
        };
        for (int k = 0; k < nb && rc == LUMAHIP_OK; k++) {
            const unsigned r0 = band0[k], rows = band0[k + 1] - r0;
            const size_t roff = (size_t)r0 * w;
            const unsigned char *bp[3];
            for (int p = 0; p < 3 && rc == LUMAHIP_OK; p++) {
                const unsigned pr0 = (p && sub) ? r0 / 2 : r0, prow = (p && sub) ? rows / 2 : rows;
                const size_t off = (size_t)pr0 * stride[p];
                bp[p] = dp[p] + off;
                rc = xfer_h2d_2d(c, dp[p] + off, stride[p], planes[p] + off, stride[p], L.row_bytes[p], prow, c->s_h2d);
            }
            if (rc)
                break;
            HIPCHK(c, hipEventRecord(c->band_h2d[k], c->s_h2d));
            HIPCHK(c, hipStreamWaitEvent(c->s_kern, c->band_h2d[k], 0));
            float *const fp[3] = {c->d_frame + roff, c->d_frame + n1 + roff, c->d_frame + 2 * n1 + roff};
            c->stream = c->s_kern;
            rc = decode_impl(c, bp, stride, pfs, 1, w, rows, profile, sc, fp, nfl, DisplayParams(), cs_eff);
            c->stream = saved;
            if (rc)
                break;
            HIPCHK(c, hipEventRecord(c->band_kern[k], c->s_kern));
            if (k >= 1)
                rc = fetch(k - 1);
        }
        if (rc == LUMAHIP_OK)
            rc = fetch(nb - 1);
        if (int r = d2h_flush(c))   // the chunks still in flight (also after an error: nothing may stay pending)
            rc = rc ? rc : r;
        else
            dn_guard.armed = false;
        c->stream = saved;
        HIPCHK(c, hipStreamSynchronize(c->s_h2d));
        HIPCHK(c, hipStreamSynchronize(c->s_kern));
        HIPCHK(c, hipStreamSynchronize(c->s_d2h));
        return rc;
    }
    for (int p = 0; p < 3; p++)
        if ((rc = xfer_h2d_2d(c, dp[p], stride[p], planes[p], stride[p], L.row_bytes[p], L.rows[p], c->stream)))
            return rc;
    {
        float *const fp[3] = {c->d_frame, c->d_frame + n1, c->d_frame + 2 * n1};
        rc = decode_impl(c, dp, stride, pfs, 1, w, h, profile, sc, fp, nfl, DisplayParams(), cs_eff);
    }
    if (rc)
        return rc;
    if ((rc = xfer_d2h(c, rgb_out, c->d_frame, nfl * sizeof(float), c->stream)))
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

extern "C" int lumahip_decode_frame_host(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                         unsigned w, unsigned h, int profile, float sc, float *rgb_out)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    return decode_frame_host_impl(c, planes, stride, w, h, profile, sc, rgb_out, c->q.cs);
}

// ---- batched host entry points: a 3-slot software pipeline over three streams.  Frame i's H2D copy runs while
// frame i-1's kernel and frame i-2's D2H copies are in flight; with pinned caller memory (lumahip_host_register) the
// two copy directions overlap as well and the rate approaches the PCIe H2D rate.
static int pipe_streams(lumahip_ctx *c)
{
    if (!c->s_h2d) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_kern, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking));
    }
    return LUMAHIP_OK;
}

static int pipe_prepare(lumahip_ctx *c, size_t frame_bytes, size_t planes_bytes, unsigned nframes)
{
    if (int rc = pipe_streams(c))
        return rc;
    if (!c->slot[0].h2d) {
        for (auto &sl : c->slot) {
            HIPCHK(c, hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&sl.kern, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&sl.d2h, hipEventDisableTiming));
            HIPCHK(c, hipMalloc(&sl.d_stats, 3 * sizeof(float)));
        }
    }
    if (c->slot_frame_cap < frame_bytes || c->slot_planes_cap < planes_bytes) {
        HIPCHK(c, hipDeviceSynchronize());
        for (auto &sl : c->slot) {
            (void)hipFree(sl.d_frame);
            (void)hipFree(sl.d_planes);
            sl.d_frame = nullptr;
            sl.d_planes = nullptr;
            HIPCHK(c, hipMalloc(&sl.d_frame, frame_bytes));
            HIPCHK(c, hipMalloc(&sl.d_planes, planes_bytes));
        }
        c->slot_frame_cap = frame_bytes;
        c->slot_planes_cap = planes_bytes;
    }
    if (c->h_stats_cap < nframes) {
        if (c->h_stats)
            (void)hipHostFree(c->h_stats);
        c->h_stats = nullptr;
        HIPCHK(c, hipHostMalloc(&c->h_stats, (size_t)nframes * 3 * sizeof(float), hipHostMallocDefault));
        c->h_stats_cap = nframes;
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_encode_frames_host(lumahip_ctx *c, const float *const *rgb, unsigned nframes, unsigned w, unsigned h,
                                          float sc, int profile, unsigned char *const *planes, const int stride[3],
                                          float *mean_lum)
{
    if (!c || !rgb || !planes || !stride || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, c->q.cs);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (unsigned i = 0; i < nframes; i++) {
        if (!rgb[i])
            return fail(c, LUMAHIP_ERR_ARG, "null frame %u", i);
        for (int p = 0; p < 3; p++)
            if (!planes[3 * i + p] || stride[p] < L.row_bytes[p])
                return fail(c, LUMAHIP_ERR_ARG, "frame %u plane %d: null or stride too small", i, p);
    }
    const size_t nfl = (size_t)3 * w * h;
    if (c->es_head != c->es_tail)
        return fail(c, LUMAHIP_ERR_STATE, "frames pushed with lumahip_encode_stream_push / lumahip_decode_stream_push are still pending: pop them first");
    if ((rc = pipe_prepare(c, nfl * sizeof(float), L.total, nframes)))
        return rc;
    if ((rc = dn_chunks_for(c, L.total)))
        return rc;
    hipStream_t saved = c->stream;
    DnGuard dn_guard{c, false, 0};   // a failing exit drops the download chunks still pointing at the caller's buffers
    const size_t pfs[3] = {0, 0, 0};
    // Frame i's upload and kernel are queued BEFORE frame i-1's planes are fetched.
    auto fetch = [&](unsigned i) -> int {
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
        (void)hipStreamWaitEvent(c->s_d2h, sl.kern, 0);
        int r = LUMAHIP_OK;
        // (deferred: pageable planes are copied out of the staging chunks when the ring comes round to them or by the
        // d2h_flush below -- not here, where it would hold up the staging of the next frame's upload)
        for (int p = 0; p < 3 && r == LUMAHIP_OK; p++)
            r = xfer_d2h_2d(c, planes[3 * i + p], stride[p], dp[p], stride[p], L.row_bytes[p], L.rows[p], c->s_d2h, true);
        if (r)
            return r;
        (void)hipMemcpyAsync(c->h_stats + 3 * (size_t)i, sl.d_stats, 3 * sizeof(float), hipMemcpyDeviceToHost, c->s_d2h);
        (void)hipEventRecord(sl.d2h, c->s_d2h);
        return LUMAHIP_OK;
    };
    for (unsigned i = 0; i < nframes && rc == LUMAHIP_OK; i++) {
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
        if (i >= 3) {
            // slot reuse: the kernel of frame i-3 must have consumed d_frame, its D2H must have drained d_planes
            (void)hipStreamWaitEvent(c->s_h2d, sl.kern, 0);
            (void)hipStreamWaitEvent(c->s_kern, sl.d2h, 0);
        }
        // half upload (xfer_h2d_f16) where the frame holds halves; a frame that does not goes up as floats behind whatever part
        // of it went up as halves (same stream, same slot, nothing launched on it yet)
        bool f16 = in16_try(c, w, false);
        if (f16) {
            bool exact = true;
            if ((rc = xfer_h2d_f16(c, sl.d_frame, rgb[i], nfl, c->s_h2d, &exact)))
                break;
            in16_result(c, exact);
            f16 = exact;
        }
        if (!f16 && (rc = xfer_h2d(c, sl.d_frame, rgb[i], nfl * sizeof(float), c->s_h2d)))
            break;
        (void)hipEventRecord(sl.h2d, c->s_h2d);
        (void)hipStreamWaitEvent(c->s_kern, sl.h2d, 0);
        c->stream = c->s_kern;
        rc = f16 ? encode_packed16(c, sl.d_frame, nfl, 1, w, h, sc, profile, dp, stride, pfs, sl.d_stats)
                 : encode_packed(c, sl.d_frame, nfl, 1, w, h, sc, profile, dp, stride, pfs, sl.d_stats);
        c->stream = saved;
        if (rc)
            break;
        (void)hipEventRecord(sl.kern, c->s_kern);
        if (i >= 1)
            rc = fetch(i - 1);
    }
    if (rc == LUMAHIP_OK)
        rc = fetch(nframes - 1);
    if (int r = d2h_flush(c))   // (also after an error: nothing may stay pending)
        rc = rc ? rc : r;
    else
        dn_guard.armed = false;
    c->stream = saved;
    HIPCHK(c, hipStreamSynchronize(c->s_h2d));
    HIPCHK(c, hipStreamSynchronize(c->s_kern));
    HIPCHK(c, hipStreamSynchronize(c->s_d2h));
    if (rc == LUMAHIP_OK && mean_lum)
        for (unsigned i = 0; i < nframes && rc == LUMAHIP_OK; i++) {
            mean_lum[i] = c->h_stats[3 * (size_t)i] / (float)((int)w * (int)h);
            if (mean_needs_reference_sum(mean_lum[i], c->h_stats[3 * (size_t)i + 1], w, h)) {  // rare: redo this frame's sum in the reference's order
                if ((rc = xfer_h2d(c, c->slot[0].d_frame, rgb[i], nfl * sizeof(float), c->stream)))
                    return rc;
                rc = mean_luminance_reference_impl(c, c->slot[0].d_frame, w, h, sc, c->q.cs, &mean_lum[i]);
            }
        }
    return rc;
}

// ---- streaming form of the batched encode: frames arrive one at a time ------------------------------------------------------
// lumahip_encode_frames_host needs the whole batch in hand.  A caller that gets its frames one by one (the reference's
// `for (...) encoder.encode(&frame)` loop, lumaenc.cpp:205-243) can still overlap the tail of frame i (kernel, download,
// copy out of the staging chunks) with the upload of frame i+1 by accepting ONE frame of latency: push(i+1), then pop(i).
// Same three device slots and three streams as the batched form; at most two frames in flight.
extern "C" int lumahip_encode_stream_push(lumahip_ctx *c, const float *rgb, unsigned w, unsigned h, float sc, int profile,
                                          unsigned char *const planes[3], const int stride[3])
{
    if (!c || !rgb || !planes || !stride)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, c->q.cs);
    if (rc)
        return rc;
    if (c->es_head != c->es_tail && c->es_dir != 0)
        return fail(c, LUMAHIP_ERR_STATE, "frames pushed with lumahip_decode_stream_push are in flight: pop them first");
    if (c->es_head - c->es_tail >= 2)
        return fail(c, LUMAHIP_ERR_STATE, "two frames are in flight already: lumahip_encode_stream_pop the oldest first");
    if (c->es_head != c->es_tail && (w != c->es_w || h != c->es_h || profile != c->es_profile || stride[0] != c->es_stride[0] ||
                                     stride[1] != c->es_stride[1] || stride[2] != c->es_stride[2]))
        return fail(c, LUMAHIP_ERR_STATE, "frame geometry (size, profile or plane strides) changed while a frame is in flight: pop it first");
    HIPCHK(c, hipSetDevice(c->device));
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (int p = 0; p < 3; p++)
        if (!planes[p] || stride[p] < L.row_bytes[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: null or stride too small", p);
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = pipe_prepare(c, nfl * sizeof(float), L.total, 1)))   // (reallocates only when nothing is in flight: same geometry otherwise)
        return rc;
    if (c->es_head == c->es_tail && (rc = dn_chunks_for(c, L.total)))
        return rc;
    if (!c->h_es_stats)
        HIPCHK(c, hipHostMalloc((void **)&c->h_es_stats, 3 * 3 * sizeof(float), hipHostMallocDefault));
    const unsigned seq = c->es_head;
    lumahip_ctx::Slot &sl = c->slot[seq % 3];
    unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
    const size_t pfs[3] = {0, 0, 0};
    // the slot's previous occupant (frame seq - 3) was popped long ago; its kernel and downloads are done, but the streams
    // still have to be told (events of that occupancy)
    if (seq >= 3) {
        (void)hipStreamWaitEvent(c->s_h2d, sl.kern, 0);
        (void)hipStreamWaitEvent(c->s_kern, sl.d2h, 0);
    }
    c->up_ramp = 0;
    DnGuard dn_guard{c, false, seq + 1};   // until the frame counts as pushed, a failure drops the download chunks queued for it
    bool pinned_in = host_range_is_pinned(rgb, nfl * sizeof(float));
    bool f16 = in16_try(c, w, false);
    if (f16) {   // half upload: the CPU converts out of the caller's memory (pinned or not), so it is free again when this returns
        bool exact = true;
        if ((rc = xfer_h2d_f16(c, sl.d_frame, rgb, nfl, c->s_h2d, &exact)))
            return rc;
        in16_result(c, exact);
        f16 = exact;
    }
    if (f16)
        pinned_in = false;
    else if ((rc = xfer_h2d(c, sl.d_frame, rgb, nfl * sizeof(float), c->s_h2d)))
        return rc;
    c->slot_in16[seq % 3] = f16;
    (void)hipEventRecord(sl.h2d, c->s_h2d);
    (void)hipStreamWaitEvent(c->s_kern, sl.h2d, 0);
    hipStream_t saved = c->stream;
    c->stream = c->s_kern;
    rc = f16 ? encode_packed16(c, sl.d_frame, nfl, 1, w, h, sc, profile, dp, stride, pfs, sl.d_stats)
             : encode_packed(c, sl.d_frame, nfl, 1, w, h, sc, profile, dp, stride, pfs, sl.d_stats);
    c->stream = saved;
    if (rc)
        return rc;
    (void)hipEventRecord(sl.kern, c->s_kern);
    // the planes come down behind the kernel; pageable ones are emptied out of the staging chunks by the pop (or earlier,
    // when the ring comes round)
    (void)hipStreamWaitEvent(c->s_d2h, sl.kern, 0);
    c->d2h_tag = seq + 1;   // (0 = not a pushed frame)
    for (int p = 0; p < 3 && rc == LUMAHIP_OK; p++)
        rc = xfer_d2h_2d(c, planes[p], stride[p], dp[p], stride[p], L.row_bytes[p], L.rows[p], c->s_d2h, true);
    c->d2h_tag = 0;
    if (rc)
        return rc;
    (void)hipMemcpyAsync(c->h_es_stats + 3 * (seq % 3), sl.d_stats, 3 * sizeof(float), hipMemcpyDeviceToHost, c->s_d2h);
    (void)hipEventRecord(sl.d2h, c->s_d2h);
    if (pinned_in)
        HIPCHK(c, hipEventSynchronize(sl.h2d));   // the copy engine read the caller's memory directly: it must be done with it
    c->es_w = w;
    c->es_h = h;
    c->es_profile = profile;
    c->es_sc = sc;
    c->es_total = L.total;
    for (int p = 0; p < 3; p++)
        c->es_stride[p] = stride[p];
    c->es_dir = 0;
    c->es_head = seq + 1;
    dn_guard.armed = false;   // the pop completes them
    return LUMAHIP_OK;
}

extern "C" int lumahip_encode_stream_pop(lumahip_ctx *c, float *mean_lum)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (c->es_head == c->es_tail || c->es_dir != 0)
        return fail(c, LUMAHIP_ERR_STATE, "no encode frame is in flight");
    HIPCHK(c, hipSetDevice(c->device));
    const unsigned seq = c->es_tail;
    lumahip_ctx::Slot &sl = c->slot[seq % 3];
    c->es_tail = seq + 1;                        // (popped even if something below fails: nothing may stay half-finished)
    int rc = d2h_flush_upto(c, seq);
    HIPCHK(c, hipEventSynchronize(sl.d2h));      // downloads into pinned planes, and the statistics
    if (rc)
        return rc;
    if (mean_lum) {
        const float *stp = c->h_es_stats + 3 * (seq % 3);
        *mean_lum = stp[0] / (float)((int)c->es_w * (int)c->es_h);
        if (mean_needs_reference_sum(*mean_lum, stp[1], c->es_w, c->es_h))   // the slot still holds the frame as it was uploaded
            return mean_luminance_reference_impl(c, sl.d_frame, c->es_w, c->es_h, c->es_sc, c->q.cs, mean_lum, c->slot_in16[seq % 3]);
    }
    return LUMAHIP_OK;
}

extern "C" int lumahip_encode_stream_pending(const lumahip_ctx *c) { return (c && c->es_dir == 0) ? (int)(c->es_head - c->es_tail) : 0; }

// The decode counterpart (LumaDecoder::decode() in a loop, lumadec.cpp:112-160): the download of frame i (12 B/pixel, the heavy
// direction here) keeps the copy engine busy while the planes of frame i+1 go up and its kernel runs.
extern "C" int lumahip_decode_stream_push(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3], unsigned w,
                                          unsigned h, int profile, float sc, float *rgb_out)
{
    if (!c || !rgb_out || !planes || !stride)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, c->q.cs);
    if (rc)
        return rc;
    if (c->es_head != c->es_tail && c->es_dir != 1)
        return fail(c, LUMAHIP_ERR_STATE, "frames pushed with lumahip_encode_stream_push are in flight: pop them first");
    if (c->es_head - c->es_tail >= 2)
        return fail(c, LUMAHIP_ERR_STATE, "two frames are in flight already: lumahip_decode_stream_pop the oldest first");
    if (c->es_head != c->es_tail && (w != c->es_w || h != c->es_h || profile != c->es_profile || stride[0] != c->es_stride[0] ||
                                     stride[1] != c->es_stride[1] || stride[2] != c->es_stride[2]))
        return fail(c, LUMAHIP_ERR_STATE, "frame geometry (size, profile or plane strides) changed while a frame is in flight: pop it first");
    HIPCHK(c, hipSetDevice(c->device));
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (int p = 0; p < 3; p++)
        if (!planes[p] || stride[p] < L.row_bytes[p])
            return fail(c, LUMAHIP_ERR_ARG, "plane %d: null or stride too small", p);
    const size_t nfl = (size_t)3 * w * h;
    if ((rc = pipe_prepare(c, nfl * sizeof(float), L.total, 1)))
        return rc;
    const unsigned seq = c->es_head;
    lumahip_ctx::Slot &sl = c->slot[seq % 3];
    unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
    const size_t pfs[3] = {0, 0, 0};
    if (seq >= 3) {
        (void)hipStreamWaitEvent(c->s_h2d, sl.kern, 0);   // planes of the slot's previous occupant consumed
        (void)hipStreamWaitEvent(c->s_kern, sl.d2h, 0);   // its floats downloaded
    }
    c->up_ramp = 0;
    DnGuard dn_guard{c, false, seq + 1};   // until the frame counts as pushed, a failure drops the download chunks queued for it
    bool pinned_in = true;
    for (int p = 0; p < 3 && rc == LUMAHIP_OK; p++) {
        pinned_in = pinned_in && host_range_is_pinned(planes[p], (size_t)(L.rows[p] - 1) * stride[p] + L.row_bytes[p]);
        rc = xfer_h2d_2d(c, dp[p], stride[p], planes[p], stride[p], L.row_bytes[p], L.rows[p], c->s_h2d);
    }
    if (rc)
        return rc;
    (void)hipEventRecord(sl.h2d, c->s_h2d);
    (void)hipStreamWaitEvent(c->s_kern, sl.h2d, 0);
    hipStream_t saved = c->stream;
    c->stream = c->s_kern;
    rc = decode_packed(c, dp, stride, pfs, 1, w, h, profile, sc, sl.d_frame, nfl);
    c->stream = saved;
    if (rc)
        return rc;
    (void)hipEventRecord(sl.kern, c->s_kern);
    (void)hipStreamWaitEvent(c->s_d2h, sl.kern, 0);
    c->d2h_tag = seq + 1;   // (0 = not a pushed frame)
    rc = xfer_d2h_deferred(c, rgb_out, sl.d_frame, nfl * sizeof(float), c->s_d2h);
    c->d2h_tag = 0;
    if (rc)
        return rc;
    (void)hipEventRecord(sl.d2h, c->s_d2h);
    if (pinned_in)   // (any pinned plane was read by the copy engine directly: it must be done with the caller's memory)
        HIPCHK(c, hipEventSynchronize(sl.h2d));
    c->es_w = w;
    c->es_h = h;
    c->es_profile = profile;
    c->es_sc = sc;
    c->es_total = L.total;
    for (int p = 0; p < 3; p++)
        c->es_stride[p] = stride[p];
    c->es_dir = 1;
    c->es_head = seq + 1;
    dn_guard.armed = false;   // the pop completes them
    return LUMAHIP_OK;
}

extern "C" int lumahip_decode_stream_pop(lumahip_ctx *c)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (c->es_head == c->es_tail || c->es_dir != 1)
        return fail(c, LUMAHIP_ERR_STATE, "no decode frame is in flight");
    HIPCHK(c, hipSetDevice(c->device));
    const unsigned seq = c->es_tail;
    lumahip_ctx::Slot &sl = c->slot[seq % 3];
    c->es_tail = seq + 1;
    const int rc = d2h_flush_upto(c, seq);
    HIPCHK(c, hipEventSynchronize(sl.d2h));      // (a download into pinned memory has no chunks to flush)
    return rc;
}

extern "C" int lumahip_decode_stream_pending(const lumahip_ctx *c) { return (c && c->es_dir == 1) ? (int)(c->es_head - c->es_tail) : 0; }

extern "C" int lumahip_decode_frames_host(lumahip_ctx *c, const unsigned char *const *planes, const int stride[3],
                                          unsigned nframes, unsigned w, unsigned h, int profile, float sc,
                                          float *const *rgb_out)
{
    if (!c || !rgb_out || !planes || !stride || nframes == 0)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    int rc = check_geom(c, w, h, profile, c->q.cs);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    PlaneLayout L;
    plane_layout(L, w, h, profile, stride);
    for (unsigned i = 0; i < nframes; i++) {
        if (!rgb_out[i])
            return fail(c, LUMAHIP_ERR_ARG, "null output frame %u", i);
        for (int p = 0; p < 3; p++)
            if (!planes[3 * i + p] || stride[p] < L.row_bytes[p])
                return fail(c, LUMAHIP_ERR_ARG, "frame %u plane %d: null or stride too small", i, p);
    }
    const size_t nfl = (size_t)3 * w * h;
    if (c->es_head != c->es_tail)
        return fail(c, LUMAHIP_ERR_STATE, "frames pushed with lumahip_encode_stream_push / lumahip_decode_stream_push are still pending: pop them first");
    if ((rc = pipe_prepare(c, nfl * sizeof(float), L.total, nframes)))
        return rc;
    hipStream_t saved = c->stream;
    DnGuard dn_guard{c, false, 0};   // a failing exit drops the download chunks still pointing at the caller's buffers
    const size_t pfs[3] = {0, 0, 0};
    auto fetch = [&](unsigned i) -> int {  // as in lumahip_encode_frames_host: frame i-1 is fetched after frame i is queued
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        (void)hipStreamWaitEvent(c->s_d2h, sl.kern, 0);
        int r = xfer_d2h_deferred(c, rgb_out[i], sl.d_frame, nfl * sizeof(float), c->s_d2h);   // (drained lazily, see stage_dn_ready)
        if (r)
            return r;
        (void)hipEventRecord(sl.d2h, c->s_d2h);
        return LUMAHIP_OK;
    };
    for (unsigned i = 0; i < nframes && rc == LUMAHIP_OK; i++) {
        lumahip_ctx::Slot &sl = c->slot[i % 3];
        unsigned char *dp[3] = {sl.d_planes + L.off[0], sl.d_planes + L.off[1], sl.d_planes + L.off[2]};
        if (i >= 3) {
            (void)hipStreamWaitEvent(c->s_h2d, sl.kern, 0);   // planes of frame i-3 consumed
            (void)hipStreamWaitEvent(c->s_kern, sl.d2h, 0);   // floats of frame i-3 copied out
        }
        for (int p = 0; p < 3 && rc == LUMAHIP_OK; p++)
            rc = xfer_h2d_2d(c, dp[p], stride[p], planes[3 * i + p], stride[p], L.row_bytes[p], L.rows[p], c->s_h2d);
        if (rc)
            break;
        (void)hipEventRecord(sl.h2d, c->s_h2d);
        (void)hipStreamWaitEvent(c->s_kern, sl.h2d, 0);
        c->stream = c->s_kern;
        rc = decode_packed(c, dp, stride, pfs, 1, w, h, profile, sc, sl.d_frame, nfl);
        c->stream = saved;
        if (rc)
            break;
        (void)hipEventRecord(sl.kern, c->s_kern);
        if (i >= 1)
            rc = fetch(i - 1);
    }
    if (rc == LUMAHIP_OK)
        rc = fetch(nframes - 1);
    if (int r = d2h_flush(c))   // (also after an error: nothing may stay pending)
        rc = rc ? rc : r;
    else
        dn_guard.armed = false;
    c->stream = saved;
    HIPCHK(c, hipStreamSynchronize(c->s_h2d));
    HIPCHK(c, hipStreamSynchronize(c->s_kern));
    HIPCHK(c, hipStreamSynchronize(c->s_d2h));
    return rc;
}

extern "C" int lumahip_transform_color_space_host(lumahip_ctx *c, float *frame, unsigned w, unsigned h, int toCs, float sc)
{
    if (!c || !frame)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    if (c->q.cs < 0 || c->q.cs > 3)
        return fail(c, LUMAHIP_ERR_UNSUPPORTED, "Error! Unrecognized color transformation");
    if (w == 0 || h == 0)
        return LUMAHIP_OK;  // the reference loops zero times and returns true
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)w * h, nfl = 3 * n;
    // device copy padded to an even pixel count per channel so the pair kernel applies to odd sizes too
    const size_t npad = (n + 1) & ~(size_t)1;
    int rc = ensure(c, (void **)&c->d_frame, &c->d_frame_cap, 3 * npad * sizeof(float));
    if (rc)
        return rc;
    if (npad == n) {
        if ((rc = xfer_h2d(c, c->d_frame, frame, nfl * sizeof(float), c->stream)))
            return rc;
    } else {
        HIPCHK(c, hipMemsetAsync(c->d_frame, 0, 3 * npad * sizeof(float), c->stream));
        for (int ch = 0; ch < 3; ch++)
            if ((rc = xfer_h2d(c, c->d_frame + ch * npad, frame + ch * n, n * sizeof(float), c->stream)))
                return rc;
    }
    // the kernel addresses channels at chan_stride = (w*h); present the padded buffer as a (npad x 1) frame
    rc = lumahip_transform_color_space_device(c, c->d_frame, 3 * npad, 1, (unsigned)npad, 1, toCs, sc);
    if (rc)
        return rc;
    if (npad == n) {
        if ((rc = xfer_d2h(c, frame, c->d_frame, nfl * sizeof(float), c->stream)))
            return rc;
    } else {
        for (int ch = 0; ch < 3; ch++)
            if ((rc = xfer_d2h(c, frame + ch * n, c->d_frame + ch * npad, n * sizeof(float), c->stream)))
                return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

static int array_op(lumahip_ctx *c, const float *in, float *out, size_t n, unsigned ch, bool quant)
{
    if (!c || !in || !out)
        return fail(c, LUMAHIP_ERR_ARG, "null argument");
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    if (n == 0)
        return LUMAHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure(c, (void **)&c->d_arr, &c->d_arr_cap, 2 * n * sizeof(float));
    if (rc)
        return rc;
    if ((rc = xfer_h2d(c, c->d_arr, in, n * sizeof(float), c->stream)))
        return rc;
    if ((rc = array_launch(c, c->d_arr, c->d_arr + n, n, ch, quant)))
        return rc;
    if ((rc = xfer_d2h(c, out, c->d_arr + n, n * sizeof(float), c->stream)))
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUMAHIP_OK;
}

extern "C" int lumahip_quantize_array_host(lumahip_ctx *c, const float *in, float *out, size_t n, unsigned ch)
{
    return array_op(c, in, out, n, ch, true);
}

extern "C" int lumahip_dequantize_array_host(lumahip_ctx *c, const float *in, float *out, size_t n, unsigned ch)
{
    return array_op(c, in, out, n, ch, false);
}

// LumaEncoder::setChannels / LumaDecoder::getVpxChannels on their own: no colour transform.  Channel 0
// goes through the LUT; channels 1,2 through the LUT for RGB / XYZ (src/luma_quantizer.cpp:219,251) --
// which is the CS_RGB kernel with sc = 1 (x*1.0f and x/1.0f are exact) -- and through the colour quantizer
// otherwise (CS_PACK).
static int pack_cs(const lumahip_ctx *c) { return (c->q.cs == CS_RGB || c->q.cs == CS_XYZ) ? CS_RGB : CS_PACK; }

extern "C" int lumahip_pack_frame_host(lumahip_ctx *c, const float *transformed, unsigned w, unsigned h, int profile,
                                       unsigned char *const planes[3], const int stride[3], float *mean_lum)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    return encode_frame_host_impl(c, transformed, w, h, 1.0f, profile, planes, stride, mean_lum, nullptr, pack_cs(c));
}

extern "C" int lumahip_unpack_frame_host(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3],
                                         unsigned w, unsigned h, int profile, float *dequantized_out)
{
    if (!c)
        return LUMAHIP_ERR_ARG;
    if (!c->have_quant)
        return fail(c, LUMAHIP_ERR_STATE, "quantizer not set");
    return decode_frame_host_impl(c, planes, stride, w, h, profile, 1.0f, dequantized_out, pack_cs(c));
}

