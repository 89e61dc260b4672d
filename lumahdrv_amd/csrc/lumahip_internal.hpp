// lumahip_internal.hpp -- what the translation units behind include/lumahip.h share: the context, error helpers,
// launch-geometry rules and the device-side implementations the host entry points call.  Not installed, not part of the ABI.
//
//   lumahip_core.hip    context life cycle, quantizer upload, layout checks, memory helpers               (no kernels)
//   lumahip_launch.hip  launch geometry: LDS bytes, threads per workgroup, persistent workgroups per CU   (no kernels)
//   lumahip_encode.hip  every k_encode / encode-side instantiation and its dispatch
//   lumahip_decode.hip  every k_decode instantiation and its dispatch
//   lumahip_misc.hip    stand-alone transform, synthetic frames, the reference's mean luminance, probes, timing helper
//   lumahip_host.hip    the _host entry points: staging, host <-> device transfers, the 3-slot pipeline   (no kernels)
//   lumahip_pool.hip    the HBM chunk pool;  lumahip_multi.hip  many GPUs in one process                  (no kernels)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#define LUMAHIP_EXPERIMENTAL   /* the library defines what the experimental section of the header declares */
#include "../../include/lumahip.h"
#include "luma_kernels.hpp"
#include "host_lut.hpp"
#include "lut_index.hpp"
#include "numa_host.hpp"

// Largest search table (encode: threshold records, decode: the luminance table) a workgroup stages in LDS; beyond it
// the table is read from global memory (L2-resident).  gfx950 has 160 KiB of LDS per CU; tables beyond 53 KiB run as one
// or two 1024-thread workgroups per CU -- for the 136 KiB of PQ 13-bit records that is still twice as fast as gathering
// them from L2 (346 against 182 Gpixel/s, profiles/r02_perf_matrix.txt).  LUMAHIP_LDS_TABLE_MAX_KB overrides it.
static constexpr size_t LUMAHIP_LDS_TABLE_MAX_DEFAULT = 144 * 1024;
static constexpr size_t LUMAHIP_LDS_PER_WORKGROUP = 160 * 1024;
static constexpr int LUMAHIP_MAX_LANES = 4;

// Kernel choice from feedback, as a function of the data and of nothing else (lumahip_core.hip lag_policy_*).  Two pairs of
// kernels compute the same results at different speeds depending on what the stream holds: the YCbCr encode kernels with the
// half-input table (fast on binary16-valued frames, 1.4 x slower than the per-pixel kernels on others) and the YCbCr decode
// kernels that may read red / blue from the per-stream tables (fast on pictures, a few per cent slower than the plain ones on
// unrelated pixels, where no wave ever takes the tables).  Every launch of the data-dependent ("fast") kernel gets its OWN word
// in a ring of pinned host memory, which the kernel sets to 1 as documented at EncArgs::half_flag / DecArgs::rb_flag, and an
// event recorded behind it; the host reads launch j's word when it issues the eligible launch LAG later, after that event has
// completed (normally long ago; at most LAG - 1 launches stay queued behind it, so the device does not run dry).  States:
//   ON_FAST     fast launches; a BAD word -> BACKOFF for 16 launches (the words of the other launches in flight are dropped);
//   BACKOFF     plain launches; when the count runs out, ONE fast launch probes -> PROBE_WAIT;
//   PROBE_WAIT  plain launches until the probe's word is read (LAG launches later): bad -> BACKOFF with the pause doubled (up
//               to max_backoff), good -> ON_FAST and the pause forgotten.
// max_backoff weighs a probe's cost against a missed switch: a half-table probe on float data costs 40 % of its launch (1024:
// 0.04 % of the stream), a red / blue probe on unrelated pixels 3 % while a picture decoded without the tables loses a third (64).
// `report_is_bad`: whether a set word (true) or a clear one (false) is the bad news.
struct LagPolicy {
    static constexpr int LAG = 4, RING = 8;
    enum { ON_FAST = 0, BACKOFF = 1, PROBE_WAIT = 2 };
    LagPolicy(bool bad_when_set, int longest_pause) : report_is_bad(bad_when_set), max_backoff(longest_pause) {}
    bool report_is_bad;
    int max_backoff;                                 // the pause doubles from 16 up to this many launches
    uint32_t *h_flag = nullptr;                      // RING words of pinned host memory
    hipEvent_t ev[RING] = {};
    struct Pending {
        unsigned long issued_at;                     // index of the eligible launch this fast launch was
        int slot;
        bool probe;                                  // the single fast launch at the end of a back-off
    };
    std::vector<Pending> pending;                    // oldest first; at most LAG entries
    unsigned long elig = 0;                          // eligible launches so far (fast or not)
    unsigned long seq = 0;                           // fast launches that were given a word
    int state = ON_FAST;
    int backoff = 0, backoff_len = 0;                // plain launches left; length of the current back-off
    unsigned long backoff_launches = 0, bad_words = 0;
};

struct lumahip_copy_pool;                                   // lumahip_host.hip: worker threads of the staging copies
void lumahip_copy_pool_destroy(lumahip_copy_pool *p);

struct lumahip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    int num_cu = 256;

    bool have_quant = false;
    int ptf = 0;
    unsigned bitdepth = 0, bitdepthC = 0;
    lh::QuantDev q{};
    std::shared_ptr<const lh::ThreshIndex> tix;  // encode-side search index, built on first use (ensure_search_index)
    std::shared_ptr<const lh::LinIndex> lix;     // value-keyed records, when the float-bit ones miss LDS and these fit (PTF_LINEAR)
    bool use_lin_index = true;                   // lumahip_tune("lin_index", 0): never (A/B, tests)
    bool index_ready = false;
    // YCbCr only: records of the composite luma -> code function (encode), the per-stream y table (decode); host_lut.cpp
    std::shared_ptr<const lh::ThreshIndex> tix_y;
    uint32_t *d_rec_y = nullptr;
    lh::QuantDev q_y{};           // q with the composite records in place of the luminance records
    float *d_ytab = nullptr;
    bool use_ycbcr_tables = true; // lumahip_tune("ycbcr_tables", 0): per-pixel PQ evaluation as in round 2 (A/B, tests)
    // YCbCr encode, binary16 inputs: device copies of the half-input table (luma_device.hpp half_lookup), one per (sc, Lmax) seen;
    // d == nullptr records "not usable for this pair" (host_lut.cpp ycbcr_half_table_host).  lumahip_core.hip half_table_for
    struct HalfTab {
        float sc = 0.0f, Lmax = 0.0f;
        float *d = nullptr;
        unsigned long last_use = 0;
    };
    std::vector<HalfTab> half_tabs;
    unsigned long half_clock = 0;
    int half_mode = 1;            // lumahip_tune("half_table"): 0 = never, 1 = while the stream looks like binary16 data (half_pol: LagPolicy), 2 = always
    LagPolicy half_pol{true, 1024};     // which kernel an eligible YCbCr encode launch takes (half_mode 1): a report = "these are not halves"
    unsigned long half_launches = 0;
    // YCbCr decode: device copies of the per-stream red / blue tables (lumahip_decode.hip rb_table_for), one per preScaling seen
    struct RbTab {
        float sc = 0.0f;
        float *d = nullptr;
        unsigned long last_use = 0;
    };
    std::vector<RbTab> rb_tabs;
    unsigned long rb_clock = 0, rb_launches = 0;
    bool test_fail_rb_alloc = false;
    bool rb_unavailable = false;  // allocating or building the tables failed for this stream: plain kernels, no retry per launch
    LagPolicy rb_pol{false, 64};      // which kernel an eligible YCbCr decode launch takes (rb_mode 1): NO report = "no wave found its codes local"
    int rb_mode = 1;              // lumahip_tune("ycbcr_rb_tables"): 0 = six powf per pixel (rounds 3-4), 1 = red / blue from the tables where a
                                  // wave's codes are close to each other (luma_kernels.hpp rb_wave_near), 2 = from the tables always
    int dec_vw = 0;               // lumahip_tune("dec_vw", 2): the decode kernels with two pixels per thread and row (0: four where alignment allows)
    int rb_near_y = 64, rb_near_c = 24;    // lumahip_tune("rb_near_y" / "rb_near_c"): the closeness bounds of mode 1 (luma_kernels.hpp rb_wave_local)
    bool force_literal = false;   // lumahip_tune("force_literal"): the reference's bisection instead of the records
    std::vector<float> h_lut;     // host copy of the table handed to lumahip_set_quantizer
    float *d_lut = nullptr;
    uint32_t *d_rec = nullptr;
    bool lut_in_lds = true;  // decode side: tables up to 12 bits are staged in LDS
    float minLum = 0.0f;

    // staging for the _host entry points
    float *d_frame = nullptr;
    size_t d_frame_cap = 0;
    unsigned char *d_planes = nullptr;
    size_t d_planes_cap = 0;
    float *d_stats = nullptr;
    float *d_stats_part = nullptr;   // partial statistics triples of the encode kernels (STATS_SLOTS per frame)
    size_t d_stats_part_cap = 0;
    float *d_arr = nullptr;
    size_t d_arr_cap = 0;

    // 3-slot pipeline of the batched host entry points (H2D / kernel / D2H on three streams)
    struct Slot {
        float *d_frame = nullptr;
        unsigned char *d_planes = nullptr;
        float *d_stats = nullptr;
        hipEvent_t h2d = nullptr, kern = nullptr, d2h = nullptr;
    } slot[3];
    size_t slot_frame_cap = 0, slot_planes_cap = 0;
    hipStream_t s_h2d = nullptr, s_kern = nullptr, s_d2h = nullptr;
    float *h_stats = nullptr;  // pinned, 3 floats per frame
    // row bands of the single-frame host entry points (H2D of band k+1 | kernel of band k | D2H of band k-1)
    static constexpr int MAX_BANDS = 8;
    hipEvent_t band_h2d[MAX_BANDS] = {}, band_kern[MAX_BANDS] = {};
    float *d_band_stats = nullptr;  // 3 floats per band
    int host_bands = 4;             // lumahip_tune("host_bands"): 1 = the whole frame in one piece
    int band_taper = 70;            // lumahip_tune("band_taper"): each band's rows in % of the previous band's (100 = uniform)
    int up_ramp = 0;                // staged uploads: how many of the small leading chunks of this call have been used
    size_t h_stats_cap = 0;

    // Pinned staging for pageable caller memory (see xfer_h2d): two chunks per direction, ping-pong
    struct Stage {
        unsigned char *h = nullptr;
        hipEvent_t ev = nullptr;
        bool pending = false;  // a DMA that reads / writes this chunk may still be in flight
        // device -> host only: what to do with the chunk once its DMA has landed (copy it out to the caller's pageable memory)
        unsigned char *out = nullptr;
        size_t out_pitch = 0, chunk_pitch = 0, width = 0, rows = 0;
        unsigned tag = 0;      // which pushed frame the chunk belongs to (lumahip_encode_stream_push); 0 outside a stream
    };
    static constexpr int N_STAGE = 4;   // upload chunks: up to four DMAs queued while the CPU fills the next
    static constexpr int N_STAGE_DN = 8;  // download chunks: a 4K frame's planes are five of them
    Stage stage_up[N_STAGE], stage_dn[N_STAGE_DN];
    unsigned up_next = 0, dn_next = 0;  // ring positions
    size_t dn_chunk = (size_t)8 << 20;  // bytes per download chunk (grows with the frames of the pipelined encode paths)
    // frames pushed with lumahip_encode_stream_push and not yet popped: sequence numbers [es_tail, es_head), frame j in slot[j % 3]
    unsigned es_head = 0, es_tail = 0;
    int es_dir = 0;                // 0: the frames in flight were pushed by lumahip_encode_stream_push, 1: by lumahip_decode_stream_push
    unsigned es_w = 0, es_h = 0;
    int es_profile = 0;
    float es_sc = 1.0f;
    size_t es_total = 0;           // plane bytes and strides of the frames in flight (a push with other strides is refused)
    int es_stride[3] = {0, 0, 0};
    float *h_es_stats = nullptr;   // pinned, 3 floats per slot
    unsigned d2h_tag = 0;          // tag given to download chunks queued now
    float *h_small = nullptr;  // pinned scratch for the few-float readbacks
    lumahip_copy_pool *copy_pool = nullptr;
    // NUMA placement of the host side (numa_host.cpp; lumahip_core.hip numa_resolve): the node of this context's GPU and that
    // node's CPUs.  The pinned staging rings are allocated on the node and the copy threads are pinned to its CPUs.
    // lumahip_tune("numa", v): 0 = no placement, as before round 4; 2 (default) = the rings on the GPU's node, the threads left
    // to the scheduler; 1 = rings and threads; 3 = threads only.  Why not 1 by default: on the shared hosts of the GPU boxes
    // (load average 16-33) pinned copy threads cannot move away from cores other tenants keep busy, and the one-frame and
    // pipelined facade calls then dip by up to 30 % every few runs (profiles/r04_numa.txt, second table)
    int numa_mode = 2;
    int numa_force_node = -1;  // lumahip_tune("numa_node", N): pretend the GPU sits on node N (A/B measurements: local against remote)
    bool numa_resolved = false;
    int numa_node = -1;        // -1: not a NUMA box / unknown / switched off
    std::vector<int> numa_cpus;
    // Half upload of the host encode entry points (lumahip_host.hip xfer_h2d_f16): host frames that hold binary16 values cross
    // PCIe as halves.  lumahip_tune("half_upload", v): 0 never, 1 (default) while the frames do hold halves, 2 always try.
    int in16_mode = 1;
    int in16_backoff = 0, in16_backoff_len = 0;
    unsigned long in16_frames = 0, in16_fallbacks = 0;   // frames (or row bands) uploaded as halves / found to hold other values
    bool slot_in16[3] = {false, false, false};           // stream push / pop: what the slot's device frame holds
    int copy_threads = 5;      // lumahip_tune("copy_threads"): worker threads of the staging copies (0 = caller only); 3 until the half
                               // upload, whose conversion is worth two more (batched half-valued 4K frames 4.9 -> 5.6 Gpixel/s; floats: no change)
    int copy_spin = 2000;      // lumahip_tune("copy_spin"): polls of an idle worker before it sleeps

    int block_threads = 256;
    bool block_forced = false;
    bool allow_alias = false;  // LUMAHIP_ALLOW_ALIASED_FRAMES=1: measurement tools alias all frames of a batch onto one
    int blocks_per_cu = 0;  // 0 = occupancy query
    long grid_override[2] = {0, 0};
    size_t lds_table_max = LUMAHIP_LDS_TABLE_MAX_DEFAULT;
    // Unordered sections (lumahip_begin_unordered): successive _device encode / decode calls go round-robin to `lanes_active`
    // internal streams, so that one batch's ramp-up and tail overlap its neighbours' steady state
    hipStream_t lane_stream[LUMAHIP_MAX_LANES] = {};
    hipEvent_t lane_done[LUMAHIP_MAX_LANES] = {};
    hipEvent_t lane_fork = nullptr;
    int lanes_active = 0;
    unsigned lane_next = 0;
    long lane_grid[2] = {0, 0};   // lumahip_tune("lane_grid_enc" / "lane_grid_dec"): workgroups per launch inside a section
    int lanes_default = 0;        // lumahip_tune("lanes"): lanes an unordered section opens with when asked for 0
};

int lumahip_fail(lumahip_ctx *c, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
#define fail lumahip_fail

#define HIPCHK(c, expr)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail((c), LUMAHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                             \
    } while (0)


namespace lhost {
using namespace lh;

static inline bool is_aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

// a pair of timing events that cannot leak on an early return
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create()
    {
        hipError_t e = hipEventCreate(&e0);
        return e != hipSuccess ? e : hipEventCreate(&e1);
    }
    ~EventPair()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};


struct DisplayParams {
    unsigned char *rgba = nullptr;
    int stride = 0;
    size_t frame_stride = 0;
    float exposure = 1.0f, gamma = 2.2f;
    int do_tmo = 0, ldr_sim = 0;
};

// ---- lumahip_launch.hip
size_t lds_bytes(const lumahip_ctx *c, bool encode_side, int cs_eff, bool ycode = false, bool half = false);   // ycode: the composite-record encode kernels; half: + the half-input table
int block_threads_for(const lumahip_ctx *c, size_t lds, bool few_waves = false, bool valu_bound = false);
int grid_for(const lumahip_ctx *c, int threads, int total_tiles, int dir, int few_writers = 0, int ycbcr = 0);   // few_writers: 0 no, 1 yes, 2 yes with the colour planes in separate buffers; ycbcr: 0 no, 1 yes, 2 the half-input encode kernels
// ---- lumahip_core.hip
int ensure_search_index(lumahip_ctx *c);   // every encode-side launch calls this first (lazy build / process-wide cache)
bool ycbcr_composite_ready(const lumahip_ctx *c);   // encode: the composite luma -> code records exist and fit LDS
int half_table_for(lumahip_ctx *c, float sc, const float **tab);   // *tab = the device half-input table of (sc, the quantizer's Lmax), or nullptr: none
// this eligible launch: the data-dependent ("fast") kernel (true) or the plain one.  On true, *flag is the launch's feedback
// word (nullptr when the feedback ring could not be set up: the policy then always answers true) and the caller calls
// lag_policy_launched(c, p, stream) right behind the kernel launch.
// lag_policy_next may BLOCK the calling thread: it reads the word of the fast launch issued LAG eligible launches earlier after
// waiting for that launch's event (hipEventSynchronize, not a poll: what the policy decides must be a function of the data, never
// of how far the device has got).  At most LAG - 1 eligible launches of a context are therefore queued behind the one being
// waited for -- 3 x 0.4 ms of work at 20 4K frames per launch, 3 x 21 us at one frame against a 6 us launch cost: the device
// does not run dry; include/lumahip.h says so at the *_device entry points.
bool lag_policy_next(LagPolicy &p, uint32_t **flag);
int lag_policy_launched(lumahip_ctx *c, LagPolicy &p, hipStream_t s);
void lag_policy_cancel(LagPolicy &p);   // the launch lag_policy_next handed a word to did not happen: forget its pending entry
struct LagLaunchGuard {                 // cancels on every return between lag_policy_next and the kernel launch
    LagPolicy &p;
    uint32_t *flag;
    bool launched = false;
    ~LagLaunchGuard()
    {
        if (flag && !launched)
            lag_policy_cancel(p);
    }
};
void lag_policy_reset(LagPolicy &p);      // a new stream: waits for the launches in flight, clears their words, state ON_FAST
void lag_policy_destroy(LagPolicy &p);
void numa_resolve(lumahip_ctx *c);                                 // fills numa_node / numa_cpus once (cheap afterwards)
int check_geom(lumahip_ctx *c, unsigned w, unsigned h, int profile, int cs_eff);
bool make_geom(FrameGeom &g, unsigned w, unsigned h, int vw, int nw, unsigned nframes);
hipStream_t launch_stream(lumahip_ctx *c, bool lanes);   // the context's stream, or -- for the entry points that take part in unordered sections -- the next lane of an open one
void plane_dims(unsigned w, unsigned h, int profile, int p, int &rows, int &row_bytes);
// rgb: the three colour-plane base pointers of the float frames (nullptr: no float frames in this call)
int check_layout(lumahip_ctx *c, unsigned w, unsigned h, int profile, unsigned nframes, const float *const rgb[3],
                 size_t frame_stride, const int stride[3], const size_t pfs[3]);

// ---- lumahip_encode.hip / lumahip_decode.hip: cs_eff = the colour space the kernels run (the context's, or CS_PACK /
// CS_RGB for the pack-only entry points)
// rgb[c]: base of colour plane c; plane c of frame f at rgb[c] + f*frame_stride floats
int encode_frames_device_impl(lumahip_ctx *c, const float *const rgb[3], size_t frame_stride, unsigned nframes,
                              unsigned w, unsigned h, float sc, int profile, unsigned char *const planes[3],
                              const int stride[3], const size_t pfs[3], float *stats, int cs_eff, bool lanes = false, bool in16 = false);
// in16: rgb[] point at binary16 planes (the half upload of the host entry points, lumahip_host.hip); same element offsets and
// strides; LUMAHIP_ERR_UNSUPPORTED unless encode_supports_in16() (records in LDS, rows of a multiple of 4 pixels)
bool encode_supports_in16(lumahip_ctx *c, unsigned w);
int decode_impl(lumahip_ctx *c, const unsigned char *const planes[3], const int stride[3], const size_t pfs[3],
                unsigned nframes, unsigned w, unsigned h, int profile, float sc, float *const rgb[3], size_t frame_stride,
                const DisplayParams &dp, int cs_eff, bool lanes = false, float *const rot[3] = nullptr);
// rot: PACKED frames rotating over three buffers, frame f at rot[f % 3] + (f / 3) * frame_stride (rgb is then ignored)
// lanes: the call is one of the four _device encode / decode entry points and goes to a lane of an open unordered section;
// every other caller (the _host entry points with their own upload / kernel / download streams, the stream push / pop, the
// display decode) stays on c->stream whether a section is open or not, as include/lumahip.h promises
int array_launch(lumahip_ctx *c, const float *d_in, float *d_out, size_t n, unsigned ch, bool quant);

// ---- lumahip_misc.hip
int seq_mean(lumahip_ctx *c, const float *chan0_dev, unsigned w, unsigned h, float *mean_host);
int mean_luminance_reference_impl(lumahip_ctx *c, const float *rgb_dev, unsigned w, unsigned h, float sc, int cs_eff,
                                  float *mean_host, bool in16 = false);   // in16: the frame at rgb_dev holds binary16 values

// ---- lumahip_host.hip
int xfer_h2d(lumahip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s);
int xfer_d2h(lumahip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s);
int read_small(lumahip_ctx *c, float *dst, const float *src_dev, int n, hipStream_t s);
int ensure(lumahip_ctx *c, void **p, size_t *cap, size_t need);

}  // namespace lhost
