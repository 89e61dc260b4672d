/*
 * lumahip.h -- C ABI of the MI355X-native Luma HDRv quantize / dequantize hot path.
 *
 * This is the drop-in boundary.  The reference (gabrieleilertsen/lumahdrv v1.0.0) has no FFI layer: its
 * boundary is the C++ class API LumaQuantizer / LumaEncoder::encode(LumaFrame*) / LumaDecoder::decode().
 * The C++ facade in include/luma/ keeps that class surface and is implemented on top of the functions
 * below; every function names the reference interface it replaces (paths relative to the reference
 * tree).  Plain pointers and sizes only -- no HIP, torch or C++ types cross this boundary.
 *
 * Conventions
 *   - every function returns LUMAHIP_OK (0) or an error code; nothing throws across the boundary;
 *     lumahip_last_error(ctx) returns a human-readable message for the last failure on that context;
 *   - frames are the reference's LumaFrame layout (include/luma/luma_frame.h:51-90): planar fp32,
 *     channel c at base + c*h*w, rows of w floats, no padding;
 *   - coded planes are the three Y/U/V planes of a vpx_image_t as the reference fills and reads them
 *     (src/luma_encoder.cpp:260-317, src/luma_decoder.cpp:205-240): planes[p] + y*stride[p] + x*bps,
 *     16-bit samples little-endian; `profile` is the VP9 profile that selects the layout exactly as
 *     src/luma_encoder.cpp:121-128 does: 0 = 4:2:0 8-bit, 1 = 4:4:4 8-bit, 2 = 4:2:0 16-bit,
 *     3 = 4:4:4 16-bit;
 *   - width and height must be even and non-zero (src/luma_encoder.cpp:118-119);
 *   - a context is bound to one GPU and one HIP stream and is not thread-safe; distinct contexts are
 *     independent.  "_host" entry points take host pointers and return when the results are in host
 *     memory; "_device" entry points take device pointers, enqueue on the context's stream and return
 *     immediately (lumahip_sync waits).
 *   - there is no CPU fallback: without a HIP device every entry point that takes a context fails with LUMAHIP_ERR_HIP.  (The
 *     functions marked "host-only" build tables or evaluate ONE value on the host, as the reference does, and take no context.)
 */
#ifndef LUMAHIP_H
#define LUMAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  The BINARY interface has only ever grown: every symbol of an earlier version is still exported with the same
 * signature, so a program linked against version 2 runs against this library.
 *   5 (round 6): additions -- lumahip_pool_create_small, lumahip_decoded_ring_* / lumahip_decode_frames_device_ring.
 *   4 (round 5): additions -- lumahip_rb_table_info, lumahip_lin_index_host, LUMAHIP_POOL_ROTATING, the lumahip_tune keys
 *     "ycbcr_rb_tables" / "rb_near_y" / "rb_near_c" / "lin_index"; lumahip_quantizer_info may answer search mode 7.  Behaviour:
 *     the half-input table's kernel choice is now a function of the stream's data only (feedback read four eligible launches
 *     later, after the launch's completion event) -- launch counts of lumahip_half_table_info on a float stream differ from
 *     version 3's, results never did; YCbCr decode launches may read the red / blue tables (same results).
 *   3 (round 4): additions (value_host scalar forms, half-input table, NUMA, multi transport).  SOURCE-level break: about ten
 *     declarations that version 2's header showed unconditionally are test / measurement hooks and moved into the
 *     `#ifdef LUMAHIP_EXPERIMENTAL` section at the end (lumahip_thresh_index_host, lumahip_ycbcr_*_host,
 *     lumahip_synth_frames_device, lumahip_time_launches, the traffic probes, ...): C / C++ callers that use them define
 *     LUMAHIP_EXPERIMENTAL before including this header; the symbols themselves are unchanged.  Behaviour changes of version 3:
 *     lumahip_time_launches returns LUMAHIP_ERR_STATE inside an unordered section, the default of lumahip_tune("copy_threads")
 *     went from 3 to 5, and lumahip_multi_* with every shard on one device takes the table from the host instead of RCCL
 *     (lumahip_multi_set_transport(m, 1) restores the broadcast). */
#define LUMAHIP_ABI_VERSION 5

enum lumahip_status {
    LUMAHIP_OK = 0,
    LUMAHIP_ERR_ARG = 1,         /* bad argument (odd size, null pointer, unknown enum, ...) */
    LUMAHIP_ERR_HIP = 2,         /* a HIP runtime call failed / no device */
    LUMAHIP_ERR_STATE = 3,       /* quantizer not set */
    LUMAHIP_ERR_UNSUPPORTED = 4  /* e.g. unknown colour space: the reference's transformColorSpace()==false */
};

/* numeric values are serialised by the reference into MKV attachments 432 / 433
 * (include/luma/luma_quantizer.h:95-96, src/luma_encoder.cpp:86-92) */
enum lumahip_ptf { LUMAHIP_PTF_PSI = 0, LUMAHIP_PTF_PQ = 1, LUMAHIP_PTF_LOG = 2, LUMAHIP_PTF_JND_HDRVDP = 3,
                   LUMAHIP_PTF_LINEAR = 4 };
enum lumahip_colorspace { LUMAHIP_CS_LUV = 0, LUMAHIP_CS_RGB = 1, LUMAHIP_CS_YCBCR = 2, LUMAHIP_CS_XYZ = 3 };

typedef struct lumahip_ctx lumahip_ctx;

/* ---- life cycle -------------------------------------------------------------------------------- */

int lumahip_abi_version(void);
int lumahip_device_count(int *count);
/* device = HIP ordinal, or -1 for the calling thread's current device */
int lumahip_create(lumahip_ctx **out, int device);
void lumahip_destroy(lumahip_ctx *ctx);
const char *lumahip_last_error(const lumahip_ctx *ctx);
int lumahip_device(const lumahip_ctx *ctx);   /* the HIP ordinal the context is bound to */
/* Run on a caller-owned hipStream_t (e.g. PyTorch's current stream) instead of the context's own
 * non-blocking stream.  NULL is a valid handle and means the device's default (null) stream -- which is what
 * PyTorch's default stream is.  lumahip_reset_stream goes back to the context's own stream. */
int lumahip_set_stream(lumahip_ctx *ctx, void *hip_stream);
int lumahip_reset_stream(lumahip_ctx *ctx);
int lumahip_sync(lumahip_ctx *ctx);

/* Measurement overrides, none of which changes a result.  Keys: "block" (threads per workgroup: 64..1024, 0 = rule),
 * "blocks_per_cu" (persistent workgroups per CU, 0 = rule), "grid_enc" / "grid_dec" (absolute workgroup count of the
 * encode / decode launches, 0 = rule), "lds_table_max_kb" (largest search table staged in LDS, default 144, -1 = default),
 * "force_literal" (1: the reference's bisection, src/luma_quantizer.cpp:222-235, run literally instead of the threshold
 * records), "allow_aliased_frames" (1: the layout check accepts batches whose frames overlap), "lanes" (default lane
 * count of lumahip_begin_unordered), "lane_grid_enc" / "lane_grid_dec" (workgroups per launch inside an unordered section,
 * 0 = rule), "copy_threads" (worker threads that copy pageable caller memory into the pinned staging chunks of the _host
 * entry points: 0..32, default 5), "host_bands" (row bands the single-frame _host entry points split a frame into so that
 * the upload of band k+1, the kernel of band k and the download of band k-1 overlap: 1..8, default 4), "band_taper" (each band's rows in
 * per cent of the previous band's, 10..100, default 70: the last band is small, so little is left to do once the upload ends), "ycbcr_tables" (0: the
 * YCbCr kernels evaluate every PQ function per pixel instead of taking the luminance code / the luma from per-stream tables),
 * "half_table" (0 / 1 / 2: when the YCbCr encode kernels use the half-input table, see lumahip_ycbcr_half_table_host),
 * "numa" / "numa_node" (NUMA placement of the staging rings and copy threads, see lumahip_numa_info),
 * "half_upload" (0 / 1 / 2: whether host frames that hold binary16 values cross PCIe as halves, see lumahip_half_upload_info).  The environment variables LUMAHIP_BLOCK, LUMAHIP_BLOCKS_PER_CU, LUMAHIP_GRID_ENC,
 * LUMAHIP_GRID_DEC, LUMAHIP_LDS_TABLE_MAX_KB, LUMAHIP_FORCE_LITERAL, LUMAHIP_ALLOW_ALIASED_FRAMES, LUMAHIP_LANES, LUMAHIP_LANE_GRID_ENC,
 * LUMAHIP_LANE_GRID_DEC, LUMAHIP_COPY_THREADS, LUMAHIP_HOST_BANDS, LUMAHIP_BAND_TAPER, LUMAHIP_YCBCR_TABLES, LUMAHIP_HALF_TABLE set the
 * same keys when a context is created, but only if LUMAHIP_TUNING=1 is set as well. */
int lumahip_tune(lumahip_ctx *ctx, const char *key, long value);

/* ---- quantizer --------------------------------------------------------------------------------- */

/* Replaces LumaQuantizer::setQuantizer (src/luma_quantizer.cpp:172-212) for the device side.  The host
 * facade builds the LUT exactly as the reference does (libm powf/log10f or the PSI/HDR-VDP tables) and,
 * on the decoder, overwrites its first getSize() entries with MKV attachment 434
 * (src/luma_decoder.cpp:121-122); the FINAL table of 2^bitdepth floats is handed over here.
 * bitdepth 1..16, bitdepthC 1..16.  The call uploads the table; the encode-side search index is built (or taken from a
 * process-wide cache keyed by the table) by the first encode-side call, so a context that only decodes never builds it. */
int lumahip_set_quantizer(lumahip_ctx *ctx, int ptf, unsigned bitdepth, int colorspace, unsigned bitdepthC,
                          float maxLum, float minLum, const float *lut_host, size_t lut_len);

/* Host-only helper: builds the 2^bitdepth-entry table exactly as LumaQuantizer::setQuantizer does
 * (src/luma_quantizer.cpp:114-169,172-212) -- host libm powf / log10f for PQ / LOG, Lmax*i/maxVal for
 * LINEAR, the captured PSI / JND-HDR-VDP data tables (lumahdrv_amd/data, or $LUMAHIP_DATA_DIR) otherwise.
 * Needs no GPU and no context.  LUMAHIP_ERR_UNSUPPORTED for PSI/HDR-VDP deeper than 12 bits (the
 * reference reads out of bounds there), LUMAHIP_ERR_STATE if a data table cannot be read. */
int lumahip_build_lut(int ptf, unsigned bitdepth, float maxLum, float minLum, float *lut_out, size_t lut_len);


/* Half-input table of the YCbCr encode kernels.  The reference's EXR reader hands the encoder binary16 values widened to float
 * (src/exr_interface.cpp:77-146 reads Imf::Rgba), and for such an input x the non-linear colour value
 * PQenc(std::max(x * sc, 1e-10f)) (src/luma_quantizer.cpp:331-333, 491-494) depends on x's 16 bits only: per (preScaling sc,
 * maxLum) the library tabulates it with the host libm for the 31745 halves +0 ... +inf (out[i], i = the half's bit pattern;
 * negative halves share entry 0) and the kernels replace the six powf of a pixel by three LDS gathers.  Pixels whose inputs
 * are not binary16 values (or NaN) are evaluated per pixel as before, inside the same launch; results are identical either
 * way.  lumahip_ycbcr_half_table_host is host-only (no GPU, no context; cap >= 31745 floats) and returns
 * LUMAHIP_ERR_UNSUPPORTED when the table is not usable for the pair (sc not positive and finite, or an entry outside
 * [7e-7, 2] that is not a NaN, maxLum outside [1e-6, 1e9]) -- the kernels then evaluate every pixel.  lumahip_half_table_info reports for the context's
 * quantizer and a preScaling: info[0] = 1 when encode launches without per-frame statistics are eligible for the half-input
 * kernels, info[1] = their LDS bytes per workgroup, info[2] = device copies of the table the context holds (one per (sc,
 * maxLum) seen, at most four), info[3] = table entries, info[4] = launches that took the half-input kernels so far, info[5] =
 * eligible launches that ran the per-pixel kernels instead because the stream did not look like binary16 data.
 * lumahip_tune("half_table", v): 0 = never use the table; 1 (default) = use it while the stream looks like binary16 data -- a
 * launch most of whose pixels are full-precision floats costs 1.4 x the per-pixel kernels' time on the table kernels, so every
 * table launch reports whether it was one (its own word of pinned host memory, read after the launch's completion event when
 * the fourth eligible launch after it is issued: the choice of kernel is a function of the stream's data, never of timing);
 * a report sends the following 16 eligible launches to the per-pixel kernels, then ONE launch tries the table again, and the
 * pause doubles (up to 1024 launches) for every probe that reports again; a clean probe returns to the table.
 * 2 = always use it.  lumahip_set_quantizer and this key start the policy afresh.  None of this changes a result. */
int lumahip_ycbcr_half_table_host(float sc, float maxLum, float *out, size_t cap);
int lumahip_half_table_info(lumahip_ctx *ctx, float sc, int info[6]);

/* Red / blue tables of the YCbCr decode kernels.  A decoded pixel's red depends on its luminance code and its Cr code only, its
 * blue on the luminance code and the Cb code (src/luma_quantizer.cpp:447-451, 460-468: y + chroma term, clamp, PQdec, / sc), so per
 * stream and preScaling the library tabulates both on the device (2 x 2^(bitdepth + bitdepthC) floats: 8 MiB for the HDR10
 * recipe, 128 MiB for 12-bit luminance and colour; not beyond 256 MiB; built by one launch with the same complete functions the
 * kernels fall back to)
 * and the kernels replace four of a pixel's six powf by two 4-byte gathers from global memory (L1 / L2) -- where the gathers are
 * cheap: a wave first checks that most of its lanes' codes are close to their neighbours' (a picture), else it
 * computes all three channels as before.  Results are identical either way.  lumahip_tune("ycbcr_rb_tables", v): 0 = never,
 * 1 (default) = as described, and launches none of whose waves found its codes local send the following 16 eligible launches to
 * the kernels without the test, then one launch probes again (the same data-driven policy as "half_table": per-launch feedback
 * words read after the launch's completion event; the pause doubles up to 64 launches here), 2 = every wave takes the tables.
 * lumahip_tune("rb_near_y" / "rb_near_c", n) = the closeness bounds of mode 1 in luminance / colour codes (64 / 24).
 * lumahip_rb_table_info: info[0] = 1 when decode launches of this context and preScaling are eligible, info[1] = bytes of the two
 * tables, info[2] = launches that took the kernels with the tables so far, info[3] = eligible launches the policy sent to the
 * plain kernels instead. */
int lumahip_rb_table_info(lumahip_ctx *ctx, float sc, int info[4]);

/* Host-only (no GPU, no context): LumaQuantizer::quantize / dequantize for ONE value (src/luma_quantizer.cpp:215-264) on a
 * table of lut_len = 2^bitdepth floats -- the reference's literal bisection + nearest-of-two for channel 0 (and for every
 * channel of the RGB / XYZ colour spaces), clamp(floor(maxC*val + 0.5f), 0, maxC) resp. std::max(val/maxC, 1e-10f) for the
 * colour channels, maxC = 2^bitdepthC - 1.  The reference's per-sample members cost ~50 ns and callers written against it may
 * loop over them; the facade's LumaQuantizer::quantize / dequantize call these instead of launching a kernel per sample.
 * Frames, planes and arrays are never processed this way: those entry points run the kernels. */
int lumahip_quantize_value_host(const float *lut, size_t lut_len, int colorspace, unsigned bitdepthC, float val, unsigned ch,
                                float *out);
int lumahip_dequantize_value_host(const float *lut, size_t lut_len, int colorspace, unsigned bitdepthC, float val, unsigned ch,
                                  float *out);

/* introspection of the search index built for the current LUT (tests):
 * info[0] = mode (0 = literal bisection, table in LDS; 2 = literal bisection, table read from global memory
 *                 (bitdepth > 12); 3 = threshold records in LDS; 4 = threshold records in global memory;
 *                 7 = value-keyed records in LDS: evenly spaced tables such as PTF_LINEAR, whose float-bit records
 *                 would miss LDS -- lumahip_tune("lin_index", 0) keeps mode 4 for them),
 * info[1] = mantissa bits of the record key (0 for mode 7), info[2] = number of records, info[3] = key shift (0 for mode 7),
 * info[4] = LDS bytes per workgroup of the encode-side kernels */
int lumahip_quantizer_info(const lumahip_ctx *ctx, int info[5]);

/* ---- host entry points (drop-in: H2D, kernel, D2H, synchronous) -------------------------------- */

/* Replaces LumaEncoder::encode(LumaFrame*) minus run(), i.e. transformColorSpace(frame,true,sc) +
 * setChannels(frame) (include/luma/luma_encoder.h:142-148, src/luma_encoder.cpp:196-201,260-317), as ONE
 * fused kernel.  `rgb` is not modified.  If `transformed_out` is non-NULL it receives the 3*w*h floats
 * the reference leaves in the caller's frame (its in-place side effect).
 * `mean_lum` (nullable) receives the plane-0 average the reference computes for its
 * "mean luminance <= 1" warning (src/luma_encoder.cpp:313-316). */
int lumahip_encode_frame_host(lumahip_ctx *ctx, const float *rgb, unsigned w, unsigned h, float sc, int profile,
                              unsigned char *const planes[3], const int stride[3], float *mean_lum,
                              float *transformed_out);

/* Replaces LumaDecoder::decode() minus run(), i.e. getVpxChannels() + transformColorSpace(frame,false,sc)
 * (include/luma/luma_decoder.h:143-161, src/luma_decoder.cpp:205-240) as one fused kernel. */
int lumahip_decode_frame_host(lumahip_ctx *ctx, const unsigned char *const planes[3], const int stride[3],
                              unsigned w, unsigned h, int profile, float sc, float *rgb_out);

/* Batched forms of the two calls above for callers that hold several frames (a transcoder, a batch job):
 * rgb[i] / rgb_out[i] are nframes host frames, planes[3*i + p] the planes of frame i.  Internally a 3-slot pipeline
 * over three HIP streams: frame i's H2D copy overlaps frame i-1's kernel and frame i-2's D2H copy (fully so when
 * the caller's buffers are pinned, lumahip_host_register).  Results are identical to nframes single calls.
 * mean_lum (nullable) receives nframes values. */
int lumahip_encode_frames_host(lumahip_ctx *ctx, const float *const *rgb, unsigned nframes, unsigned w, unsigned h,
                               float sc, int profile, unsigned char *const *planes, const int stride[3],
                               float *mean_lum);
int lumahip_decode_frames_host(lumahip_ctx *ctx, const unsigned char *const *planes, const int stride[3],
                               unsigned nframes, unsigned w, unsigned h, int profile, float sc, float *const *rgb_out);

/* Streaming form of lumahip_encode_frames_host for callers that get their frames one at a time -- the reference's loop
 * `for (...) encoder.encode(&frame)` (lumaenc.cpp:205-243) -- and accept ONE frame of latency: push(frame i+1), then pop(frame i).
 * A synchronous one-frame call cannot overlap its tail (last kernel, download, copy out of the staging chunks, ~0.5 ms of
 * 2.6 ms for a pageable 4K frame) with anything; here it runs under the next frame's upload.
 * push: uploads `rgb` (LumaFrame layout), queues the fused kernel and the download of the planes into `planes` / `stride`, and
 * returns as soon as `rgb` may be reused; `planes` must stay valid until the frame is popped.  At most two frames in flight
 * (LUMAHIP_ERR_STATE otherwise), same w / h / profile while a frame is in flight.
 * pop: completes the OLDEST pushed frame -- its planes are in the caller's memory when it returns; mean_lum (nullable) as in
 * lumahip_encode_frame_host.  Results are identical to lumahip_encode_frame_host.  The batched entry points refuse to run while
 * frames are pending; everything else may be called in between. */
int lumahip_encode_stream_push(lumahip_ctx *ctx, const float *rgb, unsigned w, unsigned h, float sc, int profile,
                               unsigned char *const planes[3], const int stride[3]);
int lumahip_encode_stream_pop(lumahip_ctx *ctx, float *mean_lum);
int lumahip_encode_stream_pending(const lumahip_ctx *ctx);   /* frames pushed and not yet popped: 0, 1 or 2 */
/* The decode counterpart, for the loop `while ((frame = decoder.decode()))` (lumadec.cpp:112-160): push uploads the planes
 * (they may be reused when it returns), queues the fused kernel and the download into `rgb_out` (LumaFrame layout), which must
 * stay valid until the frame is popped; pop completes the oldest pushed frame.  Same rules as above; encode and decode frames
 * cannot be in flight at the same time on one context. */
int lumahip_decode_stream_push(lumahip_ctx *ctx, const unsigned char *const planes[3], const int stride[3], unsigned w, unsigned h,
                               int profile, float sc, float *rgb_out);
int lumahip_decode_stream_pop(lumahip_ctx *ctx);
int lumahip_decode_stream_pending(const lumahip_ctx *ctx);

/* Replaces LumaEncoder::setChannels(LumaFrame*) on its own (src/luma_encoder.cpp:196-201): quantize + pack a
 * frame that is ALREADY colour-transformed.  And LumaDecoder::getVpxChannels on its own
 * (src/luma_decoder.cpp:205-240): unpack + dequantize without the inverse colour transform. */
int lumahip_pack_frame_host(lumahip_ctx *ctx, const float *transformed, unsigned w, unsigned h, int profile,
                            unsigned char *const planes[3], const int stride[3], float *mean_lum);
int lumahip_unpack_frame_host(lumahip_ctx *ctx, const unsigned char *const planes[3], const int stride[3],
                              unsigned w, unsigned h, int profile, float *dequantized_out);

/* Replaces LumaQuantizer::transformColorSpace(LumaFrame*, bool toCs, float sc)
 * (src/luma_quantizer.cpp:267-482): in place on a host frame.  Returns LUMAHIP_ERR_UNSUPPORTED where the
 * reference returns false. */
int lumahip_transform_color_space_host(lumahip_ctx *ctx, float *frame, unsigned w, unsigned h, int toCs, float sc);

/* LumaQuantizer::quantize / dequantize (src/luma_quantizer.cpp:215-264) over arrays; `ch` as in the
 * reference (0 = LUT channel; 1,2 = colour channels unless the colour space is RGB/XYZ). */
int lumahip_quantize_array_host(lumahip_ctx *ctx, const float *in, float *out, size_t n, unsigned ch);
int lumahip_dequantize_array_host(lumahip_ctx *ctx, const float *in, float *out, size_t n, unsigned ch);

/* ---- device entry points (batched, asynchronous on the context's stream) ----------------------- */
/* "Asynchronous" has one exception.  YCbCr streams choose between two kernels per launch from what earlier launches reported
 * about the data (binary16-valued inputs on the encode side, locality of the codes on the decode side): a call may then WAIT,
 * on the host, for the eligible launch issued four such launches earlier -- never for more recent ones, so up to three launches
 * stay queued and the device does not run dry.  The wait is an event wait, not a poll, so that which kernel runs is a function
 * of the data alone.  lumahip_tune(ctx, "half_table", 0 or 2) / ("ycbcr_rb_tables", 0 or 2) fix the choice and remove
 * the wait; other colour spaces never wait.  The red / blue tables of the YCbCr decode kernels are optional: when their memory
 * (8 MiB for the HDR10 recipe, up to 2 x 128 MiB) cannot be had, the call proceeds on the plain kernels. */

/* nframes frames, frame f at rgb_dev + f*frame_stride floats (LumaFrame layout each); plane p of frame f
 * at planes_dev[p] + f*plane_frame_stride[p] bytes.  stats_dev (nullable) receives per frame
 * {sum, min, max} of transformed channel 0 as 3 floats (zero-initialised by the call).
 * One launch covers all frames. */
int lumahip_encode_frames_device(lumahip_ctx *ctx, const float *rgb_dev, size_t frame_stride, unsigned nframes,
                                 unsigned w, unsigned h, float sc, int profile, unsigned char *const planes_dev[3],
                                 const int stride[3], const size_t plane_frame_stride[3], float *stats_dev);
int lumahip_decode_frames_device(lumahip_ctx *ctx, const unsigned char *const planes_dev[3], const int stride[3],
                                 const size_t plane_frame_stride[3], unsigned nframes, unsigned w, unsigned h,
                                 int profile, float sc, float *rgb_dev, size_t frame_stride);
/* The same two calls for float frames given as three colour-plane base pointers: plane c of frame f at
 * rgb_planes_dev[c] + f*frame_stride floats.  The LumaFrame layout of the calls above (include/luma/luma_frame.h:84-87:
 * channel c at buffer + c*h*w) is the special case rgb_planes_dev[c] = rgb_dev + c*w*h, frame_stride >= 3*w*h.  Any layout in
 * which no two planes overlap is accepted, e.g. all R planes of a batch in one buffer, all G planes in a second, all B planes
 * in a third (frame_stride = w*h).  Why: the decode kernel writes 12 of its 15 bytes per pixel, and on MI355X three write
 * streams in three HBM region groups (lumahip_pool below) run up to 11 % faster than one.  Results are
 * identical to the packed calls. */
int lumahip_encode_frames_device_planar(lumahip_ctx *ctx, const float *const rgb_planes_dev[3], size_t frame_stride,
                                        unsigned nframes, unsigned w, unsigned h, float sc, int profile,
                                        unsigned char *const planes_dev[3], const int stride[3],
                                        const size_t plane_frame_stride[3], float *stats_dev);
int lumahip_decode_frames_device_planar(lumahip_ctx *ctx, const unsigned char *const planes_dev[3], const int stride[3],
                                        const size_t plane_frame_stride[3], unsigned nframes, unsigned w, unsigned h,
                                        int profile, float sc, float *const rgb_planes_dev[3], size_t frame_stride);
/* Decoded frames in the reference's PACKED layout (include/luma/luma_frame.h:84-87 there: channel c of a frame at base + c*w*h)
 * spread over THREE buffers: frame f of the batch at bases[f % 3] + (f / 3) * frame_stride floats (frame_stride >= 3*w*h).  Every
 * frame is a LumaFrame as LumaDecoder::decode() returns it; what changes is where consecutive frames live.  With the three
 * buffers in three HBM region groups (lumahip_pool_alloc(pool, LUMAHIP_POOL_ROTATING, ...) three times) the launch walks its tiles
 * interleaved over the frames, so the workgroups running at any moment write all three groups: one launch alone then reaches
 * what the packed layout otherwise needs two launches in flight for (bench.py decode_packed_layout.frame_rotating).  Same floats
 * as lumahip_decode_frames_device. */
int lumahip_decode_frames_device_rotating(lumahip_ctx *ctx, const unsigned char *const planes_dev[3], const int stride[3],
                                          const size_t plane_frame_stride[3], unsigned nframes, unsigned w, unsigned h, int profile,
                                          float preScaling, float *const bases_dev[3], size_t frame_stride);

/* Unordered section.  Frames -- and therefore batches of frames -- are independent in this path (the quantizer is
 * read-only state, src/luma_quantizer.cpp:215-264,267-482 keep nothing between frames), so a caller with several batches to
 * process need not order them against each other.  Between lumahip_begin_unordered and lumahip_end_unordered the four
 * _device encode / decode entry points above hand successive calls round-robin to `lanes` internal streams (1..4, 0 = the
 * default of 2): one batch's ramp-up and tail overlap its neighbour's steady state (+4 % encode, +6 % decode on 20-frame 4K
 * batches, profiles/r03_layout_lab.txt).  Ordering guarantees: everything enqueued on the context's
 * stream before `begin` happens before every call of the section; everything enqueued after `end` happens after all of
 * them; calls inside the section that went to the same lane run in call order; nothing else is promised, so the calls of
 * one section must not depend on each other's output or write the same memory.  lumahip_sync inside a section waits for
 * its lanes too.  All other entry points keep using the context's stream. */
int lumahip_begin_unordered(lumahip_ctx *ctx, int lanes);
int lumahip_end_unordered(lumahip_ctx *ctx);

/* Decode fused with the display-side transform of the reference's player (the step on the far side of the decode
 * path: src/lumaplay_dequantizer.frag:145-156 -- exposure, optional 8-bit LDR simulation, optional sigmoid tone
 * curve n = sig = 0.8, display gamma) into RGBA8 (4 B/pixel, rows rgba_stride bytes apart, alpha 255).
 * rgb_dev may be NULL when only the display image is wanted (3 B read + 4 B written per pixel).  The reference's
 * shader is not bit-reproducible (GL_LINEAR-filtered LUT texture), so this output is specified to +-1 code. */
int lumahip_decode_display_frames_device(lumahip_ctx *ctx, const unsigned char *const planes_dev[3], const int stride[3],
                                         const size_t plane_frame_stride[3], unsigned nframes, unsigned w, unsigned h,
                                         int profile, float sc, float *rgb_dev_or_null, size_t frame_stride,
                                         unsigned char *rgba_dev, int rgba_stride, size_t rgba_frame_stride,
                                         float exposure, float gamma, int do_tmo, int ldr_sim);
int lumahip_transform_color_space_device(lumahip_ctx *ctx, float *frames_dev, size_t frame_stride, unsigned nframes,
                                         unsigned w, unsigned h, int toCs, float sc);

/* The reference's mean luminance of ONE device-resident frame, bit for bit: transformed channel 0 accumulated into a single
 * fp32 variable in raster order, divided by (float)(w*h) (LumaEncoder::setVpxChannel, src/luma_encoder.cpp:276,294,314; the
 * reference warns when it is <= 1).  Tens of ms at 4K (the sum is sequential by definition); synchronous.  The `mean_lum`
 * of the host entry points and the per-frame `sum` statistic of lumahip_encode_frames_device come from the encode kernel
 * instead: an accurate sum, whereas the reference's drops / rounds small addends once its running sum is large (-0.2 % at
 * 1080p, several % at 4K on wide-range content).  The host entry points call this function themselves when their value
 * lies in [0.25, 4] -- the only range in which the two sums can fall on different sides of the threshold for up to 2^25
 * non-negative values -- and always for larger frames or when channel 0 holds negative values, so the `<= 1` decision they
 * support is always the reference's.  Outside those cases mean_lum is the accurate statistic: correct to ~1e-6, but summed
 * with float atomics, i.e. not reproducible in its last bits from run to run. */
int lumahip_mean_luminance_reference_device(lumahip_ctx *ctx, const float *rgb_dev, unsigned w, unsigned h, float sc,
                                            float *mean_host);

/* array quantize / dequantize on device-resident values (asynchronous on the context's stream) */
int lumahip_quantize_array_device(lumahip_ctx *ctx, const float *in_dev, float *out_dev, size_t n, unsigned ch);
int lumahip_dequantize_array_device(lumahip_ctx *ctx, const float *in_dev, float *out_dev, size_t n, unsigned ch);


/* Pin caller-owned host memory (hipHostRegister) so that the _host entry points DMA it at PCIe rate instead
 * of going through the runtime's pageable staging path.  Optional; unregister before freeing the memory. */
int lumahip_host_register(lumahip_ctx *ctx, void *host_ptr, size_t bytes);
int lumahip_host_unregister(lumahip_ctx *ctx, void *host_ptr);


/* ---- device memory helpers (for hosts without their own allocator, e.g. the C++ facade) -------- */
int lumahip_malloc(lumahip_ctx *ctx, void **dev_ptr, size_t bytes);
int lumahip_free(lumahip_ctx *ctx, void *dev_ptr);
int lumahip_memcpy_h2d(lumahip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int lumahip_memcpy_d2h(lumahip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- Half upload of the host encode entry points -------------------------------------------------------------------------
 * The reference's LumaFrame is float, but the frames its EXR reader produces hold binary16 values widened to float
 * (src/exr_interface.cpp:77-146).  lumahip_encode_frame_host, lumahip_encode_frames_host and lumahip_encode_stream_push -- and so
 * LumaEncoder::encode(LumaFrame *) -- send such a frame over PCIe as halves (6 instead of 12 bytes per pixel): the copy threads
 * convert while they stage and check every value's round trip, the encode kernels widen the halves back exactly, and the planes
 * are those of the float upload bit for bit.  A frame (or row band) that holds anything else goes up as floats as before; after
 * one such frame the next 16 are not tried (the pause doubling up to 1024 while it keeps happening), so a stream of
 * full-precision floats pays nothing.  Needs F16C on the host CPU, rows of a multiple of 4 pixels and the luminance records
 * in LDS (every table up to 13 bits); not used when the caller asks for the transformed float frame back.
 * lumahip_tune("half_upload", 0 | 1 | 2) = never / as described (default) / always try.
 * lumahip_half_upload_info: info = {frames (or row bands) uploaded as halves, frames found to hold other values, frames left in
 * the current pause}. */
int lumahip_half_upload_info(const lumahip_ctx *ctx, long info[3]);

/* ---- NUMA placement of the host side ------------------------------------------------------------------------------------
 * On a multi-socket host the context's pinned staging rings are allocated on the NUMA node of its GPU.  lumahip_tune("numa", v):
 * 0 = nothing (the behaviour before round 4); 2 (default) = the rings; 1 = the rings, and the context's copy threads -- and, in
 * lumahip_multi_*, each shard's own thread -- pinned to that node's CPUs (within the CPUs the process may use); 3 = the threads
 * only.  Pinning is not the default because it only pays where the process owns its cores: measured on a shared 2-socket host,
 * staging on the wrong socket costs 5-10 % and a caller on the other socket gains 11 % on decode from rings + threads at the
 * GPU, but pinned threads that cannot leave cores other jobs keep busy made one call in five up to 30 % slower
 * (profiles/r04_numa.txt).  ("numa_node", N) pretends the GPU sits on node N (A/B measurements).  Both keys take effect for what
 * is allocated / started afterwards.  What this serves replaces a single-threaded loop (lumaenc.cpp:205-243 of the reference),
 * so there is no reference behaviour to keep.
 * lumahip_numa_info: info = {node of the GPU or -1 (one-node host, unknown, switched off), number of that node's CPUs the
 * process may use, the first of them}.  lumahip_numa_pin_current_thread pins the CALLING thread to them (modes 1 and 3; a no-op
 * otherwise) -- for callers that drive one context per thread themselves.  lumahip_numa_plan_host is host-only (no GPU, no
 * context): node and CPUs for a PCI bus id from a sysfs tree (NULL = /sys), optionally restricted to a cpulist such as
 * "0-63,128-191"; *ncpus = 0 when there is nothing to do. */
int lumahip_numa_info(lumahip_ctx *ctx, int info[3]);
int lumahip_numa_pin_current_thread(lumahip_ctx *ctx);
int lumahip_numa_plan_host(const char *sysfs_root, const char *pci_bus_id, const char *allowed_cpulist, int *node, int *cpus, int cap,
                           int *ncpus);

/* ---- HBM chunk pool: WHERE device-resident streams live ------------------------------------------------------------
 * On MI355X device memory falls into a few groups of multi-GiB regions, and a launch runs up to 15 % slower when the
 * stream it reads and the streams it writes meet in one group.  A resident-stream application owns its
 * buffers for a long time, so it can look first: the pool takes device memory in chunks, finds the groups with traffic-only
 * launches (lumahip_probe_encode_traffic_device), keeps n_y chunks of one group for Y planes, n_uv chunks of another for
 * U / V planes, the n_float chunks that run fastest against those for float frames and, for channel-strided decode output,
 * n_striped further chunks from EACH of the first three groups, and gives the rest back to the driver.  Nothing here touches results:
 * the pool decides addresses only.  `ctx` must have a quantizer set and is used for the probes during creation only. */
typedef struct lumahip_pool lumahip_pool;
enum lumahip_pool_kind { LUMAHIP_POOL_FLOAT = 0, LUMAHIP_POOL_Y = 1, LUMAHIP_POOL_UV = 2, LUMAHIP_POOL_STRIPED = 3,
                         LUMAHIP_POOL_ROTATING = 4 /* an allocation mode over the STRIPED chunks, see lumahip_pool_alloc */ };
typedef struct lumahip_pool_config {
    size_t chunk_bytes;      /* 0 = 2 GiB */
    int n_float, n_y, n_uv;  /* chunks wanted of each kind */
    int n_striped;           /* chunks wanted per group for LUMAHIP_POOL_STRIPED (0 = none) */
    size_t keep_free_bytes;  /* device memory left untouched while probing (0 = 6 GiB) */
    int max_chunks;          /* upper bound on the chunks taken for probing (0 = whatever is free) */
    int probe_iters;         /* launches per probe (0 = 2) */
} lumahip_pool_config;
int lumahip_pool_create(lumahip_ctx *ctx, const lumahip_pool_config *cfg, lumahip_pool **out);
/* The SMALL pool: for a caller that keeps a few GB of frames on the device and shares the GPU.  Takes max(8, chunks wanted + 4)
 * chunks for a few tens of milliseconds instead of all free memory for seconds, finds the region groups among them (one retry with
 * twice as many when the first attempt sees a single group), keeps the chunks wanted and returns the rest.  Same object, same
 * calls as above afterwards.  A 40-frame 4K stream in 2 float + 1 Y + 1 U/V chunks (8 GB): 0.73 of the roofline for one launch at
 * a time against 0.67 in plain allocations (profiles/r06_small_pool.txt; bench.py value_small_pool). */
int lumahip_pool_create_small(lumahip_ctx *ctx, int n_float, int n_y, int n_uv, int n_striped, lumahip_pool **out);
void lumahip_pool_destroy(lumahip_pool *pool);   /* frees every chunk, handed out or not */
/* One whole chunk of `kind` (fastest first).  group: -1 = any; for LUMAHIP_POOL_STRIPED the region group (0, 1, 2) the chunk
 * must come from.  LUMAHIP_ERR_STATE when none is left.
 * LUMAHIP_POOL_ROTATING: for PACKED decoded frames (the reference's LumaFrame layout, include/luma/luma_frame.h:84-87, what
 * lumahip_decode_frames_device writes).  One decode launch writes one batch, a batch of packed frames lives in one chunk, i.e. in
 * ONE region group -- 0.69 of the roofline wherever the chunk is; what helps is two launches in flight that write DIFFERENT
 * groups (an unordered section).  Consecutive ROTATING allocations hand out the STRIPED chunks of groups 0, 1, 2, 0, ... (group >=
 * 0 restarts the walk there; an exhausted group is skipped), so "allocate the output buffer of every batch in stream order" is
 * all a caller has to do: 0.69 -> 0.75 (bench.py decode_packed_layout.pool_rotating). */
int lumahip_pool_alloc(lumahip_pool *pool, int kind, int group, void **chunk_dev);
int lumahip_pool_release(lumahip_pool *pool, void *chunk_dev);
/* number of chunks of `kind` (and group, -1 = any) still available */
int lumahip_pool_available(const lumahip_pool *pool, int kind, int group);
/* region group of a chunk handed out by this pool (-1: not grouped / unknown) */
int lumahip_pool_group_of(const lumahip_pool *pool, const void *chunk_dev);
/* what the pool measured, as one JSON object (chunk count, group sizes, probe times, whether grouping was found);
 * the string lives as long as the pool */
const char *lumahip_pool_stats_json(const lumahip_pool *pool);
/* Host-only (no GPU): the pool's grouping step with a caller-supplied measurement, probe(i, r, user) = time of a launch that
 * reads chunk i while writing chunk r.  group_of[i] receives chunk i's group, *ngroups the number of groups (0 when the first
 * round shows no contrast), *fastest (nullable) the fastest pair time seen, *nprobes (nullable) how often probe was called. */
int lumahip_pool_find_groups(int n, double (*probe)(int i, int r, void *user), void *user, int *group_of, int *ngroups,
                             double *fastest, int *nprobes);

/* ---- decoded batches in buffers the LIBRARY places -------------------------------------------------------------------
 * The reference's decoder owns the frame it returns (LumaDecoder::decode() -> &m_frame, include/luma/luma_decoder.h:143-161
 * there).  The device-resident counterpart: a ring of `nbatches` batches of up to `nframes` PACKED LumaFrames (channel c of a
 * frame at frame + c*w*h) that the library allocates -- and places: the frames of a batch rotate over three buffers in three
 * HBM region groups (lumahip_decode_frames_device_rotating's layout; a small pool finds the groups in well under a second), so
 * ONE decode launch writes all three groups: 0.74 of the roofline, against 0.69 for a batch in a single caller-owned buffer
 * wherever that buffer is (lumahip_decode_frames_device).  Without region groups to be found the ring falls
 * back to plain allocations of the same layout.  ctx needs a quantizer set (the pool probes through it).
 *   lumahip_decoded_ring_frame(ring, b, f)  device pointer of frame f of batch b (a packed LumaFrame of 3*w*h floats);
 *   lumahip_decode_frames_device_ring(...)  decodes nframes <= the ring's frames into batch slot b (asynchronous, lanes apply);
 *   lumahip_decoded_ring_info               info = {placed in region groups, batches, frames per batch, groups found}. */
typedef struct lumahip_decoded_ring lumahip_decoded_ring;
int lumahip_decoded_ring_create(lumahip_ctx *ctx, unsigned nbatches, unsigned nframes, unsigned w, unsigned h,
                                lumahip_decoded_ring **out);
void lumahip_decoded_ring_destroy(lumahip_decoded_ring *ring);
int lumahip_decoded_ring_info(const lumahip_decoded_ring *ring, int info[4], size_t *frame_stride);
float *lumahip_decoded_ring_frame(const lumahip_decoded_ring *ring, unsigned batch, unsigned frame);
int lumahip_decode_frames_device_ring(lumahip_ctx *ctx, const unsigned char *const planes_dev[3], const int stride[3],
                                      const size_t plane_frame_stride[3], unsigned nframes, int profile, float sc,
                                      lumahip_decoded_ring *ring, unsigned batch);

/* ---- many GPUs in one process ------------------------------------------------------------------------------------------
 * Replaces the reference's frame loop `for (...) encoder.encode(&frame)` (lumaenc.cpp:205-243; lumadec.cpp:112-160 for
 * decode) for callers that hold a batch of frames: the batch is split into contiguous blocks, one per shard (shard i gets
 * frames lumahip_shard_range(n, i, nshards), the first n % nshards shards one frame more -- block, not round-robin, so each
 * shard's output is already in stream order for the sequential VP9 consumer, src/luma_encoder.cpp:229-257), every shard has
 * its own lumahip_ctx on its GPU and its own host thread, and no data-path communication.  The quantizer is built ONCE on
 * the host and reaches the other GPUs by an RCCL broadcast (ncclCommInitAll + ncclBroadcast over xGMI) from the first
 * device.  `devices` may name a device several times (several shards, i.e. several contexts and streams, on one GPU).
 * Results are identical to a single context processing the frames in order. */
typedef struct lumahip_multi lumahip_multi;
int lumahip_multi_create(lumahip_multi **out, const int *devices, int nshards);   /* devices NULL: all visible devices */
void lumahip_multi_destroy(lumahip_multi *m);
int lumahip_multi_shards(const lumahip_multi *m);
lumahip_ctx *lumahip_multi_ctx(lumahip_multi *m, int shard);   /* the shard's context (owned by m) */
const char *lumahip_multi_last_error(const lumahip_multi *m);
/* 1 when the last table reached the devices through RCCL; 0 when every shard's context took it from the host: only one
 * distinct device is in use (nothing to broadcast), or librccl could not be loaded (nothing to broadcast with), or host
 * copies were asked for.  lumahip_multi_transport_note says which, in words. */
int lumahip_multi_used_rccl(const lumahip_multi *m);
const char *lumahip_multi_transport_note(const lumahip_multi *m);
/* How lumahip_multi_set_quantizer carries the table: 0 (default) = RCCL broadcast when the shards span several devices and
 * librccl loads, host copies otherwise; 1 = always RCCL, also as a one-rank communicator on a single device (fails when
 * librccl cannot be loaded); 2 = always host copies.  The devices end up with the same table either way. */
int lumahip_multi_set_transport(lumahip_multi *m, int mode);
int lumahip_shard_range(unsigned nframes, int shard, int nshards, unsigned *first, unsigned *count);
/* same arguments as lumahip_set_quantizer; the table is uploaded to the first device and broadcast to the others */
int lumahip_multi_set_quantizer(lumahip_multi *m, int ptf, unsigned bitdepth, int colorspace, unsigned bitdepthC,
                                float maxLum, float minLum, const float *lut_host, size_t lut_len);
/* same arguments and results as lumahip_encode_frames_host / lumahip_decode_frames_host */
int lumahip_multi_encode_frames_host(lumahip_multi *m, const float *const *rgb, unsigned nframes, unsigned w, unsigned h,
                                     float sc, int profile, unsigned char *const *planes, const int stride[3],
                                     float *mean_lum);
int lumahip_multi_decode_frames_host(lumahip_multi *m, const unsigned char *const *planes, const int stride[3],
                                     unsigned nframes, unsigned w, unsigned h, int profile, float sc, float *const *rgb_out);
/* Device-resident form: shard i's block of `count[i]` frames lives on shard i's GPU at rgb_dev[i] (frames frame_stride floats
 * apart) with planes at planes_dev[3*i + p] (plane_frame_stride[p] bytes apart).  Enqueues one batched launch per shard
 * (asynchronous); lumahip_multi_sync waits for all shards. */
int lumahip_multi_encode_frames_device(lumahip_multi *m, const float *const *rgb_dev, size_t frame_stride,
                                       const unsigned *count, unsigned w, unsigned h, float sc, int profile,
                                       unsigned char *const *planes_dev, const int stride[3],
                                       const size_t plane_frame_stride[3]);
int lumahip_multi_decode_frames_device(lumahip_multi *m, const unsigned char *const *planes_dev, const int stride[3],
                                       const size_t plane_frame_stride[3], const unsigned *count, unsigned w, unsigned h,
                                       int profile, float sc, float *const *rgb_dev, size_t frame_stride);
int lumahip_multi_sync(lumahip_multi *m);

/* ==== EXPERIMENTAL: measurement and test hooks ===========================================================================
 * Everything below exists for this repository's tests, bench.py and the tools under tools/bench/: probes that run parts of
 * the kernels on synthetic bit patterns, host-only views of the tables the library builds, the synthetic-frame generator
 * and a launch timer.  The library always exports them, but they are NOT part of the drop-in surface: nothing a caller of
 * the reference needs is here, and they may change without a new LUMAHIP_ABI_VERSION.  Declared only when
 * LUMAHIP_EXPERIMENTAL is defined before this header is included. */
#ifdef LUMAHIP_EXPERIMENTAL

/* Host-only (no GPU, no context): the threshold records lumahip_set_quantizer builds for a monotone finite table
 * (lumahdrv_amd/csrc/lut_index.hpp): quantize(v) = (rec[clamp(bits(v) >> shift, kmin, kmin+nbuckets-1) - kmin]
 * + (bits(v) & (2^shift - 1))) >> shift for every float that is not a sign-set NaN (those give maxVal).
 * info = {ok, mantissa bits of the key, shift, kmin, nbuckets}; ok = 0 when the table does not qualify (NaNs,
 * decreasing or duplicate entries, too many records) and the kernels run the literal bisection instead.
 * rec_out (nullable, rec_cap entries) receives the records. */
int lumahip_thresh_index_host(const float *lut, size_t n, int info[5], uint32_t *rec_out, size_t rec_cap);

/* Host-only (no GPU, no context): the VALUE-keyed records of a monotone finite table whose thresholds are (about) evenly
 * spaced (PTF_LINEAR, src/luma_quantizer.cpp:200-203; lumahdrv_amd/csrc/lut_index.hpp LinIndex):
 * key = (uint32)max(fminf(v * kscale, nbuckets - 1), 0) with the product rounded to fp32, rec[2 key] = P (the bit pattern just
 * below the bucket's threshold; 0x7fffffff: none), rec[2 key + 1] = start, quantize(v) = start + ((int32)bits(v) > (int32)P) for
 * EVERY float.  info = {ok, nbuckets, the bits of
 * kscale}; ok = 0 when no scale keeps two thresholds apart within 32768 buckets (PQ, LOG: their thresholds crowd near zero). */
int lumahip_lin_index_host(const float *lut, size_t n, int info[3], uint32_t *rec_out, size_t rec_cap);

/* Host-only (no GPU, no context): the two per-stream tables of the YCbCr kernels, built with the host libm as the reference
 * would evaluate them per pixel.  (1) The threshold records -- same format and lookup as lumahip_thresh_index_host, for
 * arguments t >= +0 or NaN -- of the composite function  t -> quantize(PQdec(t / 255), 0)  with t = 219 y + 16, y the pixel's luma
 * (src/luma_quantizer.cpp:337, 496-500, 222-235), from which the encode kernels take a pixel's luminance code.
 * (2) out[i] = (255 PQenc(lut[i]) - 16) / 219 (src/luma_quantizer.cpp:447-448, 491-494), which the decode kernels read
 * instead of evaluating PQenc per pixel. */
int lumahip_ycbcr_luma_index_host(const float *lut, size_t n, float maxLum, int info[5], uint32_t *rec_out, size_t rec_cap);
int lumahip_ycbcr_ytab_host(const float *lut, size_t n, float maxLum, float *out);

/* Synthetic benchmark input, generated on the device by the integer-only recipe of SURVEY.md 8(d)
 * (identical to the oracle's lo_synth_frame): frame index first_frame + f at dst_dev + f*frame_stride. */
int lumahip_synth_frames_device(lumahip_ctx *ctx, float *dst_dev, size_t frame_stride, unsigned nframes,
                                unsigned w, unsigned h, uint64_t seed, uint64_t first_frame);

/* Timing helper for benchmarks: runs `iters` encode (dir=0) or decode (dir=1) launches of the same
 * arguments back to back on the context's stream between two hipEvents and returns the average
 * kernel-launch duration in milliseconds (events are recorded on the stream the kernels run on).
 * dir = 2: an EMPTY kernel instead (the other arguments are ignored) -- the floor of this way of timing, ~6 us per launch on an
 * MI355X, which every one-launch figure contains (profiles/r05_single_frame.txt). */
int lumahip_time_launches(lumahip_ctx *ctx, int dir, int iters, const float *rgb_dev, size_t frame_stride,
                          unsigned nframes, unsigned w, unsigned h, float sc, int profile,
                          unsigned char *const planes_dev[3], const int stride[3],
                          const size_t plane_frame_stride[3], float *avg_ms);

/* Test probe: the luminance search exactly as the encode kernels instantiate it (four values per thread, the
 * context's search mode; nonneg != 0 selects the Lu'v' kernels' variant, which relies on every value being >= 0 or NaN)
 * over the n consecutive fp32 bit patterns first_bits, first_bits+1, ...: out_dev[i] = code.  n % 4 == 0.
 * Counterpart of LumaQuantizer::quantize(val, 0) (src/luma_quantizer.cpp:222-235). */
int lumahip_quantize_probe_device(lumahip_ctx *ctx, uint16_t *out_dev, uint32_t first_bits, size_t n, int nonneg);

/* Test probe: out[i] = the device powf (pow_glibc.hpp) of the float whose bit pattern is first_bits + i, raised to
 * y; regular = 1 selects the branch-free form + fallback that the YCbCr kernels use; regular = 2 the folded form of the two
 * narrow-range powers (y = 1/78.8438f: arguments in [2^-21, 1] only; y = 78.8438f: arguments in [0.7, 1.4) only).  Lets the
 * tests compare the device function with the host libm exhaustively. */
int lumahip_powf_probe_device(lumahip_ctx *ctx, float *out_dev, uint32_t first_bits, size_t n, float y, int regular);

/* Test probe (YCbCr quantizers): out[i] = the luminance code of a pixel whose t = 219 y + 16 (y = its luma,
 * src/luma_quantizer.cpp:335-337) is the float with bit pattern first_bits + i.  direct = 0: through the composite threshold
 * records exactly as the encode kernels read them; direct = 1: the reference's arithmetic, PQdec(t / 255) then LumaQuantizer::quantize(., 0)
 * (src/luma_quantizer.cpp:337, 496-500, 222-235), evaluated on the device with the complete powf and IEEE division.
 * n % 4 == 0.  LUMAHIP_ERR_UNSUPPORTED when the table has no composite records. */
int lumahip_ycbcr_luma_probe_device(lumahip_ctx *ctx, uint16_t *out_dev, uint32_t first_bits, size_t n, int direct);

/* Benchmark probe: the loads and stores of the 4:2:0 16-bit encode kernel with no arithmetic in between (same
 * tile order, same access widths, non-temporal), `iters` launches, average milliseconds.  OVERWRITES the planes
 * with garbage.  What the memory system alone needs for the encode traffic mix on this device. */
int lumahip_probe_encode_traffic_device(lumahip_ctx *ctx, const float *rgb_dev, size_t frame_stride, unsigned nframes,
                                        unsigned w, unsigned h, unsigned char *const planes_dev[3], const int stride[3],
                                        const size_t plane_frame_stride[3], int iters, float *avg_ms);

/* The decode counterpart: the loads and stores of the 4:2:0 16-bit decode kernel (3 B read + 12 B written per pixel) with no
 * arithmetic.  OVERWRITES the frames with garbage.  Float frames as in lumahip_decode_frames_device_planar. */
int lumahip_probe_decode_traffic_device(lumahip_ctx *ctx, const unsigned char *const planes_dev[3], const int stride[3],
                                        const size_t plane_frame_stride[3], unsigned nframes, unsigned w, unsigned h,
                                        float *const rgb_planes_dev[3], size_t frame_stride, int iters, float *avg_ms);

#endif /* LUMAHIP_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif /* LUMAHIP_H */
