/*
 * luma_oracle.h -- CPU restatement of Luma HDRv's per-pixel quantize / dequantize
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a scalar, plain-C restatement of the reference
 * algorithm, compiled with `gcc -O2 -ffp-contract=off`.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / the reported CPU baseline.  The product (liblumahip.so and
 * the C++ facade) never links, loads or calls anything in oracle/.
 *
 * Pinning: the oracle is checked (tests/test_oracle_*.py) against
 *   (1) the known-answer values and FNV-1a-64 digests recorded from the real
 *       reference build in SURVEY.md section 8(c) (LUT digests, quantize pins,
 *       constant-colour pins, testFrame 1280x720 / 1920x1080 Y/U/V digests),
 *   (2) oracle/_ref/libluma_ref.so = /root/reference/src/luma_quantizer.cpp
 *       compiled unmodified (see oracle/Makefile), bit-for-bit, and
 *   (3) the committed fixtures in tests/golden/ that (2) generated.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef LUMA_ORACLE_H
#define LUMA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum values are serialised by the reference (attachments 432/433), so they
 * are part of the contract: include/luma/luma_quantizer.h:95-96 */
enum { LO_PTF_PSI = 0, LO_PTF_PQ = 1, LO_PTF_LOG = 2, LO_PTF_JND_HDRVDP = 3, LO_PTF_LINEAR = 4 };
enum { LO_CS_LUV = 0, LO_CS_RGB = 1, LO_CS_YCBCR = 2, LO_CS_XYZ = 3 };

typedef struct lo_quantizer {
    int ptf, cs;
    unsigned bitdepth, bitdepthC;
    unsigned maxVal, maxValColor;
    float Lmax, Lmin;
    float *mapping;            /* maxVal+1 entries, owned */
} lo_quantizer;

/* include/luma/luma_quantizer.h:92-111, src/luma_quantizer.cpp:44-57,172-212.
 * `table` supplies the PSI / JND_HDRVDP luminance table (>= maxVal+1 floats) because the
 * reference compiles those in as data (include/luma/ptfs/ *.h); NULL for PQ/LOG/LINEAR.
 * returns 0 on success, -1 on bad arguments. */
int  lo_quantizer_init(lo_quantizer *q);
void lo_quantizer_free(lo_quantizer *q);
int  lo_set_quantizer(lo_quantizer *q, int ptf, unsigned bitdepth, int cs, unsigned bitdepthC,
                      float maxLum, float minLum, const float *table, size_t table_len);
/* overwrite the first n floats of the LUT (what LumaDecoder::initialize does with attachment 434,
 * src/luma_decoder.cpp:121-122) */
int  lo_overwrite_mapping(lo_quantizer *q, const float *lut, size_t n);

float lo_transform_pq(float Lmax, float val, int encode);                 /* src/luma_quantizer.cpp:485-501 */
float lo_transform_log(float Lmax, float Lmin, float val, int encode);   /* src/luma_quantizer.cpp:504-510 */

float lo_quantize(const lo_quantizer *q, float val, unsigned ch);        /* src/luma_quantizer.cpp:215-244 */
float lo_dequantize(const lo_quantizer *q, float val, unsigned ch);      /* src/luma_quantizer.cpp:247-264 */

/* in-place colour transform of a planar frame (channel c at buf + c*h*w);
 * returns 1 (true) on success, 0 on unknown colour space: src/luma_quantizer.cpp:267-482 */
int lo_transform_color_space(const lo_quantizer *q, float *buf, unsigned w, unsigned h, int toCs, float sc);

/* LumaEncoder::setVpxChannel, src/luma_encoder.cpp:260-317.  d_w/d_h are luma dimensions;
 * chroma shifts and sample width follow from the VP9 profile exactly as vpx_img_alloc's format
 * does (src/luma_encoder.cpp:121-128): profile 0 = 4:2:0 8-bit, 1 = 4:4:4 8-bit,
 * 2 = 4:2:0 16-bit, 3 = 4:4:4 16-bit.  Returns plane-0 style average (sum/(w*h)) in *avg. */
void lo_pack_plane(const lo_quantizer *q, const float *src, int plane, int profile,
                   unsigned d_w, unsigned d_h, unsigned char *buf, int stride, float *avg);
/* LumaDecoder::getVpxChannels (one plane), src/luma_decoder.cpp:205-240 */
void lo_unpack_plane(const lo_quantizer *q, const unsigned char *buf, int stride, int plane, int profile,
                     unsigned d_w, unsigned d_h, float *dest);

/* whole-frame drivers: LumaEncoder::encode minus run() (include/luma/luma_encoder.h:142-148) and
 * LumaDecoder::decode minus run() (include/luma/luma_decoder.h:143-161).  encode mutates `frame`
 * exactly as the reference does.  planes[p]/stride[p] as in vpx_image_t. */
void lo_encode_frame(const lo_quantizer *q, float *frame, unsigned w, unsigned h, float sc, int profile,
                     unsigned char *const planes[3], const int stride[3], float *avg_lum);
void lo_decode_frame(const lo_quantizer *q, const unsigned char *const planes[3], const int stride[3],
                     unsigned w, unsigned h, int profile, float sc, float *frame);
/* same, rows sharded over nthreads (bench cpu_baseline with all cores; rows are independent).
 * The luminance average is then a sum of per-band float sums (documented deviation: the value only
 * feeds the reference's `avg <= 1` warning). */
void lo_encode_frame_mt(const lo_quantizer *q, float *frame, unsigned w, unsigned h, float sc, int profile,
                        unsigned char *const planes[3], const int stride[3], float *avg_lum, int nthreads);
void lo_decode_frame_mt(const lo_quantizer *q, const unsigned char *const planes[3], const int stride[3],
                        unsigned w, unsigned h, int profile, float sc, float *frame, int nthreads);

/* Checker for the device powf: compares got[i] with this host's libm powf(float-with-bits(first+i), y), threaded.
 * (The reference calls libm powf, src/luma_quantizer.cpp:485-501.) */
size_t lo_powf_compare(const float *got, uint32_t first, size_t n, float y, int nthreads, uint32_t *first_bad);

/* The player's display-side transform, src/lumaplay_dequantizer.frag:145-156, on ALREADY DECODED linear RGB (the fragment's
 * `RGB` divided by `scaling`, which LumaDecoder::decode has applied), evaluated in binary64:
 *   ldrSim:  v = exposure * max(1, min(256, floor(256 v))) / 256      else  v = v * exposure
 *   doTmo:   v = v^0.8 / (v^0.8 + 0.8^0.8)
 *   out    = v^(1/gamma), clamped to [0, 1] and converted as an 8-bit UNORM colour buffer converts gl_FragColor:
 *            floor(255 v + 0.5); alpha 255.
 * rgb: planar, 3 planes of n floats; rgba: n x 4 bytes, interleaved.  pow of a negative base (GLSL: undefined) is taken of
 * max(v, 0).  The reference reads its table through a GL_LINEAR-filtered texture, so it does not pin these bytes; this
 * restatement is the checker of the fused decode + display kernel (tolerance +-1 code, tests/test_gpu_parity.py). */
void lo_display_transform(const float *rgb, size_t n, double exposure, double gamma, int doTmo, int ldrSim,
                          unsigned char *rgba);

/* ExrInterface::testFrame pattern, src/exr_interface.cpp:50-70 */
void lo_test_frame(float *buf, unsigned w, unsigned h);

/* Synthetic benchmark frames; integer-only so host and device agree bit for bit */
uint64_t lo_splitmix64(uint64_t x);
void lo_synth_frame(float *buf, unsigned w, unsigned h, uint64_t seed, uint64_t frame);

uint64_t lo_fnv1a64(const void *data, size_t n);
/* FNV-1a-64 with a caller-chosen offset basis.  The digests quoted in SURVEY.md 8(c) were produced by a
 * survey probe whose offset basis was 1469598103934665603 (one digit short of the standard
 * 14695981039346656037); tests use this entry point to check those pins as recorded. */
uint64_t lo_fnv1a64_basis(const void *data, size_t n, uint64_t basis);
/* digest of `rows` rows of `row_bytes` bytes each taken every `stride` bytes (tightly-packed view) */
uint64_t lo_fnv1a64_rows(const void *data, size_t row_bytes, size_t rows, size_t stride);

#ifdef __cplusplus
}
#endif
#endif
