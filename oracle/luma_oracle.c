/*
 * luma_oracle.c -- CPU restatement of Luma HDRv's quantize / dequantize hot path.
 * TEST INFRASTRUCTURE ONLY (see luma_oracle.h for who may use it and how it is pinned).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  The reference is built
 * with no -O / -march / fast-math flags (CMakeLists.txt:15-17), i.e. every + - * / is rounded
 * to fp32 individually and nothing is contracted; -ffp-contract=off keeps that true here.
 *
 * libstdc++'s std::min(a,b) is (b<a)?b:a and std::max(a,b) is (a<b)?b:a -- NOT fminf/fmaxf:
 * they differ when an argument is NaN, and the reference's NaN behaviour is part of parity
 * (SURVEY.md section 5, quirk 1).
 */
#include "luma_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define STD_MIN(a, b) (((b) < (a)) ? (b) : (a))
#define STD_MAX(a, b) (((a) < (b)) ? (b) : (a))

/* include/luma/luma_quantizer.h:79-87 */
static const float rgb2xyz[3][3] = {{0.412424f, 0.357579f, 0.180464f},
                                    {0.212656f, 0.715158f, 0.072186f},
                                    {0.019332f, 0.119193f, 0.950444f}};
static const float xyz2rgb[3][3] = {{3.240708f, -1.537259f, -0.498570f},
                                    {-0.969257f, 1.875995f, 0.041555f},
                                    {0.055636f, -0.203996f, 1.057069f}};

/* ---------------------------------------------------------------- quantizer set-up */

int lo_quantizer_init(lo_quantizer *q)
{
    /* src/luma_quantizer.cpp:44-51 */
    memset(q, 0, sizeof *q);
    q->Lmax = 10000.0f;
    q->Lmin = 0.005f;
    q->mapping = NULL;
    q->cs = LO_CS_LUV;
    return 0;
}

void lo_quantizer_free(lo_quantizer *q)
{
    free(q->mapping);
    q->mapping = NULL;
}

/* src/luma_quantizer.cpp:485-501.  The literals are double literals narrowed to `const float`. */
float lo_transform_pq(float Lmax, float val, int encode)
{
    const float L = Lmax, m = 78.8438, n = 0.1593, c1 = 0.8359, c2 = 18.8516, c3 = 18.6875;
    if (encode) {
        float Lp = powf(val / L, n);
        return powf((c1 + c2 * Lp) / (1 + c3 * Lp), m);
    } else {
        float Vp = powf(val, 1.0f / m);
        float d = Vp - c1;
        return L * powf(STD_MAX(0.0f, d) / (c2 - c3 * Vp), 1.0f / n);
    }
}

/* src/luma_quantizer.cpp:504-510 */
float lo_transform_log(float Lmax, float Lmin, float val, int encode)
{
    if (encode)
        return (log10f(val) - log10f(Lmin)) / (log10f(Lmax) - log10f(Lmin));
    else
        return powf(10.0f, val * (log10f(Lmax) - log10f(Lmin)) + log10f(Lmin));
}

int lo_set_quantizer(lo_quantizer *q, int ptf, unsigned bitdepth, int cs, unsigned bitdepthC,
                     float maxLum, float minLum, const float *table, size_t table_len)
{
    /* src/luma_quantizer.cpp:172-212 */
    size_t i;
    free(q->mapping);
    q->mapping = NULL;
    q->ptf = ptf;
    q->bitdepth = bitdepth;
    q->maxVal = (unsigned)((int)powf(2.0f, (float)bitdepth) - 1);
    q->cs = cs;
    q->bitdepthC = bitdepthC;
    q->maxValColor = (unsigned)((int)powf(2.0f, (float)bitdepthC) - 1);
    q->Lmax = maxLum;
    q->Lmin = minLum;
    q->mapping = (float *)malloc(((size_t)q->maxVal + 1) * sizeof(float));
    if (!q->mapping)
        return -1;

    switch (ptf) {
    case LO_PTF_PQ: /* :114-118 */
        for (i = 0; i <= q->maxVal; i++)
            q->mapping[i] = lo_transform_pq(q->Lmax, (float)i / q->maxVal, 0);
        break;
    case LO_PTF_LOG: /* :121-125 */
        for (i = 0; i <= q->maxVal; i++)
            q->mapping[i] = lo_transform_log(q->Lmax, q->Lmin, (float)i / q->maxVal, 0);
        break;
    case LO_PTF_LINEAR: /* :200-203 */
        for (i = 0; i <= q->maxVal; i++)
            q->mapping[i] = q->Lmax * ((float)i / q->maxVal);
        break;
    case LO_PTF_JND_HDRVDP: /* :128-147 */
    case LO_PTF_PSI:        /* :150-169 */
    default:
        /* the reference copies the first maxVal+1 entries of a compiled-in table (10/11-bit
         * tables for those depths, the 12-bit table for anything else) and reads out of bounds
         * when the table is shorter (quirk 3); the restatement rejects that case instead. */
        if (!table || table_len < (size_t)q->maxVal + 1) {
            free(q->mapping);
            q->mapping = NULL;
            return -1;
        }
        for (i = 0; i <= q->maxVal; i++)
            q->mapping[i] = table[i];
        break;
    }
    return 0;
}

int lo_overwrite_mapping(lo_quantizer *q, const float *lut, size_t n)
{
    if (!q->mapping || n > (size_t)q->maxVal + 1)
        return -1;
    memcpy(q->mapping, lut, n * sizeof(float));
    return 0;
}

/* ---------------------------------------------------------------- per-value quantize / dequantize */

float lo_quantize(const lo_quantizer *q, float val, unsigned ch)
{
    /* src/luma_quantizer.cpp:215-244 */
    float res;
    if (ch == 0 || q->cs == LO_CS_RGB || q->cs == LO_CS_XYZ) {
        int l = 0, r = (int)q->maxVal;
        while (l + 1 < r) {
            int m = (l + r) / 2;
            if (val < q->mapping[m])
                r = m;
            else
                l = m;
        }
        if (val - q->mapping[l] < q->mapping[r] - val)
            res = (float)l;
        else
            res = (float)r;
    } else {
        float maxC = (float)q->maxValColor;
        res = floorf(maxC * val + 0.5f);
        res = STD_MIN(maxC, res);
        res = STD_MAX(0.0f, res);
    }
    return res;
}

float lo_dequantize(const lo_quantizer *q, float val, unsigned ch)
{
    /* src/luma_quantizer.cpp:247-264 */
    float res;
    if (ch == 0 || q->cs == LO_CS_RGB || q->cs == LO_CS_XYZ) {
        if (val < 0)
            res = q->mapping[0];
        else if (val >= q->maxVal)
            res = q->mapping[q->maxVal];
        else
            res = q->mapping[(int)val];
    } else {
        float t = val / q->maxValColor;
        res = STD_MAX(t, 1e-10f);
    }
    return res;
}

/* ---------------------------------------------------------------- colour transform */

static inline float clampXYZ(float v)
{
    /* std::max(std::min(v, 100000000.0f), 0.0001f) */
    float t = STD_MIN(v, 100000000.0f);
    return STD_MAX(t, 0.0001f);
}

static int transform_rows(const lo_quantizer *q, float *c0, float *c1, float *c2, size_t i0, size_t i1,
                          int toCs, float sc)
{
    size_t i;
    if (toCs) {
        switch (q->cs) {
        case LO_CS_XYZ: /* src/luma_quantizer.cpp:273-290 */
            for (i = i0; i < i1; i++) {
                float R = c0[i] * sc, G = c1[i] * sc, B = c2[i] * sc;
                c0[i] = clampXYZ(rgb2xyz[0][0] * R + rgb2xyz[0][1] * G + rgb2xyz[0][2] * B);
                c1[i] = clampXYZ(rgb2xyz[1][0] * R + rgb2xyz[1][1] * G + rgb2xyz[1][2] * B);
                c2[i] = clampXYZ(rgb2xyz[2][0] * R + rgb2xyz[2][1] * G + rgb2xyz[2][2] * B);
            }
            break;
        case LO_CS_LUV: /* :291-316 */
            for (i = i0; i < i1; i++) {
                float R = c0[i] * sc, G = c1[i] * sc, B = c2[i] * sc;
                float X = clampXYZ(rgb2xyz[0][0] * R + rgb2xyz[0][1] * G + rgb2xyz[0][2] * B);
                float Y = clampXYZ(rgb2xyz[1][0] * R + rgb2xyz[1][1] * G + rgb2xyz[1][2] * B);
                float Z = clampXYZ(rgb2xyz[2][0] * R + rgb2xyz[2][1] * G + rgb2xyz[2][2] * B);
                float sum = X + Y + Z;
                float x = X / sum;
                float y = Y / sum;
                c0[i] = Y;
                c1[i] = 4.0f * x / (-2.0f * x + 12.0f * y + 3.0f) * 410.f / 255.0f;
                c2[i] = 9.0f * y / (-2.0f * x + 12.0f * y + 3.0f) * 410.f / 255.0f;
            }
            break;
        case LO_CS_YCBCR: /* :317-354 */
            for (i = i0; i < i1; i++) {
                float r = c0[i] * sc, g = c1[i] * sc, b = c2[i] * sc, y;
                float R = lo_transform_pq(q->Lmax, STD_MAX(r, 1e-10f), 1);
                float G = lo_transform_pq(q->Lmax, STD_MAX(g, 1e-10f), 1);
                float B = lo_transform_pq(q->Lmax, STD_MAX(b, 1e-10f), 1);
                y = 0.2627f * R + 0.6780f * G + 0.0593f * B;
                c0[i] = lo_transform_pq(q->Lmax, (219.0f * y + 16.0f) / 255.0f, 0);
                c1[i] = (224.0f * ((B - y) / 1.8814f) + 128.0f) / 255.0f;
                c2[i] = (224.0f * ((R - y) / 1.4746f) + 128.0f) / 255.0f;
            }
            break;
        case LO_CS_RGB: /* :355-367 */
            for (i = i0; i < i1; i++) {
                c0[i] *= sc;
                c1[i] *= sc;
                c2[i] *= sc;
            }
            break;
        default: /* :368-371 */
            return 0;
        }
    } else {
        switch (q->cs) {
        case LO_CS_XYZ: /* :378-395 */
            for (i = i0; i < i1; i++) {
                float X = c0[i], Y = c1[i], Z = c2[i];
                c0[i] = (xyz2rgb[0][0] * X + xyz2rgb[0][1] * Y + xyz2rgb[0][2] * Z) / sc;
                c1[i] = (xyz2rgb[1][0] * X + xyz2rgb[1][1] * Y + xyz2rgb[1][2] * Z) / sc;
                c2[i] = (xyz2rgb[2][0] * X + xyz2rgb[2][1] * Y + xyz2rgb[2][2] * Z) / sc;
            }
            break;
        case LO_CS_LUV: /* :396-421 */
            for (i = i0; i < i1; i++) {
                float L = c0[i];
                float u = c1[i] * 255.0f / 410.0f;
                float v = c2[i] * 255.0f / 410.0f;
                float x = 9.0f * u / (6.0f * u - 16.0f * v + 12.0f);
                float y = 4.0f * v / (6.0f * u - 16.0f * v + 12.0f);
                float Y = clampXYZ(L);
                float X = clampXYZ(x / y * L);
                float Z = clampXYZ((1.0f - x - y) / y * L);
                c0[i] = (xyz2rgb[0][0] * X + xyz2rgb[0][1] * Y + xyz2rgb[0][2] * Z) / sc;
                c1[i] = (xyz2rgb[1][0] * X + xyz2rgb[1][1] * Y + xyz2rgb[1][2] * Z) / sc;
                c2[i] = (xyz2rgb[2][0] * X + xyz2rgb[2][1] * Y + xyz2rgb[2][2] * Z) / sc;
            }
            break;
        case LO_CS_RGB: /* :422-435 */
            for (i = i0; i < i1; i++) {
                c0[i] /= sc;
                c1[i] /= sc;
                c2[i] /= sc;
            }
            break;
        case LO_CS_YCBCR: /* :436-473 */
            for (i = i0; i < i1; i++) {
                float y = lo_transform_pq(q->Lmax, c0[i], 1), red, green, blue, t;
                y = (255.0f * y - 16.0f) / 219.0f;
                blue = y + 1.8814f * (255.0f * c1[i] - 128.0f) / 224.0f;
                red = y + 1.4746f * (255.0f * c2[i] - 128.0f) / 224.0f;
                green = (y - 0.2627f * red - 0.0593f * blue) / 0.6780f;
                t = STD_MIN(1.0f, red);
                red = STD_MAX(0.0f, t);
                t = STD_MIN(1.0f, green);
                green = STD_MAX(0.0f, t);
                t = STD_MIN(1.0f, blue);
                blue = STD_MAX(0.0f, t);
                c0[i] = lo_transform_pq(q->Lmax, red, 0) / sc;
                c1[i] = lo_transform_pq(q->Lmax, green, 0) / sc;
                c2[i] = lo_transform_pq(q->Lmax, blue, 0) / sc;
            }
            break;
        default: /* :474-477 */
            return 0;
        }
    }
    return 1;
}

int lo_transform_color_space(const lo_quantizer *q, float *buf, unsigned w, unsigned h, int toCs, float sc)
{
    size_t n = (size_t)w * h;
    return transform_rows(q, buf, buf + n, buf + 2 * n, 0, n, toCs, sc);
}

/* ---------------------------------------------------------------- plane pack / unpack */

/* rows [y0,y1) of the destination plane; returns the float sum of what the reference adds to `avg` */
static float pack_rows(const lo_quantizer *q, const float *src, int plane, int profile, unsigned d_w,
                       unsigned d_h, unsigned char *buf, int stride, int y0, int y1)
{
    /* src/luma_encoder.cpp:260-311 */
    const int sub = (profile == 2 || profile == 0);
    const int w = (plane > 0 && sub) ? (int)((d_w + 1) >> 1) : (int)d_w;
    const int m = (profile > 1) ? 2 : 1; /* VPX_IMG_FMT_HIGHBITDEPTH for I42016 / I44416 */
    float avg = 0.0f;
    int x, y;
    (void)d_h;
    for (y = y0; y < y1; y++) {
        for (x = 0; x < w; x++) {
            float res;
            if (plane && sub) {
                size_t ind1 = 2 * (size_t)x + 4 * (size_t)y * w;
                size_t ind2 = ind1 + 2 * (size_t)w;
                res = 0.25f * (src[ind1] + src[ind1 + 1] + src[ind2] + src[ind2 + 1]);
            } else {
                res = src[x + (size_t)y * w];
                avg += res;
            }
            res = lo_quantize(q, res, (unsigned)plane);
            if (profile > 1) {
                unsigned char bl = (unsigned char)(res / 256);
                unsigned char bh = (unsigned char)(res - bl * 256);
                buf[m * x + (size_t)y * stride + 1] = bl;
                buf[m * x + (size_t)y * stride] = bh;
            } else {
                /* float -> unsigned char of a value > 255 is UB in the reference (quirk 2); on
                 * x86-64 gcc it truncates to int then keeps the low 8 bits.  Restated as that. */
                buf[m * x + (size_t)y * stride] = (unsigned char)(int)res;
            }
        }
    }
    return avg;
}

void lo_pack_plane(const lo_quantizer *q, const float *src, int plane, int profile, unsigned d_w,
                   unsigned d_h, unsigned char *buf, int stride, float *avg)
{
    const int sub = (profile == 2 || profile == 0);
    const int w = (plane > 0 && sub) ? (int)((d_w + 1) >> 1) : (int)d_w;
    const int h = (plane > 0 && sub) ? (int)((d_h + 1) >> 1) : (int)d_h;
    float a = pack_rows(q, src, plane, profile, d_w, d_h, buf, stride, 0, h);
    a /= (w * h); /* src/luma_encoder.cpp:314 */
    if (avg)
        *avg = a;
}

static void unpack_rows(const lo_quantizer *q, const unsigned char *buf, int stride, int plane, int profile,
                        unsigned d_w, unsigned d_h, float *dest, int y0, int y1)
{
    /* src/luma_decoder.cpp:205-240; width/height/profile derivation :151-162 */
    const int sub = (profile == 2 || profile == 0);
    const int w = (plane > 0 && sub) ? (int)((d_w + 1) >> 1) : (int)d_w;
    int x, y;
    (void)d_h;
    for (y = y0; y < y1; y++) {
        for (x = 0; x < w; x++) {
            float val;
            if (profile > 1)
                val = lo_dequantize(q, buf[2 * x + (size_t)y * stride + 1] * 256.0f + buf[2 * x + (size_t)y * stride],
                                    (unsigned)plane);
            else
                val = lo_dequantize(q, buf[x + (size_t)y * stride], (unsigned)plane);
            if (plane && sub) {
                size_t ind1 = 2 * (size_t)x + 4 * (size_t)y * w;
                size_t ind2 = ind1 + 2 * (size_t)w;
                dest[ind1] = dest[ind1 + 1] = dest[ind2] = dest[ind2 + 1] = val;
            } else {
                dest[x + (size_t)y * w] = val;
            }
        }
    }
}

void lo_unpack_plane(const lo_quantizer *q, const unsigned char *buf, int stride, int plane, int profile,
                     unsigned d_w, unsigned d_h, float *dest)
{
    const int sub = (profile == 2 || profile == 0);
    const int h = (plane > 0 && sub) ? (int)((d_h + 1) >> 1) : (int)d_h;
    unpack_rows(q, buf, stride, plane, profile, d_w, d_h, dest, 0, h);
}

/* ---------------------------------------------------------------- whole-frame drivers */

void lo_encode_frame(const lo_quantizer *q, float *frame, unsigned w, unsigned h, float sc, int profile,
                     unsigned char *const planes[3], const int stride[3], float *avg_lum)
{
    /* include/luma/luma_encoder.h:142-148, src/luma_encoder.cpp:196-201 */
    size_t n = (size_t)w * h;
    int p;
    lo_transform_color_space(q, frame, w, h, 1, sc);
    for (p = 0; p < 3; p++)
        lo_pack_plane(q, frame + p * n, p, profile, w, h, planes[p], stride[p], p == 0 ? avg_lum : NULL);
}

void lo_decode_frame(const lo_quantizer *q, const unsigned char *const planes[3], const int stride[3],
                     unsigned w, unsigned h, int profile, float sc, float *frame)
{
    /* include/luma/luma_decoder.h:143-161 */
    size_t n = (size_t)w * h;
    int p;
    for (p = 0; p < 3; p++)
        lo_unpack_plane(q, planes[p], stride[p], p, profile, w, h, frame + p * n);
    lo_transform_color_space(q, frame, w, h, 0, sc);
}

typedef struct {
    const lo_quantizer *q;
    float *frame;
    unsigned w, h;
    float sc;
    int profile;
    unsigned char *planes[3];
    int stride[3];
    int row0, row1; /* luma rows, even boundaries */
    int encode;
    float sum;
} band_job;

static void *band_main(void *arg)
{
    band_job *j = (band_job *)arg;
    size_t n = (size_t)j->w * j->h;
    const int sub = (j->profile == 2 || j->profile == 0);
    int p;
    if (j->encode) {
        transform_rows(j->q, j->frame, j->frame + n, j->frame + 2 * n, (size_t)j->row0 * j->w,
                       (size_t)j->row1 * j->w, 1, j->sc);
        for (p = 0; p < 3; p++) {
            int y0 = (p && sub) ? j->row0 / 2 : j->row0, y1 = (p && sub) ? j->row1 / 2 : j->row1;
            float s = pack_rows(j->q, j->frame + p * n, p, j->profile, j->w, j->h, j->planes[p], j->stride[p], y0, y1);
            if (p == 0)
                j->sum = s;
        }
    } else {
        for (p = 0; p < 3; p++) {
            int y0 = (p && sub) ? j->row0 / 2 : j->row0, y1 = (p && sub) ? j->row1 / 2 : j->row1;
            unpack_rows(j->q, j->planes[p], j->stride[p], p, j->profile, j->w, j->h, j->frame + p * n, y0, y1);
        }
        transform_rows(j->q, j->frame, j->frame + n, j->frame + 2 * n, (size_t)j->row0 * j->w,
                       (size_t)j->row1 * j->w, 0, j->sc);
    }
    return NULL;
}

static void run_bands(band_job *proto, int nthreads, float *sum_out)
{
    int t, pairs = (int)(proto->h / 2);
    pthread_t *th;
    band_job *jobs;
    float total = 0.0f;
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > pairs)
        nthreads = pairs > 0 ? pairs : 1;
    th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    jobs = (band_job *)malloc(sizeof(band_job) * nthreads);
    for (t = 0; t < nthreads; t++) {
        jobs[t] = *proto;
        jobs[t].row0 = 2 * (int)((long)pairs * t / nthreads);
        jobs[t].row1 = 2 * (int)((long)pairs * (t + 1) / nthreads);
        jobs[t].sum = 0.0f;
        pthread_create(&th[t], NULL, band_main, &jobs[t]);
    }
    for (t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        total += jobs[t].sum;
    }
    if (sum_out)
        *sum_out = total;
    free(th);
    free(jobs);
}

void lo_encode_frame_mt(const lo_quantizer *q, float *frame, unsigned w, unsigned h, float sc, int profile,
                        unsigned char *const planes[3], const int stride[3], float *avg_lum, int nthreads)
{
    band_job j;
    float sum = 0.0f;
    int p;
    memset(&j, 0, sizeof j);
    j.q = q; j.frame = frame; j.w = w; j.h = h; j.sc = sc; j.profile = profile; j.encode = 1;
    for (p = 0; p < 3; p++) { j.planes[p] = planes[p]; j.stride[p] = stride[p]; }
    run_bands(&j, nthreads, &sum);
    if (avg_lum)
        *avg_lum = sum / (float)((int)w * (int)h);
}

void lo_decode_frame_mt(const lo_quantizer *q, const unsigned char *const planes[3], const int stride[3],
                        unsigned w, unsigned h, int profile, float sc, float *frame, int nthreads)
{
    band_job j;
    int p;
    memset(&j, 0, sizeof j);
    j.q = q; j.frame = frame; j.w = w; j.h = h; j.sc = sc; j.profile = profile; j.encode = 0;
    for (p = 0; p < 3; p++) { j.planes[p] = (unsigned char *)planes[p]; j.stride[p] = stride[p]; }
    run_bands(&j, nthreads, NULL);
}

/* ---------------------------------------------------------------- libm powf sweep (checker for the device powf) */

typedef struct {
    const float *got;
    uint32_t first;
    size_t i0, i1;
    float y;
    size_t bad;
    uint32_t first_bad;
} powf_job;

static void *powf_main(void *arg)
{
    powf_job *j = (powf_job *)arg;
    size_t i;
    for (i = j->i0; i < j->i1; i++) {
        uint32_t b = j->first + (uint32_t)i, g, e;
        float x, r;
        memcpy(&x, &b, 4);
        r = powf(x, j->y);
        memcpy(&g, &j->got[i], 4);
        memcpy(&e, &r, 4);
        if (g != e && !(r != r && j->got[i] != j->got[i])) {
            if (!j->bad)
                j->first_bad = b;
            j->bad++;
        }
    }
    return NULL;
}

/* compares got[i] with libm powf(bits(first+i), y) for i in [0,n); returns the number of mismatches (NaN == NaN) */
size_t lo_powf_compare(const float *got, uint32_t first, size_t n, float y, int nthreads, uint32_t *first_bad)
{
    pthread_t th[256];
    powf_job jobs[256];
    size_t bad = 0;
    int t;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    for (t = 0; t < nthreads; t++) {
        jobs[t].got = got; jobs[t].first = first; jobs[t].y = y; jobs[t].bad = 0; jobs[t].first_bad = 0;
        jobs[t].i0 = n * (size_t)t / nthreads;
        jobs[t].i1 = n * (size_t)(t + 1) / nthreads;
        pthread_create(&th[t], NULL, powf_main, &jobs[t]);
    }
    for (t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].bad && !bad && first_bad)
            *first_bad = jobs[t].first_bad;
        bad += jobs[t].bad;
    }
    return bad;
}

/* ---------------------------------------------------------------- input generators and digests */

void lo_test_frame(float *buf, unsigned w_, unsigned h_)
{
    /* src/exr_interface.cpp:50-70.  size_t index arithmetic, integer divisions as in the source. */
    size_t w = w_, h = h_, x, y;
    float *c0 = buf, *c1 = buf + w * h, *c2 = buf + 2 * w * h;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            float top = y < h / 10 ? 10000.0f * ((float)(x * x)) / (w * w) : 10000.0f * ((20 * x) / w) / 20.0f;
            if (y < h / 5) {
                c0[x + y * w] = top;
                c1[x + y * w] = top;
                c2[x + y * w] = top;
            } else {
                c0[x + y * w] = 10000.0f * (((20 * y / h) % 2) ^ ((30 * x / w) % 2));
                c1[x + y * w] = 10000.0f * ((20 * y / h) % 2) * ((float)(y * y)) / (h * h);
                c2[x + y * w] = 10000.0f * ((20 * y / h) % 2) * ((float)(x * x)) / (w * w);
            }
        }
}

uint64_t lo_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void lo_synth_frame(float *buf, unsigned w, unsigned h, uint64_t seed, uint64_t frame)
{
    /* log-uniform in [2^-10, 2^14), 10-bit mantissa, finite, positive */
    size_t n = (size_t)w * h, i;
    unsigned ch;
    for (ch = 0; ch < 3; ch++)
        for (i = 0; i < n; i++) {
            uint64_t h64 = lo_splitmix64(seed ^ (frame * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)ch * n + i));
            uint32_t e = 117u + (uint32_t)((h64 >> 40) % 24u);
            uint32_t bits = (e << 23) + (uint32_t)(h64 & 0x7FE000u);
            memcpy(&buf[ch * n + i], &bits, 4);
        }
}

uint64_t lo_fnv1a64_basis(const void *data, size_t n, uint64_t basis)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t hsh = basis;
    size_t i;
    for (i = 0; i < n; i++) {
        hsh ^= p[i];
        hsh *= 0x100000001b3ull;
    }
    return hsh;
}

uint64_t lo_fnv1a64(const void *data, size_t n)
{
    return lo_fnv1a64_basis(data, n, 0xcbf29ce484222325ull);
}

uint64_t lo_fnv1a64_rows(const void *data, size_t row_bytes, size_t rows, size_t stride)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t hsh = 0xcbf29ce484222325ull;
    size_t r, i;
    for (r = 0; r < rows; r++)
        for (i = 0; i < row_bytes; i++) {
            hsh ^= p[r * stride + i];
            hsh *= 0x100000001b3ull;
        }
    return hsh;
}

/* src/lumaplay_dequantizer.frag:145-156 (see luma_oracle.h) */
void lo_display_transform(const float *rgb, size_t n, double exposure, double gamma, int doTmo, int ldrSim,
                          unsigned char *rgba)
{
    size_t i;
    int c;
    for (i = 0; i < n; i++) {
        for (c = 0; c < 3; c++) {
            double v = (double)rgb[(size_t)c * n + i];
            if (ldrSim > 0) {                                   /* :145-146 */
                double f = floor(256.0 * v);
                f = f < 256.0 ? f : 256.0;                      /* min(vec3(256), .) */
                f = f > 1.0 ? f : 1.0;                          /* max(vec3(1), .)   */
                v = exposure * f / 256.0;
            } else {
                v = v * exposure;                               /* :148 */
            }
            if (doTmo > 0) {                                    /* :150-154 */
                const double nn = 0.8, sig = 0.8;
                const double vn = pow(v > 0.0 ? v : 0.0, nn);
                v = vn / (vn + pow(sig, nn));
            }
            v = pow(v > 0.0 ? v : 0.0, 1.0 / gamma);            /* :156 */
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);            /* the 8-bit colour buffer's clamp ... */
            rgba[4 * i + c] = (unsigned char)floor(255.0 * v + 0.5);   /* ... and conversion */
        }
        rgba[4 * i + 3] = 255;
    }
}
