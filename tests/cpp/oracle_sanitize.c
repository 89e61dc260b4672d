/* tests/cpp/oracle_sanitize.c -- runs the oracle's whole-frame paths under AddressSanitizer + UBSan
 * (tests/test_oracle_sanitizers.py builds it with -fsanitize=address,undefined -fno-sanitize-recover).
 * Every colour space x profile on a ragged frame with NaN / Inf / negative inputs, plus the generators. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/luma_oracle.h"

int main(void)
{
    const unsigned w = 66, h = 34;
    const size_t n = (size_t)w * h;
    float *frame = malloc(3 * n * sizeof(float)), *back = malloc(3 * n * sizeof(float));
    int cs, profile, ptf;
    unsigned long long acc = 0;
    for (ptf = 1; ptf <= 4; ptf += (ptf == 2 ? 2 : 1)) /* PQ, LOG, LINEAR */
        for (cs = 0; cs < 4; cs++)
            for (profile = 0; profile < 4; profile++) {
                lo_quantizer q;
                const unsigned bits = profile < 2 ? 8 : 11;
                unsigned char *planes[3];
                int stride[3], p;
                float avg;
                lo_quantizer_init(&q);
                if (lo_set_quantizer(&q, ptf, bits, cs, 8, 1e4f, 0.005f, NULL, 0))
                    return 2;
                lo_synth_frame(frame, w, h, 20250929ull, (unsigned long long)(cs * 4 + profile));
                frame[0] = NAN; frame[1] = INFINITY; frame[2] = -5.0f; frame[3] = 0.0f; frame[n + 1] = -INFINITY;
                for (p = 0; p < 3; p++) {
                    const int sub = (profile == 0 || profile == 2) && p;
                    const int pw = sub ? (int)(w + 1) / 2 : (int)w, ph = sub ? (int)(h + 1) / 2 : (int)h;
                    stride[p] = pw * (profile > 1 ? 2 : 1) + 3; /* odd stride */
                    planes[p] = malloc((size_t)ph * stride[p]);
                    memset(planes[p], 0, (size_t)ph * stride[p]);
                }
                lo_encode_frame_mt(&q, frame, w, h, 2.0f, profile, planes, stride, &avg, 3);
                lo_decode_frame_mt(&q, (const unsigned char *const *)planes, stride, w, h, profile, 2.0f, back, 3);
                lo_decode_frame(&q, (const unsigned char *const *)planes, stride, w, h, profile, 2.0f, back);
                for (p = 0; p < 3; p++) {
                    acc += lo_fnv1a64(planes[p], 64);
                    free(planes[p]);
                }
                lo_quantizer_free(&q);
            }
    lo_test_frame(frame, w, h);
    acc += lo_fnv1a64(frame, 3 * n * sizeof(float));
    printf("ok %llx\n", acc);
    free(frame);
    free(back);
    return 0;
}
