// half_stage.cpp -- float -> binary16 conversion of a host frame with an exactness check, for the half upload of the host
// entry points (lumahip_host.hip: xfer_h2d_f16).  Why: the reference's LumaFrame is float, but its only frame source, the EXR
// reader, fills it with widened halves (src/exr_interface.cpp:77-146); such a frame crosses PCIe in half the bytes and the
// encode kernels widen it back exactly.  F16C (vcvtps2ph / vcvtph2ps) where the CPU has it -- every x86-64 host a MI355X sits
// in -- and no half upload where it does not.  Plain C++, compiled by the host compiler.
#include "half_stage.hpp"

#include <immintrin.h>

namespace lh {

bool f16c_available()
{
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("f16c");
    return ok;
}

// dst[i] = binary16(src[i]) for i < n; returns true iff every value survives the round trip bit for bit (so: is a half; a NaN
// only if its payload fits; -0 and +-inf do)
__attribute__((target("avx2,f16c"))) bool convert_f32_to_f16_checked(const float *src, uint16_t *dst, size_t n)
{
    __m256 bad = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(src + i);
        const __m128i h = _mm256_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        bad = _mm256_or_ps(bad, _mm256_xor_ps(_mm256_cvtph_ps(h), v));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i), h);
    }
    bool ok = _mm256_testz_si256(_mm256_castps_si256(bad), _mm256_castps_si256(bad)) != 0;
    for (; i < n; i++) {
        const __m128 v = _mm_set_ss(src[i]);
        const __m128i h = _mm_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const float back = _mm_cvtss_f32(_mm_cvtph_ps(h));
        dst[i] = (uint16_t)_mm_extract_epi16(h, 0);
        uint32_t a, b;
        __builtin_memcpy(&a, &back, 4);
        __builtin_memcpy(&b, &src[i], 4);
        ok = ok && a == b;
    }
    return ok;
}

}  // namespace lh
