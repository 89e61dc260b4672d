// luma_device.hpp -- per-pixel device arithmetic of the Luma HDRv quantize / dequantize path (gfx950).
//
// Bit-exactness rules (DESIGN.md "float semantics"):
//   * compiled with -ffp-contract=off: every + - * / below is one correctly-rounded fp32 operation,
//     evaluated in the reference's association order; fmaf() appears only inside the division helpers,
//     where it is part of a correctly-rounded divide, never as a contraction of reference arithmetic;
//   * std::min / std::max semantics (libstdc++: min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a) are spelled
//     out as compare + select: v_min_f32 / v_max_f32 / v_med3_f32 treat NaN differently and the
//     reference's NaN results (e.g. rgb(NaN,1,1) -> Y=2047,U=255,V=255) are part of parity;
//   * NaN sign / payload never reaches an output: x86 produces 0xFFC00000 where gfx950 produces
//     0x7FC00000, so every consumer of a possibly-NaN value decides by ordered compares only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pow_glibc.hpp"

namespace lh {

// 0..3 are the reference's colorSpace_t values; CS_PACK is internal: "frame is already colour-transformed"
// (LumaEncoder::setChannels / LumaDecoder::getVpxChannels on their own), identity transform, channel 0
// through the LUT and channels 1,2 through the colour quantizer.
enum : int { CS_LUV = 0, CS_RGB = 1, CS_YCBCR = 2, CS_XYZ = 3, CS_PACK = 4 };

#define LH_DEV static __device__ __forceinline__
#define LH_DEVS __device__ __forceinline__  // explicit specialisations take no storage class

LH_DEV float std_min(float a, float b) { return (b < a) ? b : a; }
LH_DEV float std_max(float a, float b) { return (a < b) ? b : a; }

// std::max(std::min(v, 1e8f), 1e-4f): src/luma_quantizer.cpp:285-287,305-307,412-414.  libstdc++'s min / max return
// their FIRST argument when a comparison with NaN is false, so a NaN v passes through both; every other v (+-inf
// included) is clamped.  That is exactly IEEE-754-2019 maximum(minimum(v, 1e8), 1e-4) with its NaN propagation --
// gfx950's v_minimum3_f32 / v_maximum3_f32 (two instructions, no compare + select pairs, no separate NaN test;
// v_min_f32 / v_max_f32 / v_med3_f32 would return the non-NaN operand).  A NaN's sign / payload may differ from the
// x86 result, as everywhere (see the header comment).
LH_DEV float clamp_xyz(float v)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_minimum(v, 100000000.0f), 0.0001f);
}

// std::max(v, 1e-10f), src/luma_quantizer.cpp:331-333: (v < 1e-10f) ? 1e-10f : v -- a NaN v passes through, anything
// else (-0, negatives, -inf) becomes 1e-10f: IEEE-754-2019 maximum, one v_maximum3_f32 instead of compare + select.
LH_DEV float floor_1e10(float v) { return __builtin_elementwise_maximum(v, 1e-10f); }

// ---------------------------------------------------------------------------------------------------
// Division.
//
// div_ieee(a,b): plain fp32 '/', which hipcc expands to the IEEE-correct v_div_scale / v_rcp /
// v_fma x5 / v_div_fmas / v_div_fixup sequence (12 VALU ops).
//
// div_nr(a,b): the same Newton-Raphson core WITHOUT v_div_scale / v_div_fixup.  For operands whose
// magnitudes are far from the fp32 exponent limits (no operand or quotient denormal, exponent
// difference < 96, denominator exponent < 253) v_div_scale returns its input unchanged, v_div_fmas is a
// plain fma and v_div_fixup passes the quotient through, so the result is bit-identical to div_ieee --
// and NaN in gives NaN out either way.  Each call site states the operand range that licenses it.
// rcp_nr(b) exposes the refined reciprocal so that two quotients over one denominator share it.
// ---------------------------------------------------------------------------------------------------
LH_DEV float div_ieee(float a, float b) { return a / b; }

#ifdef LH_NO_FAST_DIV  // A/B switch: every division through the compiler's full IEEE sequence
LH_DEV float rcp_nr(float b) { return b; }
LH_DEV float div_nr_r(float a, float b, float) { return a / b; }
LH_DEV float div_nr(float a, float b) { return a / b; }
#else

LH_DEV float rcp_nr(float b)
{
    float r0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

LH_DEV float div_nr_r(float a, float b, float r1)
{
    float q0 = a * r1;
    float e0 = __builtin_fmaf(-b, q0, a);
    float q1 = __builtin_fmaf(e0, r1, q0);
    float e1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e1, r1, q1);
}

LH_DEV float div_nr(float a, float b) { return div_nr_r(a, b, rcp_nr(b)); }
#endif

// a / 255.0f in three operations (Markstein): q = a*RN(1/255); r = fma(-255,q,a); q' = fma(r,RN(1/255),q).
// Verified exhaustively against fp32 division over all 2^32 bit patterns of a (tools/verify_constdiv.c):
// bit-identical except a = -0 (gives +0) and a = +-inf (gives NaN).  Licensed only where a > 0, finite
// or NaN.  The same construction is NOT exact for 410, 224, 1.8814f, 1.4746f, 0.6780f (checked), so those
// divisors keep div_ieee.
// Same construction for divisor 219 (also verified exhaustively: exact except a = -0 and a = +-inf), and a
// version of /255 whose licence is "a is finite and not -0" (e.g. the sum x + 128.0f, which is never -0).
LH_DEV float div_219_fin(float a)
{
#ifdef LH_NO_FAST_DIV
    return a / 219.0f;
#else
    const float rc = 1.0f / 219.0f;
    const float q = a * rc;
    const float r = __builtin_fmaf(-219.0f, q, a);
    return __builtin_fmaf(r, rc, q);
#endif
}

LH_DEV float div_255_pos(float a)
{
#ifdef LH_NO_FAST_DIV
    return a / 255.0f;
#else
    const float rc = 1.0f / 255.0f;  // 0x1.010102p-8, constant-folded, correctly rounded
    const float q = a * rc;
    const float r = __builtin_fmaf(-255.0f, q, a);
    return __builtin_fmaf(r, rc, q);
#endif
}

// ---------------------------------------------------------------------------------------------------
// Colour transforms, one pixel.
//   xform_fwd: in = r*sc, g*sc, b*sc (the caller applies the reference's `*sc` -- and skips it when
//              sc == 1.0f, x*1.0f being exact); out = the three channel values the reference stores back.
//   xform_inv: out = the reference's values BEFORE its final `/sc` (the caller divides -- and skips the
//              division when sc == 1.0f, x/1.0f being exact).  Exception: ycbcr_inv_n divides itself.
// ---------------------------------------------------------------------------------------------------

// Tab: which powf tables the YCbCr path reads -- PowfTablesWide (33 KiB, one LDS read per log2) everywhere except the
// half-input encode kernels, whose LDS belongs to the 124 KiB table of luma_device.hpp half_lookup and whose (rare) general
// path therefore runs on the 768-byte PowfTables.  Same arithmetic, same results (pow_glibc.hpp).
template <typename Tab>
struct XformConstT {
    float sc;               // preScaling
    float Lmax;             // PQ peak for the YCbCr path
    const Tab *pw;          // powf tables (LDS copy), YCbCr only
    // refined reciprocals (rcp_nr) of the YCbCr path's constant divisors, computed once per thread instead of once
    // per division: div_nr_r(a, b, rcp_nr(b)) is div_nr(a, b) by definition
    float rLmax, r18814, r14746, r224, r0678;
    // YCbCr decode: the final `/ sc` (src/luma_quantizer.cpp:466-468).  sc_mode 0: sc == 1.0f, nothing to do; 1: sc is a normal
    // float in [2^-16, 2^16] and the quotients take the short division on rsc = rcp_nr(sc) (ycbcr_inv_n<., ., SCDIV> states the
    // licence); 2: anything else -- the complete functions with IEEE division
    float rsc;
    int sc_mode;
};
using XformConst = XformConstT<PowfTablesWide>;

template <int CS, typename Tab = PowfTablesWide>
LH_DEV XformConstT<Tab> make_xform_const(float sc, float Lmax, const Tab *pw)
{
    XformConstT<Tab> k;
    k.sc = sc;
    k.Lmax = Lmax;
    k.pw = pw;
    k.rLmax = k.r18814 = k.r14746 = k.r224 = k.r0678 = k.rsc = 0.0f;
    k.sc_mode = (sc == 1.0f) ? 0 : 2;
    if constexpr (CS == CS_YCBCR) {
#ifndef LH_NO_FAST_DIV
        if (sc != 1.0f && sc >= 0x1p-16f && sc <= 0x1p16f) {
            k.sc_mode = 1;
            k.rsc = rcp_nr(sc);
        }
#endif
        k.rLmax = rcp_nr(Lmax);
        k.r18814 = rcp_nr(1.8814f);
        k.r14746 = rcp_nr(1.4746f);
        k.r224 = rcp_nr(224.0f);
        k.r0678 = rcp_nr(0.6780f);
    }
    return k;
}

// LumaQuantizer::transformPQ, src/luma_quantizer.cpp:485-501.  The constants are double literals
// narrowed to `const float` in the reference; 1.0f/m and 1.0f/n are single fp32 divisions.
template <typename K>
LH_DEV float pq_encode(float val, const K &k)
{
    const float m = 78.8438f, n = 0.1593f, c1 = 0.8359f, c2 = 18.8516f, c3 = 18.6875f;
    const float Lp = powf_glibc(div_ieee(val, k.Lmax), n, *k.pw);
    return powf_glibc(div_ieee(c1 + c2 * Lp, 1.0f + c3 * Lp), m, *k.pw);
}

template <typename K>
LH_DEV float pq_decode(float val, const K &k)
{
    const float m = 78.8438f, n = 0.1593f, c1 = 0.8359f, c2 = 18.8516f, c3 = 18.6875f;
    const float Vp = powf_glibc(val, 1.0f / m, *k.pw);
    return k.Lmax * powf_glibc(div_ieee(std_max(0.0f, Vp - c1), c2 - c3 * Vp), 1.0f / n, *k.pw);
}

// The same two functions on the branch-free powf.  A SlowAcc collects, over as many pixels as the caller likes (the
// kernels: one thread's whole unit), whether any argument left the domain in which the straight-line arithmetic
// (powf_regular, div_nr) is licensed; the caller then redoes those pixels with the complete functions and IEEE division
// throughout.  Two collectors, because a compare costs twice an integer max (tools/bench/valu_bench.hip) and keeps a lane
// mask alive in scalar registers: `umax` = running unsigned max of (bits(x) - 0x00800000) over arguments that must be
// positive normal floats (one compare against 0x7f000000 at the end), `flag` for the tests that need their own compare.
// Lmax in [1e-6, 1e9] is checked by the caller.
struct SlowAcc {
    uint32_t umax = 0;
    uint32_t umin = 0xffffffffu;  // running unsigned min of (bits(x) - 1) over arguments that must be +0 or >= pw_range_low
    uint32_t emax = 0;            // running unsigned max of the high word of |y log2 t| over PQdec's second powers (pow_glibc.hpp EMAX)
    bool flag = false;
};
// ELIM: the bound pq_decode_r's callers put on |log2| of PQdec's second power (1 = powf's own 126); 0 = no such power in the unit
template <int ELIM = 0, typename K>
LH_DEV bool slow_any(const SlowAcc &a, const K &k)
{
    return a.flag || a.umax >= pw_range_limit(*k.pw) || a.umin < pw_range_low(*k.pw) - 1u ||
           (ELIM != 0 && a.emax >= pw_emax_limit(ELIM == 1 ? 126 : ELIM));
}

// pq_encode_r<ANYVAL>: with ANYVAL (decode side: val is a table value, possibly 0, tiny, negative or NaN) the first
// power tests its argument itself (ZERO, CHECK_X).  Without (encode side: val in [1e-10, FLT_MAX], or NaN / +inf) the
// argument val / Lmax joins `umax` -- it may overflow or be NaN.  When that test passes, x1 = val/Lmax is a positive
// normal float (or +0): |n*log2(x1)| <= 20.4, so Lp in {0} u [7e-7, 1.4e6]; c1 + c2*Lp in [0.83, 2.7e7] and
// 1 + c3*Lp in [1, 2.7e7] are normal, their quotient x2 lies in [0.8359, 1.0088], and m*log2(x2) in [-20.4, 1.0]:
// the second power needs no test at all, and the result lies in [7.3e-7, 1.995].
template <bool ANYVAL, typename K>
LH_DEV float pq_encode_r(float val, const K &k, SlowAcc &acc)
{
    const float m = 78.8438f, n = 0.1593f, c1 = 0.8359f, c2 = 18.8516f, c3 = 18.6875f;
    const float x1 = div_nr_r(val, k.Lmax, k.rLmax);
    float Lp;
    if constexpr (ANYVAL) {
        Lp = powf_regular<true, true, false>(x1, n, *k.pw, acc.flag);
    } else {
        acc.umax = max(acc.umax, pw_range_key(*k.pw, __float_as_uint(x1)));
        Lp = powf_regular<false, false, false>(x1, n, *k.pw, acc.flag);
    }
    // ... and, being confined to [0.8359, 1.0088], it takes the folded 13-operation form where the kernel carries its table
    // (pow_glibc.hpp powf_folded<1>: every float of [0.7, 1.4) returns what the complete chain returns)
    const float x2 = div_nr(c1 + c2 * Lp, 1.0f + c3 * Lp);
    if constexpr (std::is_same<typename std::remove_cv<typename std::remove_reference<decltype(*k.pw)>::type>::type, PowfTablesWide>::value)
        return powf_folded<1>(x2, *k.pw);
    else
        return powf_regular<false, false, false>(x2, m, *k.pw, acc.flag);
}

// The second half of PQdec (BOUNDED / POSVAL / ELIM: see pq_decode_r below): from the first power's result Vp to L * t^(1/n), t = max(0, Vp - c1) / (c2 - c3 Vp).
// (The folded form of this power over the 2 753 141 values t can take was measured slower and is not used: pow_glibc.hpp.)
template <bool BOUNDED, bool POSVAL, int ELIM, typename K>
LH_DEV float pq_decode_tail(float Vp, const K &k, SlowAcc &acc)
{
    const float n = 0.1593f, c1 = 0.8359f, c2 = 18.8516f, c3 = 18.6875f;
    // std::max(0.0f, Vp - c1): with BOUNDED the difference is positive and the max is the identity
    const float num = BOUNDED ? Vp - c1 : std_max(0.0f, Vp - c1);
    const float t = div_nr(num, c2 - c3 * Vp);
    if constexpr (!BOUNDED && std::is_same<typename std::remove_cv<typename std::remove_reference<decltype(*k.pw)>::type>::type, PowfTablesWide>::value)
        // the bound on |log2| is tested once per unit on a running maximum (pow_glibc.hpp EMAX; slow_any<ELIM>)
        return k.Lmax * powf_regular<true, false, 0>(t, 1.0f / n, *k.pw, acc.flag, &acc.emax);
    else
        return k.Lmax * powf_regular<!BOUNDED, false, BOUNDED ? 0 : ELIM>(t, 1.0f / n, *k.pw, acc.flag);
}

template <bool BOUNDED, bool POSVAL = false, int ELIM = 1, typename K>
LH_DEV float pq_decode_r(float val, const K &k, SlowAcc &acc)
{
    const float m = 78.8438f;
    // POSVAL: val in [2^-21, 1] -- the folded form's range (pow_glibc.hpp powf_folded<0>, checked for every float of [2^-32, 1])
    float Vp;
    if constexpr (POSVAL && std::is_same<typename std::remove_cv<typename std::remove_reference<decltype(*k.pw)>::type>::type, PowfTablesWide>::value)
        Vp = powf_folded<0>(val, *k.pw);
    else
        Vp = powf_regular<!BOUNDED && !POSVAL, false, false>(val, 1.0f / m, *k.pw, acc.flag);
    return pq_decode_tail<BOUNDED, POSVAL, ELIM>(Vp, k, acc);
}

template <int CS>
LH_DEV void xform_fwd(float r, float g, float b, const XformConst &k, float &c0, float &c1, float &c2);

// RGB -> XYZ: src/luma_quantizer.cpp:273-290 (matrix include/luma/luma_quantizer.h:79-82)
LH_DEV void rgb_to_xyz(float R, float G, float B, float &X, float &Y, float &Z)
{
    X = clamp_xyz((0.412424f * R + 0.357579f * G) + 0.180464f * B);
    Y = clamp_xyz((0.212656f * R + 0.715158f * G) + 0.072186f * B);
    Z = clamp_xyz((0.019332f * R + 0.119193f * G) + 0.950444f * B);
}

template <>
LH_DEVS void xform_fwd<CS_XYZ>(float r, float g, float b, const XformConst &k, float &c0, float &c1, float &c2)
{
    rgb_to_xyz(r, g, b, c0, c1, c2);
}

// RGB -> Lu'v': src/luma_quantizer.cpp:291-316
template <>
LH_DEVS void xform_fwd<CS_LUV>(float r, float g, float b, const XformConst &k, float &c0, float &c1, float &c2)
{
    // A NaN among r, g, b (or an inf - inf among the products) makes X, Y and Z NaN together: the three rows have the
    // same pattern of positive coefficients.  Y is clamped NaN-propagating (clamp_xyz) and carries the NaN into c0
    // (-> code maxVal) and, through `sum`, into x, y, den, c1 and c2 (-> code maxC) exactly as in the reference.  X and Z
    // only ever reach the outputs through `sum` and x, so for them the one-instruction median clamp is enough: for
    // every non-NaN value (+-inf included) it IS std::max(std::min(v, 1e8f), 1e-4f), and what it returns for a NaN is
    // irrelevant because Y's NaN dominates everything computed from it.
    const float X = __builtin_amdgcn_fmed3f((0.412424f * r + 0.357579f * g) + 0.180464f * b, 0.0001f, 100000000.0f);
    const float Y = clamp_xyz((0.212656f * r + 0.715158f * g) + 0.072186f * b);
    const float Z = __builtin_amdgcn_fmed3f((0.019332f * r + 0.119193f * g) + 0.950444f * b, 0.0001f, 100000000.0f);
    const float sum = (X + Y) + Z;
    // X,Y,Z in [1e-4,1e8] (or NaN) after the clamp, sum in [3e-4,3e8]: div_nr is exact here
    const float rs = rcp_nr(sum);
    const float x = div_nr_r(X, sum, rs);
    const float y = div_nr_r(Y, sum, rs);
    // x,y in (0,1], x+y<=1+ulp: den = 3 - 2x + 12y in [1,15]; numerators in [1e-12,9].
    // (-2.0f*x) is exact (a power of two), so (-2x) + t rounds once: the fused form is the same float.
#ifdef LH_NO_FAST_DIV
    const float den = ((-2.0f * x) + 12.0f * y) + 3.0f;
#else
    const float den = __builtin_fmaf(-2.0f, x, 12.0f * y) + 3.0f;
#endif
    const float rd = rcp_nr(den);
    // (4x/den)*410 and (9y/den)*410 are > 0 and <= 9*410: div_255_pos is exact.
    // 4.0f*x is exact as well and scaling by 4 commutes with the rounding of the quotient (x/den >= 2e-14, far from the
    // denormals), so RN(RN(4x/den) * 410) = RN(RN(x/den) * 1640): one multiplication less.
    c0 = Y;  // >= 1e-4, or a NaN of either sign: the luminance search is told so (NONNEG)
#ifdef LH_NO_FAST_DIV
    c1 = div_255_pos(div_nr_r(4.0f * x, den, rd) * 410.f);
#else
    c1 = div_255_pos(div_nr_r(x, den, rd) * 1640.f);
#endif
    c2 = div_255_pos(div_nr_r(9.0f * y, den, rd) * 410.f);
}

template <>
LH_DEVS void xform_fwd<CS_RGB>(float r, float g, float b, const XformConst &k, float &c0, float &c1, float &c2)
{
    // src/luma_quantizer.cpp:355-367
    c0 = r;
    c1 = g;
    c2 = b;
}

template <>
LH_DEVS void xform_fwd<CS_PACK>(float r, float g, float b, const XformConst &, float &c0, float &c1, float &c2)
{
    c0 = r;
    c1 = g;
    c2 = b;
}

// RGB -> Y'CbCr (BT.2020, PQ): src/luma_quantizer.cpp:317-354
// YCODE: channel 0 leaves as t = 219 y + 16 (the numerator of src/luma_quantizer.cpp:337, y = the luma) instead of
// PQdec(t / 255): the encode kernel then takes the luminance CODE straight from threshold records built for the composite
// function t -> search(PQdec(t / 255)) (host_lut.cpp ycbcr_luma_code_host, evaluated with the host libm; lut_index.hpp
// build_thresh_index_fn) -- two powf, a division and the table search collapse into one 4-byte LDS gather.  y is >= +0 or
// NaN (a sum of non-negative products), so t is >= 16 or NaN, which is what the NONNEG form of the record search needs;
// beyond t ~ 508 and for NaN the reference's arithmetic ends in code maxVal, and so does the records' top bucket.
template <bool REGULAR, bool YCODE = false, typename K>
LH_DEV void ycbcr_fwd(float r, float g, float b, const K &k, float &c0, float &c1, float &c2, SlowAcc &slow)
{
    float R, G, B;
    if constexpr (REGULAR) {
        R = pq_encode_r<false>(floor_1e10(r), k, slow);
        G = pq_encode_r<false>(floor_1e10(g), k, slow);
        B = pq_encode_r<false>(floor_1e10(b), k, slow);
    } else {
        R = pq_encode(std_max(r, 1e-10f), k);
        G = pq_encode(std_max(g, 1e-10f), k);
        B = pq_encode(std_max(b, 1e-10f), k);
    }
    const float y = (0.2627f * R + 0.6780f * G) + 0.0593f * B;
    if constexpr (REGULAR) {
        // R, G, B in [7e-7, 2] once the three input tests passed (pq_encode_r), so y is too: 219y+16 in [16, 454];
        // 224t+128 (never -0) is finite; |B-y|, |R-y| <= ~2.1, zero or >= ~1e-13
        if constexpr (YCODE)
            c0 = 219.0f * y + 16.0f;
        else
            c0 = pq_decode_r<true>(div_255_pos(219.0f * y + 16.0f), k, slow);
        c1 = div_255_pos(224.0f * div_nr_r(B - y, 1.8814f, k.r18814) + 128.0f);
        c2 = div_255_pos(224.0f * div_nr_r(R - y, 1.4746f, k.r14746) + 128.0f);
    } else {
        if constexpr (YCODE)
            c0 = 219.0f * y + 16.0f;
        else
            c0 = pq_decode(div_ieee(219.0f * y + 16.0f, 255.0f), k);
        c1 = div_ieee(224.0f * div_ieee(B - y, 1.8814f) + 128.0f, 255.0f);
        c2 = div_ieee(224.0f * div_ieee(R - y, 1.4746f) + 128.0f, 255.0f);
    }
}

// N pixels at once: straight-line evaluation first; if any of them had a NaN / inf / denormal / out-of-range power
// argument (rare), all N are redone with the complete powf.  Both produce identical bits wherever the straight-line
// form applies.  One flag and one branch per thread and unit, not per pixel.
template <int N, bool YCODE = false, typename K>
LH_DEV void ycbcr_fwd_n(const float (&r)[N], const float (&g)[N], const float (&b)[N], const K &k, float (&c0)[N],
                        float (&c1)[N], float (&c2)[N])
{
    SlowAcc slow;
    slow.flag = !(k.Lmax >= 1e-6f && k.Lmax <= 1e9f);
#pragma unroll
    for (int i = 0; i < N; i++)
        ycbcr_fwd<true, YCODE>(r[i], g[i], b[i], k, c0[i], c1[i], c2[i], slow);
    if (__builtin_expect(slow_any(slow, k), 0)) {
        for (int i = 0; i < N; i++)  // not unrolled: cold code
            ycbcr_fwd<false, YCODE>(r[i], g[i], b[i], k, c0[i], c1[i], c2[i], slow);
    }
}

// ---------------------------------------------------------------------------------------------------
// Half-input table (YCbCr encode).  The reference's EXR reader delivers binary16 values only (src/exr_interface.cpp:77-146
// reads Imf::Rgba and widens), and for such an input x the non-linear value
//     R' = PQenc(std::max(x * sc, 1e-10f))                                     (src/luma_quantizer.cpp:331-333, 491-494)
// is a function of x's 16 bits alone once the stream's preScaling `sc` and peak Lmax are fixed.  The host tabulates it with
// its libm -- the function the reference calls -- for the 31745 halves +0 ... +inf (host_lut.cpp ycbcr_half_table_host;
// HALF_TABLE_LEN entries, 124 KiB, staged in LDS), and the six powf of a pixel's three channels become three LDS gathers:
//   * index = the binary16 bit pattern of x, sign-extended and clamped to [0, 0x7C00]: every negative half (-0 and -inf
//     included) reads entry 0, which is right because x * sc <= -0 < 1e-10 for the sc > 0 the table is built for, so the
//     reference's std::max answers 1e-10 exactly as it does for x = +0;
//   * `miss` is raised when x is not a half (the round trip through binary16 changes it: any rounding, overflow to inf, or a
//     NaN of either sign, which compares unequal to everything): the caller then evaluates the pixel's unit with the general
//     functions.  -0 == -0 and inf == inf compare equal, so those stay on the table.
// ---------------------------------------------------------------------------------------------------
constexpr int HALF_TABLE_LEN = 0x7C00 + 1;

LH_DEV float half_lookup(float x, const float *tab, bool &miss)
{
    const _Float16 h = (_Float16)x;  // v_cvt_f16_f32, round to nearest even
    miss = miss || ((float)h != x);
    // (bits << 16) >> 14 (arithmetic) = 4 * the sign-extended pattern: the byte offset, clamped in one v_med3_i32
    const int off = (int)((uint32_t)__builtin_bit_cast(unsigned short, h) << 16) >> 14;
    const int o = min(max(off, 0), 4 * 0x7C00);
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(tab) + o);
}

// The unit's pixels through the half table; c0 is always the YCODE form t = 219 y + 16 (see ycbcr_fwd).  r, g, b are the RAW
// inputs (before `* sc`, which the table has folded in).  Every table entry is either NaN or lies in [7e-7, 2] -- the host
// checks that before it hands the table out -- which is the operand range ycbcr_fwd<REGULAR> licenses its short divisions
// for (and a NaN gives NaN through either form of division: code maxVal / maxC as in the reference).
// Returns whether this thread's unit took the general path.
template <int N, typename K>
LH_DEV bool ycbcr_fwd_half_n(const float (&r)[N], const float (&g)[N], const float (&b)[N], const K &k, const float *tab,
                             float (&c0)[N], float (&c1)[N], float (&c2)[N])
{
    bool miss = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float R = half_lookup(r[i], tab, miss);
        const float G = half_lookup(g[i], tab, miss);
        const float B = half_lookup(b[i], tab, miss);
        const float y = (0.2627f * R + 0.6780f * G) + 0.0593f * B;
        c0[i] = 219.0f * y + 16.0f;
        c1[i] = div_255_pos(224.0f * div_nr_r(B - y, 1.8814f, k.r18814) + 128.0f);
        c2[i] = div_255_pos(224.0f * div_nr_r(R - y, 1.4746f, k.r14746) + 128.0f);
    }
    if (__builtin_expect(miss, 0)) {
        // some input of this thread's unit is not a half: the whole unit again through the general functions
        float rs[N], gs[N], bs[N];
        for (int i = 0; i < N; i++) {
            rs[i] = r[i] * k.sc;
            gs[i] = g[i] * k.sc;
            bs[i] = b[i] * k.sc;
        }
        ycbcr_fwd_n<N, true>(rs, gs, bs, k, c0, c1, c2);
    }
    return miss;
}

template <>
LH_DEVS void xform_fwd<CS_YCBCR>(float r, float g, float b, const XformConst &k, float &c0, float &c1, float &c2)
{
    const float ri[1] = {r}, gi[1] = {g}, bi[1] = {b};
    float o0[1], o1[1], o2[1];
    ycbcr_fwd_n<1>(ri, gi, bi, k, o0, o1, o2);
    c0 = o0[0];
    c1 = o1[0];
    c2 = o2[0];
}

template <int CS>
LH_DEV void xform_inv(float c0, float c1, float c2, const XformConst &k, float &r, float &g, float &b);

template <>
LH_DEVS void xform_inv<CS_PACK>(float c0, float c1, float c2, const XformConst &, float &r, float &g, float &b)
{
    r = c0;
    g = c1;
    b = c2;
}

// Y'CbCr -> RGB: src/luma_quantizer.cpp:436-473
// YT: c0 is already y = (255 PQenc(table value) - 16) / 219, read from the per-stream table the host built with its libm
// (host_lut.cpp ycbcr_ytab_host; QuantDev::ytab): the first of the four PQ evaluations of a pixel depends on the luminance
// CODE alone, so it is done once per table entry and stream instead of once per pixel.
template <bool REGULAR, bool YT = false, int ELIM = 1, bool CT = false, typename K>
LH_DEV void ycbcr_inv(float c0, float c1, float c2, const K &k, float &r, float &g, float &b, SlowAcc &slow)
{
    float y, blue, red, green;
    if constexpr (REGULAR) {
        // c0 is a table value in {0} u [1e-6, Lmax]; c1, c2 in [1e-10, ~260]: 255y-16 is finite and never -0,
        // the other numerators are zero or of magnitude >= ~1e-8 and <= ~1e5
        if constexpr (YT)
            y = c0;
        else
            y = div_219_fin(255.0f * pq_encode_r<true>(c0, k, slow) - 16.0f);
        if constexpr (CT) {   // c1, c2 are already the chroma TERMS, read from the per-code tables (ycbcr_chroma_term)
            blue = y + c1;
            red = y + c2;
        } else {
            blue = y + div_nr_r(1.8814f * (255.0f * c1 - 128.0f), 224.0f, k.r224);
            red = y + div_nr_r(1.4746f * (255.0f * c2 - 128.0f), 224.0f, k.r224);
        }
        green = div_nr_r((y - 0.2627f * red) - 0.0593f * blue, 0.6780f, k.r0678);
    } else {
        if constexpr (YT)
            y = c0;
        else
            y = div_ieee(255.0f * pq_encode(c0, k) - 16.0f, 219.0f);
        blue = y + div_ieee(1.8814f * (255.0f * c1 - 128.0f), 224.0f);
        red = y + div_ieee(1.4746f * (255.0f * c2 - 128.0f), 224.0f);
        green = div_ieee((y - 0.2627f * red) - 0.0593f * blue, 0.6780f);
    }
    if constexpr (REGULAR) {
        // The reference clamps to [0, 1] (src/luma_quantizer.cpp:460-462) and evaluates PQdec.  Every value below
        // c1^m = 0.8359^78.8438 = 7.3e-7 decodes to exactly 0: Vp = val^(1/m) <= c1, so std::max(0, Vp - c1) = 0, the quotient is 0
        // and L * 0^(1/n) = 0.  Clamping to [2^-21, 1] instead (2^-21 = 4.8e-7: Vp = 0.8314 < c1) therefore changes no result
        // and hands the first power a positive normal argument in every case: no zero select, no lower-bound test for green
        // (a difference of differences with no short proof of one).
        // One v_med3_f32 per channel (round 6) instead of v_min + v_max (where a NaN became 1, as in the reference's std::min /
        // std::max -- the complete functions below still do exactly that): the median of {v, 2^-21, 1} IS that clamp for every
        // non-NaN v, and a NaN cannot reach this line unflagged -- y is a finite table value or the unit is flagged (c0), the
        // chroma terms are finite for codes <= maxC and larger codes flag the unit (ycbcr_inv_n) -- so what the median makes of
        // one is discarded with the rest of the straight-line results.
        red = __builtin_amdgcn_fmed3f(red, 0x1p-21f, 1.0f);
        green = __builtin_amdgcn_fmed3f(green, 0x1p-21f, 1.0f);
        blue = __builtin_amdgcn_fmed3f(blue, 0x1p-21f, 1.0f);
        r = pq_decode_r<false, true, ELIM>(red, k, slow);
        g = pq_decode_r<false, true, ELIM>(green, k, slow);
        b = pq_decode_r<false, true, ELIM>(blue, k, slow);
    } else {
        red = std_max(0.0f, std_min(1.0f, red));
        green = std_max(0.0f, std_min(1.0f, green));
        blue = std_max(0.0f, std_min(1.0f, blue));
        r = pq_decode(red, k);
        g = pq_decode(green, k);
        b = pq_decode(blue, k);
    }
}

// colour-channel dequantizer: std::max(val/maxC, 1e-10f), src/luma_quantizer.cpp:261 (dequantize_color further down is the same)
LH_DEV float dequantize_color_ieee(int code, float maxC) { return std_max(div_ieee((float)code, maxC), 1e-10f); }

// The chroma term of a colour code: what src/luma_quantizer.cpp:261 (dequantize) and :450-451 make of it before y is added,
//   Cb: 1.8814 (255 c - 128) / 224,  Cr: 1.4746 (255 c - 128) / 224,  c = std::max(code / maxC, 1e-10f)
// with IEEE division.  The YCbCr decode kernels keep it per code in LDS (stage_tables, STAGE_CT): two dequantisations, four
// products / differences and two divisions per 2x2 quad become two LDS reads.
LH_DEV float ycbcr_chroma_term(int code, float maxC, float coef)
{
    const float c = std_max(div_ieee((float)code, maxC), 1e-10f);
    return div_ieee(coef * (255.0f * c - 128.0f), 224.0f);
}

// The reference's final `/ sc` (src/luma_quantizer.cpp:466-468) is part of this function since round 4 (xform_inv's other
// colour spaces leave it to the caller).  SCDIV is how the STRAIGHT-LINE code treats it -- a template choice, because a run-time
// choice between "nothing", five fused multiply-adds and an IEEE division is turned into all three plus selects by the compiler:
//   -1  never divide: the values are the reference's before that division (xform_inv's contract; k_transform divides itself);
//    0  no division in the straight-line code: right for sc == 1 (k.sc_mode 0); with k.sc_mode 2 (sc outside [2^-16, 2^16],
//       or the LH_NO_FAST_DIV build) the unit is sent to the complete functions below, which divide with IEEE division;
//    1  k.sc_mode == 1: sc is a normal float in [2^-16, 2^16].  A decoded value is 0 or L * P, and the second power of PQdec
//       sends the unit to the complete functions unless P lies in [2^-56, 2^56] (ELIM = 56 instead of powf's own 126: the
//       same compare, another constant; what it newly turns away are values below 1e-17 of the peak), L in [2^-20, 2^30]: operands
//       and quotients are normal and their exponents less than 96 apart, so div_nr_r on the refined reciprocal of sc IS the
//       IEEE quotient (see div_nr) -- five instructions instead of eleven; 0 / sc = +0 either way.
// CT: c1 / c2 are the chroma TERMS of the two colour codes (ycbcr_chroma_term, read from LDS by the caller) and code1 / code2
//     the codes themselves, which only the complete functions look at; bad_code = some code exceeds maxC (the tables end there).
// element i of a small register array, i known only at run time (cold code): a chain of selects, so that the array stays in registers
template <int M>
LH_DEV int pick(const int (&a)[M], int i)
{
    int v = a[0];
#pragma unroll
    for (int j = 1; j < M; j++)
        v = (i == j) ? a[j] : v;
    return v;
}

// NC / SUB: code1 / code2 hold one entry per 2x2 quad of the unit (SUB, N = 2 * VW pixels in two rows of VW) or one per pixel
template <int N, bool YT = false, int SCDIV = 0, bool CT = false, int NC = 1, bool SUB = false, typename K>
LH_DEV void ycbcr_inv_n(const float (&c0)[N], const float (&c1)[N], const float (&c2)[N], const K &k, float (&r)[N],
                        float (&g)[N], float (&b)[N], const int (&code1)[NC], const int (&code2)[NC], float maxC, bool bad_code)
{
    constexpr int ELIM = SCDIV == 1 ? 56 : 1;
    SlowAcc slow;
    slow.flag = !(k.Lmax >= 1e-6f && k.Lmax <= 1e9f) || (SCDIV == 0 && k.sc_mode == 2) || (CT && bad_code);
#pragma unroll
    for (int i = 0; i < N; i++) {
        // out-of-range colour codes (c1, c2 > 1) or a non-finite table value leave the licensed ranges: slow path
        // (YT: the y table exists only for tables whose entries are all finite and non-negative, lumahip_core.hip)
        if constexpr (CT)
            ;   // (bad_code says it)
        else if constexpr (YT)
            slow.flag = slow.flag || !(c1[i] <= 1.0f && c2[i] <= 1.0f);
        else
            slow.flag = slow.flag || !(c1[i] <= 1.0f && c2[i] <= 1.0f && c0[i] <= 3.0e38f);
        ycbcr_inv<true, YT, ELIM, CT>(c0[i], c1[i], c2[i], k, r[i], g[i], b[i], slow);
    }
    if constexpr (SCDIV == 1) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            r[i] = div_nr_r(r[i], k.sc, k.rsc);
            g[i] = div_nr_r(g[i], k.sc, k.rsc);
            b[i] = div_nr_r(b[i], k.sc, k.rsc);
        }
    }
    if (__builtin_expect(slow_any<ELIM>(slow, k), 0)) {
        for (int i = 0; i < N; i++) {  // not unrolled: cold code
            float d1 = c1[i], d2 = c2[i];
            if constexpr (CT) {
                const int j = SUB ? (i % (N / 2)) / 2 : i;
                d1 = dequantize_color_ieee(pick(code1, j), maxC);
                d2 = dequantize_color_ieee(pick(code2, j), maxC);
            }
            ycbcr_inv<false, YT>(c0[i], d1, d2, k, r[i], g[i], b[i], slow);
            if (SCDIV >= 0) {
                r[i] = div_ieee(r[i], k.sc);   // (x / 1.0f == x)
                g[i] = div_ieee(g[i], k.sc);
                b[i] = div_ieee(b[i], k.sc);
            }
        }
    }
}

// The same with red and blue NOT evaluated in the straight-line code: the decode kernels with the per-stream (Y', Cb) -> blue and
// (Y', Cr) -> red tables (luma_kernels.hpp, RB) read those two channels from global memory and compute green only -- it depends on
// all three codes.  c0 = y (the y table), c1 / c2 = the chroma TERMS (CT form of ycbcr_inv).  Returns true when the unit took
// the complete functions (bad code, or green's arguments left the licensed ranges); r, g, b then hold all three results and the
// caller drops what it gathered.
template <int N, int SCDIV, int NC, bool SUB, typename K>
LH_DEV bool ycbcr_inv_green_n(const float (&c0)[N], const float (&c1)[N], const float (&c2)[N], const K &k, float (&r)[N],
                              float (&g)[N], float (&b)[N], const int (&code1)[NC], const int (&code2)[NC], float maxC, bool bad_code)
{
    constexpr int ELIM = SCDIV == 1 ? 56 : 1;
    SlowAcc slow;
    slow.flag = !(k.Lmax >= 1e-6f && k.Lmax <= 1e9f) || (SCDIV == 0 && k.sc_mode == 2) || bad_code;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float y = c0[i];
        const float blue = y + c1[i], red = y + c2[i];
        float green = div_nr_r((y - 0.2627f * red) - 0.0593f * blue, 0.6780f, k.r0678);
        green = __builtin_amdgcn_fmed3f(green, 0x1p-21f, 1.0f);   // (see ycbcr_inv on the lower bound and on the median)
        g[i] = pq_decode_r<false, true, ELIM>(green, k, slow);
        if constexpr (SCDIV == 1)
            g[i] = div_nr_r(g[i], k.sc, k.rsc);
    }
    const bool redo = slow_any<ELIM>(slow, k);
    if (__builtin_expect(redo, 0)) {
        for (int i = 0; i < N; i++) {  // not unrolled: cold code
            const int j = SUB ? (i % (N / 2)) / 2 : i;
            const float d1 = dequantize_color_ieee(pick(code1, j), maxC), d2 = dequantize_color_ieee(pick(code2, j), maxC);
            ycbcr_inv<false, true>(c0[i], d1, d2, k, r[i], g[i], b[i], slow);
            r[i] = div_ieee(r[i], k.sc);   // (x / 1.0f == x)
            g[i] = div_ieee(g[i], k.sc);
            b[i] = div_ieee(b[i], k.sc);
        }
    }
    return redo;
}

// One entry of the per-stream red / blue tables: what the reference makes of luminance code `ycode` and colour code `ccode`
// (src/luma_quantizer.cpp:253-261 dequantize, :447-451 y and the chroma term, :460-468 clamp, PQdec, / sc) with the complete
// functions and IEEE division throughout -- the arithmetic every straight-line form above is checked against.
// coef = 1.8814f (blue from Cb) or 1.4746f (red from Cr); yval = the y table's entry of the luminance code.
template <typename K>
LH_DEV float ycbcr_rb_entry(float yval, int ccode, float maxC, float coef, const K &k)
{
    const float v = yval + ycbcr_chroma_term(ccode, maxC, coef);
    const float cl = std_max(0.0f, std_min(1.0f, v));
    return div_ieee(pq_decode(cl, k), k.sc);
}

template <int N, bool YT = false, int SCDIV = 0, typename K>
LH_DEV void ycbcr_inv_n(const float (&c0)[N], const float (&c1)[N], const float (&c2)[N], const K &k, float (&r)[N],
                        float (&g)[N], float (&b)[N])
{
    const int none[1] = {0};
    ycbcr_inv_n<N, YT, SCDIV, false, 1, false>(c0, c1, c2, k, r, g, b, none, none, 0.0f, false);
}

template <>
LH_DEVS void xform_inv<CS_YCBCR>(float c0, float c1, float c2, const XformConst &k, float &r, float &g, float &b)
{
    const float i0[1] = {c0}, i1[1] = {c1}, i2[1] = {c2};
    float ro[1], go[1], bo[1];
    ycbcr_inv_n<1, false, -1>(i0, i1, i2, k, ro, go, bo);
    r = ro[0];
    g = go[0];
    b = bo[0];
}

// XYZ -> RGB (before the final /sc): src/luma_quantizer.cpp:378-395 (matrix include/luma/luma_quantizer.h:84-87)
LH_DEV void xyz_to_rgb(float X, float Y, float Z, float &r, float &g, float &b)
{
    r = (3.240708f * X + -1.537259f * Y) + -0.498570f * Z;
    g = (-0.969257f * X + 1.875995f * Y) + 0.041555f * Z;
    b = (0.055636f * X + -0.203996f * Y) + 1.057069f * Z;
}

template <>
LH_DEVS void xform_inv<CS_XYZ>(float c0, float c1, float c2, const XformConst &k, float &r, float &g, float &b)
{
    xyz_to_rgb(c0, c1, c2, r, g, b);
}

// Lu'v' -> RGB: src/luma_quantizer.cpp:396-421.  The chroma-only part (everything that does not involve L)
// is split out so that a 4:2:0 decoder evaluates it once per 2x2 quad:
//   u = c1*255/410, v = c2*255/410, d = 6u - 16v + 12, x = 9u/d, y = 4v/d,  xy = x/y,  zy = (1-x-y)/y.
// SAFE = both colour codes are <= maxC, so c1,c2 in [1e-10,1], u,v in [6e-11,0.63], d in [2.05,15.8],
// x in [3e-11,2.8], y in [1.6e-11,1.3], |1-x-y| either 0 or >= ~1e-15: every operand and quotient is far
// from the fp32 exponent limits and div_nr is bit-identical to IEEE division (see div_nr).  Otherwise
// (out-of-range codes from a lossy upstream decoder: d may be <= 0, quotients may overflow) plain IEEE '/'.
struct LuvChroma {
    float xy, zy;
};

template <bool SAFE>
LH_DEV LuvChroma luv_chroma(float c1, float c2)
{
    LuvChroma o;
    if constexpr (SAFE) {
        const float r410 = rcp_nr(410.0f);
        const float u = div_nr_r(c1 * 255.0f, 410.0f, r410);
        const float v = div_nr_r(c2 * 255.0f, 410.0f, r410);
        const float d = ((6.0f * u) - 16.0f * v) + 12.0f;
        const float rd = rcp_nr(d);
        const float x = div_nr_r(9.0f * u, d, rd);
        const float y = div_nr_r(4.0f * v, d, rd);
        const float ry = rcp_nr(y);
        o.xy = div_nr_r(x, y, ry);
        o.zy = div_nr_r((1.0f - x) - y, y, ry);
    } else {
        const float u = div_ieee(c1 * 255.0f, 410.0f);
        const float v = div_ieee(c2 * 255.0f, 410.0f);
        const float d = ((6.0f * u) - 16.0f * v) + 12.0f;
        const float x = div_ieee(9.0f * u, d);
        const float y = div_ieee(4.0f * v, d);
        o.xy = div_ieee(x, y);
        o.zy = div_ieee((1.0f - x) - y, y);
    }
    return o;
}

// The same factors from u = uv_table[code1], v = uv_table[code2], where the decode kernels' LDS table holds
// (max(code/maxC, 1e-10) * 255) / 410 for every code 0..maxC, computed with IEEE division when the table is staged
// (stage_tables): the two dequantisations and the two divisions by 410 become two LDS reads.  u, v in [6e-11, 0.63] as in
// luv_chroma<true>, so the remaining divisions take the short path under the same licence.
LH_DEV LuvChroma luv_chroma_uv(float u, float v)
{
    LuvChroma o;
    const float d = ((6.0f * u) - 16.0f * v) + 12.0f;
    const float rd = rcp_nr(d);
    const float x = div_nr_r(9.0f * u, d, rd);
    const float y = div_nr_r(4.0f * v, d, rd);
    const float ry = rcp_nr(y);
    o.xy = div_nr_r(x, y, ry);
    o.zy = div_nr_r((1.0f - x) - y, y, ry);
    return o;
}

LH_DEV float uv_table_entry(int code, float maxC)
{
    return div_ieee(std_max(div_ieee((float)code, maxC), 1e-10f) * 255.0f, 410.0f);  // src/luma_quantizer.cpp:261,405-406
}

LH_DEV void luv_apply(float L, const LuvChroma &q, float &r, float &g, float &b)
{
    const float Y = clamp_xyz(L);
    const float X = clamp_xyz(q.xy * L);
    const float Z = clamp_xyz(q.zy * L);
    xyz_to_rgb(X, Y, Z, r, g, b);
}

template <>
LH_DEVS void xform_inv<CS_LUV>(float c0, float c1, float c2, const XformConst &k, float &r, float &g, float &b)
{
    luv_apply(c0, luv_chroma<false>(c1, c2), r, g, b);
}

template <>
LH_DEVS void xform_inv<CS_RGB>(float c0, float c1, float c2, const XformConst &k, float &r, float &g, float &b)
{
    // src/luma_quantizer.cpp:422-435
    r = c0;
    g = c1;
    b = c2;
}

// ---------------------------------------------------------------------------------------------------
// Quantizer description handed to every kernel by value.
// ---------------------------------------------------------------------------------------------------
struct QuantDev {
    const float *lut;        // global: maxVal+1 floats followed by `pad` NaNs
    const uint32_t *rec;     // global: nbuckets threshold records (modes 3, 4, 5; lut_index.hpp)
    const float *ytab;       // global, nullable: YCbCr decode, y = (255 PQenc(lut[i]) - 16) / 219 per table entry (host_lut.cpp)
    int lut_len;             // maxVal + 1
    int pad;
    int maxVal;
    int mode;                // lh::LutMode
    int shift, kmin, nbuckets;  // threshold records: key = bits >> shift, clamped to [kmin, kmin+nbuckets-1]
    float kscale;            // value-keyed records (mode 7, lut_index.hpp LinIndex): key = cvt_u32(min(v * kscale, nbuckets - 1))
    float maxC;              // (float)m_maxValColor
    int cs;
    float Lmax;
};

// colour-channel quantizer: floor(maxC*val + 0.5f) clamped with std::min / std::max, src/luma_quantizer.cpp:238-241.
// With t = maxC*val + 0.5f the reference returns max(0, min(maxC, floor(t))).  std::min(maxC, res) = (res < maxC) ? res :
// maxC returns maxC when res is NaN -- IEEE minNum (v_min_f32); the result is then non-NaN and std::max(0.0f, .) is
// maxNum.  Clamping BEFORE taking the integer part gives the same integer without the floor instruction: t >= maxC ->
// maxC either way (maxC is an integer); 0 <= t < maxC -> the float -> int conversion truncates, which is floor for a
// non-negative value; t < 0 -> floor(t) <= -1 is raised to 0, and so is t itself; NaN -> maxC by the min.
// NONNEG: the caller guarantees val >= 0 or NaN (Lu'v' chroma), so t >= 0.5 and the max(0, .) is the identity.
template <bool NONNEG = false>
LH_DEV int quantize_color_t(float t, float maxC)
{
#ifdef LH_NO_FAST_DIV  // the literal form, for the A/B build
    float res = floorf(t);
    res = __builtin_fminf(res, maxC);
    res = __builtin_fmaxf(res, 0.0f);
    return (int)res;
#else
    float res = __builtin_fminf(t, maxC);
    if (!NONNEG)
        res = __builtin_fmaxf(res, 0.0f);
    return (int)res;
#endif
}

template <bool NONNEG = false>
LH_DEV int quantize_color(float val, float maxC)
{
    return quantize_color_t<NONNEG>(maxC * val + 0.5f, maxC);
}

// The 4:2:0 chroma sample: quantize_color(0.25f * s4, maxC) with s4 the sum of the four pixels
// (src/luma_encoder.cpp:287-289 followed by src/luma_quantizer.cpp:238-241).  0.25f * s4 is exact (a power of two; s4 is
// zero or far above the denormals) and so is qc = 0.25f * maxC (maxC = 2^n - 1, n <= 16), hence
// RN(maxC * (0.25 * s4)) = RN(qc * s4): one multiplication instead of two.
template <bool NONNEG = false>
LH_DEV int quantize_color_sum4(float s4, float maxC, float qc)
{
#ifdef LH_NO_FAST_DIV
    return quantize_color<NONNEG>(0.25f * s4, maxC);
#else
    return quantize_color_t<NONNEG>(qc * s4 + 0.5f, maxC);
#endif
}

// colour-channel dequantizer: std::max(val/maxC, 1e-10f), src/luma_quantizer.cpp:261
LH_DEV float dequantize_color(int code, float maxC) { return std_max(div_ieee((float)code, maxC), 1e-10f); }
// code <= maxC (both < 2^16, maxC >= 1): operands and quotient in [0, 65535] -> div_nr is exact
LH_DEV float dequantize_color_safe(int code, float maxC, float rmaxC)
{
    return std_max(div_nr_r((float)code, maxC, rmaxC), 1e-10f);
}

// LUT-channel dequantizer, src/luma_quantizer.cpp:253-258 (code is an unsigned sample, so val<0 never holds)
template <typename LutPtr>
LH_DEV float dequantize_lut(int code, LutPtr lut, int maxVal)
{
    return lut[code < maxVal ? code : maxVal];
}

// ---------------------------------------------------------------------------------------------------
// LUT-channel quantizer for N values at once.
// ---------------------------------------------------------------------------------------------------

// the reference's loop, literally: src/luma_quantizer.cpp:222-235
template <typename LutPtr>
LH_DEV int quantize_lut_literal(float v, LutPtr lut, int maxVal)
{
    int l = 0, r = maxVal;
    while (l + 1 < r) {
        const int m = (l + r) >> 1;
        if (v < lut[m])
            r = m;
        else
            l = m;
    }
    return ((v - lut[l]) < (lut[r] - v)) ? l : r;
}

LH_DEV float lds_f32(const float *lut, int byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(lut) + byte_off);
}

// signed median of three (lo <= hi): clamp in one VALU op
LH_DEV int med3_i32(int x, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "s"(hi));  // one SGPR per VOP3 (constant bus)
    return r;
}

// unsigned median of three (lo <= hi)
LH_DEV uint32_t med3_u32(uint32_t x, uint32_t lo, uint32_t hi)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "s"(hi));
    return r;
}

// Threshold records (lut_index.hpp, "Threshold records"): code = (rec[clamped key] + low bits of v) >> shift --
// one 4-byte gather, no table probes, no subtractions; equal to the reference's bisection + nearest-of-two for
// every float by construction of the records (host) and by the exhaustive sweeps of tests/test_gpu_exhaustive.py.
// `rec` points at the record of key kmin >= 0 (LDS copy or global).
// NONNEG = false: any float.  The key is the SIGNED bit pattern >> shift, so negative values (and sign-set NaNs) clamp
//          to the bottom bucket; the reference answers maxVal for every NaN, hence the explicit NaN test.
// NONNEG = true: the caller guarantees v >= 0 or v is a NaN of EITHER sign (the Lu'v' luminance: clamped to >= 1e-4
//          or NaN).  The key is the UNSIGNED bit pattern >> shift: every NaN exceeds every finite key and lands in the
//          top bucket (start maxVal, no threshold), no NaN test needed.
template <int N, bool NONNEG, typename RecPtr>
LH_DEV void quantize_thresh(const float (&v)[N], int (&code)[N], RecPtr rec, const QuantDev &q)
{
    const auto biased = rec - q.kmin;
    const int khi = q.kmin + q.nbuckets - 1;
    const uint32_t lowmask = (1u << q.shift) - 1u;  // shift <= 23
    uint32_t r[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (NONNEG)
            r[i] = biased[med3_u32(__float_as_uint(v[i]) >> q.shift, (uint32_t)q.kmin, (uint32_t)khi)];
        else
            r[i] = biased[med3_i32(__float_as_int(v[i]) >> q.shift, q.kmin, khi)];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int c = (int)((r[i] + (__float_as_uint(v[i]) & lowmask)) >> q.shift);
        if constexpr (NONNEG)
            code[i] = c;
        else
            code[i] = (v[i] != v[i]) ? q.maxVal : c;
    }
}

// Value-keyed records (lut_index.hpp LinIndex; evenly spaced tables, PTF_LINEAR): rec[key] = {P, start}, P = bits(T) - 1,
// code = start + ((int)bits(v) > (int)P), key = cvt_u32(min(v * kscale, nbuckets - 1)).  The product is one rounded fp32
// operation exactly as the host evaluated it when it sorted the thresholds into buckets; v_min_f32 returns the non-NaN
// operand, so a NaN (the product quiets a signalling one) lands in the top bucket, whose start is maxVal and whose P = 0x7fffffff no
// signed integer exceeds -- the reference's answer for every NaN; v_cvt_u32_f32 truncates and saturates negatives to 0,
// and in bucket 0 the SIGNED comparison is false for every negative float (sign bit set): code c0.  One 8-byte gather.
template <int N, typename RecPtr>
LH_DEV void quantize_linkey(const float (&v)[N], int (&code)[N], RecPtr rec, const QuantDev &q)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const float top = (float)(q.nbuckets - 1);
    u32x2 r[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float p = __builtin_fminf(v[i] * q.kscale, top);
        uint32_t k;
        asm("v_cvt_u32_f32 %0, %1" : "=v"(k) : "v"(p));   // (a C++ cast of a negative float is undefined; the instruction saturates)
        r[i] = *reinterpret_cast<const u32x2 *>(rec + 2 * k);
    }
#pragma unroll
    for (int i = 0; i < N; i++)
        code[i] = (int)r[i].y + (__float_as_int(v[i]) > (int)r[i].x ? 1 : 0);
}

// MODE (lut_index.hpp LutMode): 0 literal bisection (table in LDS), 2 literal bisection (table in global memory),
//       3 / 4 threshold records (LDS / global); `idx` = the records for 3 / 4, unused otherwise;
//       5 = records (LDS) of the YCbCr composite t -> code (ycbcr_fwd<., YCODE>): v is t = 219 y + 16, >= 16 or NaN;
//       7 = value-keyed records in LDS (quantize_linkey)
template <int MODE, int N, bool NONNEG = false, typename LutPtr, typename IdxPtr>
LH_DEV void quantize_lut(const float (&v)[N], int (&code)[N], LutPtr lut, IdxPtr idx, const QuantDev &q)
{
    if constexpr (MODE == 7) {
        quantize_linkey<N>(v, code, idx, q);
    } else if constexpr (MODE == 5) {
        quantize_thresh<N, true>(v, code, idx, q);
    } else if constexpr (MODE == 3 || MODE == 4) {
        quantize_thresh<N, NONNEG>(v, code, idx, q);
    } else {
#pragma unroll
        for (int i = 0; i < N; i++)
            code[i] = quantize_lut_literal(v[i], lut, q.maxVal);
    }
}

// wave64 reductions (stats)
LH_DEV float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        x += __shfl_xor(x, o, 64);
    return x;
}
LH_DEV float wave_min(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        x = fminf(x, __shfl_xor(x, o, 64));
    return x;
}
LH_DEV float wave_max(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

}  // namespace lh
