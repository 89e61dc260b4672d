// pow_glibc.hpp -- powf(x, y) with the results of glibc 2.35's libm on x86-64 (FMA ifunc variant),
// usable on the device (gfx950 fp64 VALU) and on the host.
//
// Why: the reference's YCbCr (BT.2020 + PQ) colour transform calls libm powf eight times per pixel
// (LumaQuantizer::transformPQ, src/luma_quantizer.cpp:485-501, call sites :331-337 and :447-459).  The
// arithmetic therefore lives in a third-party dependency that is not in the reference tree and that the
// reference does not pin: glibc libm (this image: 2.35-0ubuntu3.11).  Bit-exact Y/Cb/Cr planes need that
// function's exact results, not a correctly-rounded powf: glibc's powf differs from the correctly
// rounded result for ~0.05-0.2 % of arguments.
//
// What is restated: glibc's sysdeps/ieee754/flt-32/e_powf.c (Szabolcs Nagy's algorithm, also published
// in ARM optimized-routines, math/powf.c): x^y = exp2(y * log2(x)) evaluated in binary64 with a 16-entry
// log2 table + degree-5 polynomial and a 32-entry exp2 table + degree-3 polynomial, no rounding-mode or
// errno handling (WANT_ROUNDING / errno affect only flags, not values, in round-to-nearest).  x86-64
// glibc selects __powf_fma at load time on any CPU with FMA+AVX2 (every host this runs on); in that
// build each `a*b + c` of the source is one fused multiply-add.  The restatement spells those fma()s
// out explicitly, so it does not depend on compiler contraction, host CPU or GPU.
//
// Constants: __powf_log2_data (POWF_LOG2_TABLE_BITS=4, POWF_LOG2_POLY_ORDER=5, POWF_SCALE_BITS=0) and
// __exp2f_data (EXP2F_TABLE_BITS=5) as published; the 32 exp2 table words equal
// bits(RN(2^(i/32))) - (i << 47) and were cross-checked against that formula, the log2 table against
// the bytes of this image's libm.so.6.  tests/test_powf.py compares the restatement with the host
// libm powf (exhaustively for the four PQ exponents with tools/verify_powf, sampled in the suite).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LH_HD __host__ __device__ __forceinline__
#else
#define LH_HD inline
#endif

namespace lh {

struct PowfTables {
    // {invc, logc} x 16, then 32 exp2 table words reinterpreted as double bit patterns
    double log2_tab[16][2];
    uint64_t exp2_tab[32];
};

// clang-format off
#define LH_POWF_LOG2_TAB { \
  { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 }, \
  { 0x1.49539f0f010bp+0,  -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 }, \
  { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8eap+0,  -0x1.97c1d1b3b7afp-3 }, \
  { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 }, \
  { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1p+0, 0x0p+0 }, \
  { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 },  { 0x1.ca4b31f026aap-1,  0x1.476a9543891bap-3 }, \
  { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 }, \
  { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 },  { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 } }
#define LH_POWF_EXP2_TAB { \
  0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, \
  0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, \
  0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, \
  0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, \
  0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, \
  0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, \
  0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, \
  0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull }
// clang-format on

static const PowfTables kPowfTablesHost = {LH_POWF_LOG2_TAB, LH_POWF_EXP2_TAB};

// The LDS copy the kernels use adds a "wide" log2 table: entry (k & 127) * 16 + i holds {invc[i] * 2^-k, logc[i] + (double)k}
// for the binary exponents k in [-64, 63] -- the two operands log2_inline needs, so that ONE 16-byte LDS read keyed by
// bits 19..29 of (ix - OFF) replaces the 16-entry read, the arithmetic shift, the int -> double conversion and the fp64
// add `logc + (double)k` (10 issue cycles per powf on gfx950; the sum is the same single rounding, done when the table
// is filled).  Arguments outside 2^-64 .. 2^64 take the complete function (pw_range_key).
//
// Round 6 adds the tables of the FOLDED form (powf_folded below) for the two powers whose arguments live in one narrow, fully
// enumerable range -- PQdec's first power val^(1/m), val in [2^-21, 1], and PQenc's second power q^m, q in [0.7, 1.4):
//   foldA[e] = {invc[i] * 2^-k, (1/m) * (logc[i] + k)}   e = (k + 21) * 16 + i, k in [-21, 0]   (352 entries)
//   foldC[i] = {invc[i],        m * logc[i]}              (k = 0 throughout)
constexpr int FOLD_A_KMIN = -21, FOLD_A_LEN = (0 - FOLD_A_KMIN + 1) * 16;
struct PowfTablesWide : PowfTables {
    double wide[2048][2];
    double foldA[FOLD_A_LEN][2];
    double foldC[16][2];
};
static_assert(sizeof(PowfTablesWide) == sizeof(PowfTables) + 32768 + FOLD_A_LEN * 16 + 256, "layout");

// entry e of the wide table from the 16-entry table: e = (k & 127) * 16 + i
LH_HD void pw_wide_entry(int e, const double (&lt)[16][2], double &invc, double &y0)
{
    const int kk = e >> 4, i = e & 15;
    const int k = kk < 64 ? kk : kk - 128;
    // round 6: the reciprocal carries 2^-k, so that log2_inline's z = x 2^-k (an integer subtraction on the argument's exponent
    // field and the mask that finds it: two instructions per powf) need not be formed at all -- r = fma(x, invc 2^-k, -1) is the
    // same double as fma(z, invc, -1): both factors are scaled by exact powers of two, the product is the same real number
    invc = __builtin_ldexp(lt[i][0], -k);
    y0 = lt[i][1] + (double)k;  // the one rounding of log2_inline's `logc + (double) k`
}

LH_HD uint32_t pw_asuint(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
LH_HD float pw_asfloat(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
LH_HD uint64_t pw_asuint64(double f)
{
    uint64_t u;
    memcpy(&u, &f, 8);
    return u;
}
LH_HD double pw_asdouble(uint64_t u)
{
    double f;
    memcpy(&f, &u, 8);
    return f;
}

// 0: not an integer, 1: odd integer, 2: even integer (e_powf.c checkint)
LH_HD int pw_checkint(uint32_t iy)
{
    int e = iy >> 23 & 0xff;
    if (e < 0x7f)
        return 0;
    if (e > 0x7f + 23)
        return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1))
        return 0;
    if (iy & (1u << (0x7f + 23 - e)))
        return 1;
    return 2;
}

LH_HD int pw_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000 - 1; }

// Tab is anything indexable like PowfTables (host struct, LDS copy, ...)
template <typename Tab>
LH_HD float powf_glibc(float x, float y, const Tab &T)
{
    const uint32_t SIGN_BIAS = 1u << (5 + 11);
    uint32_t sign_bias = 0;
    uint32_t ix = pw_asuint(x), iy = pw_asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || pw_zeroinfnan(iy)) {
        // Either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan).
        if (pw_zeroinfnan(iy)) {
            if (2 * iy == 0)
                return 1.0f;  // (issignaling(x) ? x + y : 1) -- value 1 or NaN; a signalling x gives NaN
            if (ix == 0x3f800000u)
                return 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u)
                return x + y;
            if (2 * ix == 2 * 0x3f800000u)
                return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u))
                return 0.0f;  // |x|<1 && y==inf or |x|>1 && y==-inf
            return y * y;
        }
        if (pw_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && pw_checkint(iy) == 1)
                x2 = -x2;
            return (iy & 0x80000000u) ? (1 / x2) : x2;
        }
        // x and y are non-zero finite
        if (ix & 0x80000000u) {
            int yint = pw_checkint(iy);
            if (yint == 0)
                return (x - x) / (x - x);  // __math_invalidf: NaN
            if (yint == 1)
                sign_bias = SIGN_BIAS;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {
            // normalize subnormal x so exponent becomes negative
            ix = pw_asuint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // ---- log2_inline(ix): log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const uint32_t OFF = 0x3f330000u;
    const uint32_t tmp = ix - OFF;
    const int i = (tmp >> (23 - 4)) % 16;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;  // arithmetic shift
    const double invc = T.log2_tab[i][0], logc = T.log2_tab[i][1];
    const double z = (double)pw_asfloat(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);
    const double logx = yy;
    const double ylogx = (double)y * logx;  // cannot overflow, y is single precision
    if ((pw_asuint64(ylogx) >> 47 & 0xffff) >= (pw_asuint64(126.0) >> 47)) {
        // |y*log(x)| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6)
            return sign_bias ? -__builtin_inff() : __builtin_inff();  // __math_oflowf
        if (ylogx <= -150.0)
            return sign_bias ? -0.0f : 0.0f;  // __math_uflowf
    }
    // ---- exp2_inline(ylogx, sign_bias): x = k/N + r, r in [-1/(2N), 1/(2N)], N = 32
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    const double SHIFT = 0x1.8p+52 / 32;
    double kd = ylogx + SHIFT;
    const uint64_t ki = pw_asuint64(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = T.exp2_tab[ki % 32];
    const uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    const double s = pw_asdouble(t);
    const double zz = __builtin_fma(C0, rr, C1);
    const double rr2 = rr * rr;
    double e = __builtin_fma(C2, rr, 1.0);
    e = __builtin_fma(zz, rr2, e);
    e = e * s;
    return (float)e;
}

// log2_inline's table operands for tmp = ix - OFF: {invc, logc + (double)k}
// ... and z, the factor r = fma(z, invc, -1) is formed from: the argument with its exponent reduced (16-entry table) or the
// argument itself (wide table, whose reciprocals carry the 2^-k)
LH_HD void pw_log2_operands(const PowfTables &T, uint32_t ix, uint32_t tmp, double &invc, double &y0, double &z)
{
    const int i = (tmp >> (23 - 4)) % 16;
    const uint32_t top = tmp & 0xff800000u;
    const int k = (int32_t)top >> 23;
    invc = T.log2_tab[i][0];
    y0 = T.log2_tab[i][1] + (double)k;
    z = (double)pw_asfloat(ix - top);
}
LH_HD void pw_log2_operands(const PowfTablesWide &T, uint32_t ix, uint32_t tmp, double &invc, double &y0, double &z)
{
    z = (double)pw_asfloat(ix);
#if defined(__HIP_DEVICE_COMPILE__)
    // v_bfe_u32, then one v_lshl_add_u32 forms the LDS address; written as a builtin or as shifts the compiler turns
    // it into shift + and + add (three instructions)
    uint32_t e;
    asm("v_bfe_u32 %0, %1, 19, 11" : "=v"(e) : "v"(tmp));
#else
    const uint32_t e = (tmp >> (23 - 4)) & 2047u;
#endif
    invc = T.wide[e][0];
    y0 = T.wide[e][1];
}
// Range key of an argument's bits: powf_regular applies iff key < pw_range_limit (unsigned) -- x a positive normal
// finite float, and with the wide table also 2^-64 <= x / 0.7 < 2^64.  One unsigned subtract; callers with many
// arguments keep a running unsigned maximum of the keys and compare once.
LH_HD uint32_t pw_range_key(const PowfTables &, uint32_t ix) { return ix - 0x00800000u; }
LH_HD uint32_t pw_range_limit(const PowfTables &) { return 0x7f000000u; }
LH_HD uint32_t pw_range_key(const PowfTablesWide &, uint32_t ix) { return ix - (0x3f330000u - (64u << 23)); }
LH_HD uint32_t pw_range_limit(const PowfTablesWide &) { return 128u << 23; }
// bits of the smallest argument the straight-line form accepts (for "x is +0 or >= this" tests)
LH_HD uint32_t pw_range_low(const PowfTables &) { return 0x00800000u; }
LH_HD uint32_t pw_range_low(const PowfTablesWide &) { return 0x3f330000u - (64u << 23); }

// exp2_inline's argument split: ylogx = k/32 + rr with rr in [-1/64, 1/64], s = 2^(k/32) from the 32-entry table
template <typename Tab>
LH_HD void pw_exp2_split(double ylogx, const Tab &T, double &rr, double &s)
{
    const double SHIFT = 0x1.8p+52 / 32;
    double kd = ylogx + SHIFT;
    const uint64_t ki = pw_asuint64(kd);
    kd -= SHIFT;
    rr = ylogx - kd;
    // s = bits(tab[ki % 32] + (ki << 47)): the low 32 bits of ki << 47 are zero, so only the high word changes -- one
    // 32-bit shift-add on the device (the compiler otherwise builds the 64-bit shift and add out of five instructions)
    const uint64_t t0 = T.exp2_tab[ki % 32];
#if defined(__HIP_DEVICE_COMPILE__)
    s = __hiloint2double((int)((uint32_t)(t0 >> 32) + ((uint32_t)ki << 15)), (int)(uint32_t)t0);
#else
    s = pw_asdouble(t0 + (ki << (52 - 5)));
#endif
}

// The polynomial of exp2_inline and the final scaling.  e_powf.c evaluates 1 + C2 r + r^2 (C1 + C0 r) as two independent halves
// joined by a third fma (five operations with the scaling); this is Horner's form of the same polynomial (four).  The two differ
// in the last bits of the DOUBLE now and then, never in the float it is rounded to: for each of the four PQ exponents, every
// positive normal x with |y log2 x| < 126 -- 4 x 2^31 arguments less the out-of-range ones -- returns the same float either way
// (tools/bench/powf_variants.cpp on the host, 32 s; tests/test_gpu_exhaustive.py sweeps the device function against the host
// libm over the same arguments).  powf_glibc above keeps glibc's order and is what every exceptional pixel is redone with.
template <bool ZERO>
LH_HD float pw_exp2_tail(double rr, double s, bool zero)
{
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double e = __builtin_fma(C0, rr, C1);
    e = __builtin_fma(e, rr, C2);
    e = __builtin_fma(e, rr, 1.0);
    e = e * s;
    if (ZERO)
        return zero ? 0.0f : (float)e;
    return (float)e;
}

// Straight-line form for the arguments the PQ transforms see almost always: x positive, normal and finite (or +0
// when ZERO) with |y*log2(x)| < 126; y is one of the four positive PQ exponents.  For those arguments it performs
// exactly the arithmetic of powf_glibc above (same operations, same order) without any of its branches; for
// everything else it sets `slow` and returns garbage -- the caller then redoes the pixel with powf_glibc.
// What is tested is a template choice, because each test costs a half-rate compare (tools/bench/valu_bench.hip) and the
// call sites can often prove a test away (luma_device.hpp states the argument range at each one):
//   ZERO    x may be +0: pow(+0, y > 0) = +0 is folded in as a select (black pixels are common on the decode side);
//   CHECK_X x may be anything: raise `slow` unless it is a positive normal finite float (or +0 when ZERO);
//   CHECK_E |y*log2(x)| may reach 126 (over/underflow handling of e_powf.c): raise `slow` then.  (1 / true: at 126; any other
//           non-zero value: already at that bound -- a caller that wants the result's exponent inside a narrower range.)
// powf_regular<true, true, true> accepts every argument and is what tests/test_gpu_exhaustive.py sweeps.
// EMAX (round 6; wide table, ZERO, no CHECK_X): the |y log2 x| test as an INTEGER running maximum instead of a compare per call --
// `emax` collects the high word of |ylogx| and the caller compares it once per unit with pw_emax_limit(bound).  A compare per
// call keeps a lane mask alive per pixel and channel; with 24 of them in a unit the compiler ran out of scalar registers and
// moved them through VGPRs as 0 / 1 (about twelve extra half-rate instructions per pixel in the decode kernel).  No `!zero`
// term is needed: with the wide table an argument of +0 reads the entry of k = 1, i = 9 ({0.5, 1.0}), r = -1, and
// |y log2| comes out as 2.29 y -- 14.4 for y = 1/n, far below every bound (tools/verify_powf.cpp asserts it).
LH_HD uint32_t pw_emax_limit(int bound) { return (uint32_t)(pw_asuint64((double)bound) >> 47) << 15; }
template <bool ZERO, bool CHECK_X, int CHECK_E, typename Tab>
LH_HD float powf_regular(float x, float y, const Tab &T, bool &slow, uint32_t *emax = nullptr)
{
    const uint32_t ix = pw_asuint(x);
    bool zero = false;
    if (ZERO)
        zero = (ix == 0);
    if (CHECK_X)
        slow = slow || (!zero && (pw_range_key(T, ix) >= pw_range_limit(T)));
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const uint32_t tmp = ix - 0x3f330000u;
    double invc, y0, z;
    pw_log2_operands(T, ix, tmp, invc, y0, z);
    const double r = __builtin_fma(z, invc, -1.0);
    const double r2 = r * r;
    double yy = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    if (CHECK_E) {
        // the limit's top five mantissa bits are all the compare sees: 126 and 56 are exact in them
        const double lim = CHECK_E == 1 ? 126.0 : (double)CHECK_E;
        slow = slow || (!zero && ((pw_asuint64(ylogx) >> 47 & 0xffff) >= (pw_asuint64(lim) >> 47)));
    }
    if (emax) {
        const uint32_t a = (uint32_t)(pw_asuint64(ylogx) >> 32) & 0x7fffffffu;
        *emax = *emax > a ? *emax : a;
    }
    double rr, s;
    pw_exp2_split(ylogx, T, rr, s);
    return pw_exp2_tail<ZERO>(rr, s, zero);
}

// ---- the folded form: 13 fp64 operations instead of 17 -----------------------------------------------------------------
// Two of the four powers of a PQ pair take their argument from ONE narrow range (luma_device.hpp proves it at the call sites):
//   WHICH = 0   PQdec's first power, val^(1/m): val in [2^-21, 1] after the reference's clamp (FOLD_A_KMIN);
//   WHICH = 1   PQenc's second power, q^m: q = (c1 + c2 Lp) / (1 + c3 Lp) in [0.8359, 1.0088], inside [OFF, 2 OFF) =
//               [0.69921875, 1.3984375) where k = 0.
// Over such a range the function can be checked for EVERY argument, so any evaluation of glibc's polynomials that returns
// glibc's float for all of them is as good as glibc's own order.  This one folds the exponent into the log2 polynomial (y A0 ..
// y A4 as constants, y (logc + k) in the table: ylogx comes straight out of the last fma) and walks it in Horner's form: five
// fma from r to ylogx instead of eight operations, plus the Horner exp2 tail.  Checked: val in [2^-32, 1] (268 435 457
// arguments) and q in [OFF, 2 OFF) (8 388 608): no float differs from the exact chain's (tools/bench/powf_variants.cpp);
// tests/test_gpu_exhaustive.py runs the same sweep through the device code.  (The other two powers -- t^(1/n) over 2.6e8
// arguments, x^n over 2^31 -- do NOT survive the same treatment as functions of an arbitrary float: 6 to 10 floats differ;
// profiles/r06_ycbcr_powf_ledger.txt.)
// The SECOND power of PQdec, t^(1/n), has a small argument set too -- t = (Vp - c1) / (c2 - c3 Vp) is a function of the first
// power's float result, 2 753 141 values -- and the folded form returns glibc's float for all of them but one
// (t = 0x1.7bd282p-6, where glibc's double lands within a unit of a rounding boundary).  It was built, proven and MEASURED
// SLOWER: 13 or 14 operations in Horner's order are one long dependency chain per power, and with the exception's bookkeeping and
// a second set of folded coefficients the decode kernel ran 2.5 - 2.9 % slower than with powf_regular for that power
// (profiles/r06_ycbcr_powf_ledger.txt).  Not kept.
template <int WHICH>
LH_HD double pw_fold_y()
{
    const float m = 78.8438f;
    return WHICH == 0 ? (double)(1.0f / m) : (double)m;
}
// entry e of the folded tables from the 16-entry table
template <int WHICH>
LH_HD void pw_fold_entry(int e, const double (&lt)[16][2], double &invc, double &y0)
{
    const int i = e & 15, k = WHICH == 0 ? (e >> 4) + FOLD_A_KMIN : 0;
    // as in the wide table, the reciprocal carries 2^-k and the argument itself is the other factor of r = fma(x, invc 2^-k, -1)
    invc = __builtin_ldexp(lt[i][0], -k);
    y0 = pw_fold_y<WHICH>() * (lt[i][1] + (double)k);
}

template <int WHICH>
LH_HD float powf_folded(float x, const PowfTablesWide &T)
{
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const double Y = pw_fold_y<WHICH>();
    const uint32_t ix = pw_asuint(x);
    double invc, y0, z;
    if (WHICH == 0) {
        const uint32_t tmp = ix - (0x3f330000u + ((uint32_t)FOLD_A_KMIN << 23));   // (FOLD_A_KMIN < 0: the offset shrinks)
        z = (double)x;                                                              // (the table's reciprocals carry the 2^-k)
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t e;
        asm("v_bfe_u32 %0, %1, 19, 9" : "=v"(e) : "v"(tmp));
#else
        const uint32_t e = (tmp >> 19) & 511u;
#endif
        invc = T.foldA[e][0];
        y0 = T.foldA[e][1];
    } else {
        const uint32_t e = ((ix - 0x3f330000u) >> 19) & 15u;
        z = (double)x;
        invc = T.foldC[e][0];
        y0 = T.foldC[e][1];
    }
    const double r = __builtin_fma(z, invc, -1.0);
    // (Estrin's grouping of the same folded polynomial -- one operation more, a shorter chain -- measured 1 % slower; glibc's
    //  grouping of the exp2 tail 2 % slower: profiles/r06_ycbcr_powf_ledger.txt)
    double h = __builtin_fma(Y * A0, r, Y * A1);
    h = __builtin_fma(h, r, Y * A2);
    h = __builtin_fma(h, r, Y * A3);
    h = __builtin_fma(h, r, Y * A4);
    const double ylogx = __builtin_fma(h, r, y0);
    double rr, s;
    pw_exp2_split(ylogx, T, rr, s);
    return pw_exp2_tail<false>(rr, s, false);
}

}  // namespace lh
