// tools/bench/powf_variants.cpp -- VERDICT r05 item 3: can the fp64 chain of the YCbCr kernels' powf be shortened without losing
// bit-equality with glibc's powf?  Host-side, exhaustive over the argument sets the kernels prove for their powers
// (luma_device.hpp pq_encode_r / pq_decode_r); the table of profiles/r06_ycbcr_powf_ledger.txt section 1 is this program's output.
//     A  PQdec 1st power  val^(1/m), val in [2^-21, 1]        C  PQenc 2nd power  q^m, q in [0.8359, 1.0088]
//     B  PQdec 2nd power  t^(1/n),   t in [3e-9, 6.4]          E  PQenc 1st power  x^n, x in [2^-64, 2^40]
//     B* PQdec 2nd power over the t it can be handed: t = (Vp - c1) / (c2 - c3 Vp) for every float Vp of (c1, 1]
// Each candidate evaluates the same real-valued function as e_powf.c with fewer operations (the number in brackets: fp64
// operations of the chain, 17 in glibc's order); what counts is whether the FLOAT it returns equals the exact chain's for every
// argument.  What the kernels use is checked against the host libm itself by tools/verify_powf.cpp.
//
//   g++ -O2 -ffp-contract=off -std=c++17 -pthread -o /tmp/powf_variants tools/bench/powf_variants.cpp && /tmp/powf_variants   (~70 s, 8 threads)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../lumahdrv_amd/csrc/pow_glibc.hpp"
using namespace lh;
static const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
static const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
static const PowfTables T = kPowfTablesHost;
#define FMA __builtin_fma
struct Red { double r, y0; };
static inline Red reduce(float x) {
    const uint32_t ix = pw_asuint(x), tmp = ix - 0x3f330000u; const int i = (tmp >> 19) % 16; const uint32_t top = tmp & 0xff800000u;
    const int k = (int32_t)top >> 23; const double z = (double)pw_asfloat(ix - top);
    return {FMA(z, T.log2_tab[i][0], -1.0), T.log2_tab[i][1] + (double)k};
}
// log stage variants -> ylogx
static inline double log_exact(float x, double Y) { Red q = reduce(x); double r=q.r, r2=r*r, r4=r2*r2; double yy=FMA(A0,r,A1), p=FMA(A2,r,A3), t=FMA(A4,r,q.y0); t=FMA(p,r2,t); return Y*FMA(yy,r4,t); }
static inline double log_foldy(float x, double Y) { Red q = reduce(x); double r=q.r, r2=r*r, r4=r2*r2; double yy=FMA(Y*A0,r,Y*A1), p=FMA(Y*A2,r,Y*A3), t=FMA(Y*A4,r,Y*q.y0); t=FMA(p,r2,t); return FMA(yy,r4,t); }
static inline double log_horner(float x, double Y) { Red q = reduce(x); double r=q.r; double h=FMA(A0,r,A1); h=FMA(h,r,A2); h=FMA(h,r,A3); h=FMA(h,r,A4); return Y*FMA(h,r,q.y0); }
static inline double log_horner_foldy(float x, double Y) { Red q = reduce(x); double r=q.r; double h=FMA(Y*A0,r,Y*A1); h=FMA(h,r,Y*A2); h=FMA(h,r,Y*A3); h=FMA(h,r,Y*A4); return FMA(h,r,Y*q.y0); }
// mixed: Estrin-ish with 6 ops: r2; p=A2 r+A3; q=A4 r+y0; hi=A0 r + A1; t = hi*r2+p ; yy = t*r2+q  (mul,fma,fma,fma,fma,fma = 6) folded
static inline double log_estrin6_foldy(float x, double Y) { Red q = reduce(x); double r=q.r, r2=r*r; double hi=FMA(Y*A0,r,Y*A1), p=FMA(Y*A2,r,Y*A3), t=FMA(Y*A4,r,Y*q.y0); double u=FMA(hi,r2,p); return FMA(u,r2,t); }
// exp stage variants
static inline double kd_rr(double ylogx, uint64_t &ki) { const double SHIFT = 0x1.8p+52/32; double kd = ylogx + SHIFT; ki = pw_asuint64(kd); kd -= SHIFT; return ylogx - kd; }
static inline float exp_exact(double ylogx) { uint64_t ki; double rr=kd_rr(ylogx,ki); double s=pw_asdouble(T.exp2_tab[ki%32]+(ki<<47)); double zz=FMA(C0,rr,C1), rr2=rr*rr, e=FMA(C2,rr,1.0); e=FMA(zz,rr2,e); return (float)(e*s); }
static inline float exp_horner(double ylogx) { uint64_t ki; double rr=kd_rr(ylogx,ki); double s=pw_asdouble(T.exp2_tab[ki%32]+(ki<<47)); double h=FMA(C0,rr,C1); h=FMA(h,rr,C2); h=FMA(h,rr,1.0); return (float)(h*s); }
static inline float exp_premul(double ylogx) { uint64_t ki; double rr=kd_rr(ylogx,ki); double s=pw_asdouble(T.exp2_tab[ki%32]+(ki<<47)); double sr=s*rr; double h=FMA(C0,rr,C1); h=FMA(h,rr,C2); return (float)FMA(sr,h,s); }
typedef double (*LogF)(float,double); typedef float (*ExpF)(double);
int main() {
    const float m = 78.8438f, n = 0.1593f; volatile float one = 1.0f;
    struct Dom { const char *name; float y, lo, hi; } doms[] = {{"A val^(1/m) [2^-21,1]", one/m, 0x1p-21f, 1.0f}, {"C q^m [0.8359,1.0088]", m, 0.8359f, 1.0088f}, {"B t^(1/n) [3e-9,6.4]", one/n, 3e-9f, 6.4f}, {"E x^n [2^-64, 2^40] (encode first power, part)", n, 0x1p-64f, 0x1p40f}};
    struct V { const char *name; LogF l; ExpF e; int ops; } vs[] = {
        {"foldy + exact exp (16)", log_foldy, exp_exact,16}, {"horner + exact exp (15)", log_horner, exp_exact,15}, {"horner_foldy + exact exp (14)", log_horner_foldy, exp_exact,14},
        {"estrin6_foldy + exact exp (15)", log_estrin6_foldy, exp_exact,15},
        {"exact log + horner exp (16)", log_exact, exp_horner,16}, {"exact log + premul exp (16)", log_exact, exp_premul,16},
        {"horner_foldy + horner exp (13)", log_horner_foldy, exp_horner,13}, {"horner_foldy + premul exp (13)", log_horner_foldy, exp_premul,13},
        {"estrin6_foldy + horner exp (14)", log_estrin6_foldy, exp_horner,14}, {"estrin6_foldy + premul exp (14)", log_estrin6_foldy, exp_premul,14},
        {"foldy + horner exp (15)", log_foldy, exp_horner,15}, {"foldy + premul exp (15)", log_foldy, exp_premul,15}};
    const int NV = sizeof vs / sizeof vs[0];
    const unsigned nt = 8;
    for (auto &d : doms) {
        const uint32_t b0 = pw_asuint(d.lo), b1 = pw_asuint(d.hi); const double Y = (double)d.y;
        std::vector<std::atomic<uint64_t>> bad(NV); for (auto &b : bad) b = 0;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] { uint64_t c[32] = {0};
            for (uint64_t b = (uint64_t)b0 + t; b <= b1; b += nt) { float x = pw_asfloat((uint32_t)b); uint32_t ex = pw_asuint(exp_exact(log_exact(x, Y)));
                for (int v = 0; v < NV; v++) c[v] += pw_asuint(vs[v].e(vs[v].l(x, Y))) != ex; }
            for (int v = 0; v < NV; v++) bad[v] += c[v]; });
        for (auto &x : th) x.join();
        printf("%s: %.0f args\n", d.name, (double)b1 - b0 + 1);
        for (int v = 0; v < NV; v++) printf("   %-36s %llu\n", vs[v].name, (unsigned long long)bad[v].load());
    }
    {   // B*: the arguments PQdec's second power can actually be handed
        const float c1 = 0.8359f, c2 = 18.8516f, c3 = 18.6875f; const double Y = (double)(one / n);
        uint64_t cnt = 0, bad[NV]; for (int v = 0; v < NV; v++) bad[v] = 0;
        for (uint32_t b = pw_asuint(c1); b <= pw_asuint(1.0f); b++) { const float Vp = pw_asfloat(b), num = Vp - c1; if (!(num > 0.0f)) continue;
            volatile float prod = c3 * Vp; const float t = num / (c2 - prod); const double yl = log_exact(t, Y); if (fabs(yl) >= 126.0) continue;
            cnt++; const uint32_t ex = pw_asuint(exp_exact(yl));
            for (int v = 0; v < NV; v++) if (pw_asuint(vs[v].e(vs[v].l(t, Y))) != ex) { bad[v]++; if (v == 6) printf("   (differs at Vp=%a t=%a)\n", Vp, t); } }
        printf("B* t^(1/n) over t(Vp), Vp in (c1, 1]: %llu args\n", (unsigned long long)cnt);
        for (int v = 0; v < NV; v++) printf("   %-36s %llu\n", vs[v].name, (unsigned long long)bad[v]);
    }
}
