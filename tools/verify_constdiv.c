// tools/verify_constdiv.c -- exhaustive check of the 3-op constant division used by luma_device.hpp:div_255_pos.
//   gcc -O2 -mfma -ffp-contract=off -o /tmp/verify_constdiv tools/verify_constdiv.c -lm && /tmp/verify_constdiv   (~6 min, 1 thread)
// Result on this image: C=255 and C=219 mismatch only for a in {-0, +inf, -inf}; 410, 224, 1.8814, 1.4746, 0.678 are NOT exact.
// exhaustive check: q = a*RC; r = fma(-C,q,a); q' = fma(r,RC,q)  ==  a / C   for all float a
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void){
  const float Cs[] = {255.0f, 410.0f, 1.8814f, 1.4746f, 224.0f, 219.0f, 0.6780f};
  for (unsigned ci=0; ci<sizeof Cs/sizeof Cs[0]; ci++){
    const float C=Cs[ci]; volatile float one=1.0f; const float RC=one/C;
    uint64_t bad=0; uint32_t lo_bad=0xffffffff, hi_bad=0; 
    for (uint64_t u=0; u<=0xffffffffull; u++){
      uint32_t b=(uint32_t)u; float a; memcpy(&a,&b,4);
      float q=a*RC; float r=fmaf(-C,q,a); float q2=fmaf(r,RC,q);
      float ref=a/C;
      uint32_t x,y; memcpy(&x,&q2,4); memcpy(&y,&ref,4);
      if (x!=y && !(q2!=q2 && ref!=ref)){ bad++; uint32_t m=b&0x7fffffff; if(m<lo_bad)lo_bad=m; if(m>hi_bad)hi_bad=m; }
    }
    float lo,hi; memcpy(&lo,&lo_bad,4); memcpy(&hi,&hi_bad,4);
    printf("C=%g RC=%a: %llu mismatches; |a| range of mismatches [%g, %g]\n", C, RC, (unsigned long long)bad, bad?lo:0.0f, bad?hi:0.0f);
  }
  return 0;
}
