// CPU check of sincos_cr_core (bgk_kernels.h): every fp32 t in [0, 2 pi] against (float)sin((double)t), (float)cos((double)t);
// the same IEEE operations as the device (fmaf / fma).  build: gcc -O2 -mfma -fopenmp -ffp-contract=off tools/check/sincos_sweep.c -o /tmp/sincos_sweep -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <omp.h>
static const unsigned long long kSinCosTab[128][2] = {
#include "../../la3dm_amd/csrc/sincos_table.inc"
};
static inline double D(unsigned long long b){ double d; memcpy(&d,&b,8); return d; }
static inline float F(uint32_t b){ float f; memcpy(&f,&b,4); return f; }
static inline uint32_t U(float f){ uint32_t b; memcpy(&b,&f,4); return b; }
static inline void sc(float t, float *s, float *c, int *bad){
  const float magic = 12582912.0f, h = 0x1.921fb6p-4f, ih = 10.1859164f; /* 32/pi */
  float u = fmaf(t, ih, magic);
  float kf = u - magic;
  float y32 = fmaf(-kf, h, t);
  if ((double)y32 != (double)t - (double)kf*(double)h) *bad = 1;
  uint32_t q = U(u) & 127u;
  if ((float)q != kf) *bad = 2;
  double sa = D(kSinCosTab[q][0]), ca = D(kSinCosTab[q][1]);
  double y = y32, z = y*y, zy = z*y;
  double ps = fma(z, D(0xbf2a014a5f1de813ull), D(0x3f81111110c194d4ull));
  ps = fma(z, ps, D(0xbfc555555555552aull));
  double sy = fma(zy, ps, y);
  double pc = fma(z, D(0x3efa015b846c26deull), D(0xbf56c16c1681d56aull));
  pc = fma(z, pc, D(0x3fa5555555555544ull));
  pc = fma(z, pc, -0.5);
  double cm1 = z*pc;
  double sd = fma(ca, sy, fma(sa, cm1, sa));
  double cd = fma(-sa, sy, fma(ca, cm1, ca));
  *s = (float)sd; *c = (float)cd;
}
int main(){
  uint32_t hi = U(6.2831855f);
  unsigned long long ms=0, mc=0, nb=0;
  #pragma omp parallel for reduction(+:ms,mc,nb) schedule(dynamic, 1<<20)
  for (uint32_t b = 0; b <= hi; ++b){
    float t = F(b), s, c; int bad = 0;
    sc(t,&s,&c,&bad);
    float rs = (float)sin((double)t), rc = (float)cos((double)t);
    if (bad) nb++;
    if (U(s)!=U(rs)) { ms++; if (ms < 5) printf("sin t=%a got %a want %a\n", t, s, rs);}
    if (U(c)!=U(rc)) { mc++; if (mc < 5) printf("cos t=%a got %a want %a\n", t, c, rc);}
  }
  printf("swept %u values: sin mismatches %llu, cos mismatches %llu, inexact reductions %llu\n", hi+1, ms, mc, nb);
  return 0;
}
