#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static const double C = 0.49999999999999994; /* nextbefore(0.5) */
static inline long long ref(double v){ double r = round(v); if (r!=r) return 0; if (r < -2147483648.0) return INT32_MIN; if (r > 2147483647.0) return INT32_MAX; return (long long)r; }
/* v_cvt_i32_f64: truncate toward zero, NaN -> 0, saturate */
static inline long long cvt_i32(double x){ if (x!=x) return 0; if (x <= -2147483648.0) return INT32_MIN; if (x >= 2147483647.0) return INT32_MAX; return (long long)x; }
static inline long long fast(double v){ return cvt_i32(v + copysign(C, v)); }
static unsigned long long bad=0, n=0;
static void chk(double v){ n++; if (ref(v)!=fast(v)) { if (bad<20) printf("MISMATCH v=%.17g (%a): ref %lld fast %lld\n", v, v, ref(v), fast(v)); bad++; } }
int main(){
  /* around every half-integer and integer up to 2^22, +-4 ulps, both signs */
  for (long long k=0;k<(1<<23);k++){ double h=k*0.5; double lo=h, hi=h; chk(h); chk(-h);
    for(int u=0;u<4;u++){ lo=nextafter(lo,-INFINITY); hi=nextafter(hi,INFINITY); chk(lo); chk(hi); chk(-lo); chk(-hi);} }
  /* powers of two neighbourhood up to 2^62 */
  for (int e=-1074;e<64;e++){ double p=ldexp(1.0,e); double lo=p,hi=p; chk(p); chk(-p); chk(p+0.5); chk(p-0.5); chk(-(p+0.5));
    for(int u=0;u<4;u++){ lo=nextafter(lo,0); hi=nextafter(hi,INFINITY); chk(lo); chk(hi); chk(-lo); chk(-hi); chk(lo+0.5); chk(hi+0.5);} }
  /* random */
  srand48(1);
  for (long long i=0;i<400000000LL;i++){ double s=(drand48()-0.5); int e=(int)(lrand48()%40)-4; chk(ldexp(s,e)); }
  for (long long i=0;i<50000000LL;i++){ uint64_t b=((uint64_t)lrand48()<<33)^((uint64_t)lrand48()<<11)^(uint64_t)lrand48(); double v; memcpy(&v,&b,8); chk(v); }
  chk(NAN); chk(INFINITY); chk(-INFINITY); chk(0.0); chk(-0.0);
  printf("checked %llu values, %llu mismatches\n", n, bad); return bad!=0; }
