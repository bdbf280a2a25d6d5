/* x / 255.0 as multiply + two fmas (tr_device.h div255) against the division, on the host. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline double fast(double x){ double ax=fabs(x); if(!(ax<1e150)||ax<1e-150) return x/255.0; const double r=1.0/255.0; double q=x*r; double e=fma(-255.0,q,x); return fma(e,r,q); }
static unsigned long long bad=0,n=0;
static void chk(double x){ n++; double a=x/255.0,b=fast(x); if(memcmp(&a,&b,8)!=0 && !(a!=a&&b!=b)){ if(bad<10) printf("MISMATCH %a: %a vs %a\n",x,a,b); bad++; } }
int main(){
  srand48(7);
  /* what the shader feeds it: bilinear mixes of bytes with weights in [0,1] */
  for(long long i=0;i<200000000LL;i++){ double fx=drand48(),fy=drand48(); int a=lrand48()&255,b=lrand48()&255,c=lrand48()&255,d=lrand48()&255;
    double w00=(1-fx)*(1-fy),w10=fx*(1-fy),w01=(1-fx)*fy,w11=fx*fy; chk(a*w00+b*w10+c*w01+d*w11); }
  for(int k=0;k<=255*256;k++) chk(k/256.0);
  for(long long i=0;i<100000000LL;i++){ uint64_t b=((uint64_t)lrand48()<<33)^((uint64_t)lrand48()<<11)^(uint64_t)lrand48(); double v; memcpy(&v,&b,8); chk(v); }
  for(int e=-1074;e<1024;e++){ double p=ldexp(1.0,e); chk(p); chk(-p); chk(p*255.0); chk(nextafter(p*255.0,0)); chk(nextafter(p*255.0,INFINITY)); }
  chk(0.0); chk(-0.0); chk(INFINITY); chk(-INFINITY); chk(NAN);
  printf("checked %llu values, %llu mismatches\n",n,bad); return bad!=0; }
