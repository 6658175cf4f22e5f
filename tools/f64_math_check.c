// Host emulation of the f64 fast pair force pieces; checks ulp error vs glibc / long double.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static double seed_rsq(double x){ return (double)(1.0f/sqrtf((float)x)) * (1.0 + 3e-8); }
static double seed_rcp(double x){ return (double)(1.0f/(float)x) * (1.0 - 3e-8); }
static double exp_neg(double a){ // exp(-a), a>=0
  double y = fmax(-a, -800.0);
  double n = rint(y * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, y);
  r = fma(n, -1.90821492927058770002e-10, r);
  static const double c[14] = {1.0,1.0,0.5,1.0/6,1.0/24,1.0/120,1.0/720,1.0/5040,1.0/40320,1.0/362880,1.0/3628800,1.0/39916800,1.0/479001600,1.0/6227020800.0};
  double p = c[13];
  for (int k = 12; k >= 0; --k) p = fma(p, r, c[k]);
  return ldexp(p, (int)n);
}
static double log1p_01(double t){ // t in [0,1]
  const double ln2_hi=6.93147180369123816490e-01, ln2_lo=1.90821492927058770002e-10,
  Lp1=6.666666666666735130e-01,Lp2=3.999999999940941908e-01,Lp3=2.857142874366239149e-01,Lp4=2.222219843214978396e-01,
  Lp5=1.818357216161805012e-01,Lp6=1.531383769920937332e-01,Lp7=1.479819860511658591e-01;
  double u = 1.0 + t;
  double cn = t - (u - 1.0);
  int k = u >= 1.4142135623730951 ? 1 : 0;
  double m = k ? u*0.5 : u;
  double yu = seed_rcp(u); double eu = fma(-u, yu, 1.0); yu = fma(yu, eu, yu); eu = fma(-u, yu, 1.0); yu = fma(yu, eu, yu); double c = cn*yu;
  double f = m - 1.0;
  double hfsq = 0.5*f*f;
  double den = 2.0 + f;
  double y = seed_rcp(den);
  double e = fma(-den, y, 1.0); y = fma(y, e, y);
  e = fma(-den, y, 1.0); y = fma(y, e, y);
  double s = f*y, z = s*s;
  double R = z*fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, Lp7, Lp6), Lp5), Lp4), Lp3), Lp2), Lp1);
  double kd = k;
  double lo = fma(kd, ln2_lo, c);
  double res = kd*ln2_hi - ((hfsq - fma(s, hfsq + R, lo)) - f);
  return res;
}
static void sqrt_rcp(double x, double *sq, double *rc){
  double y0 = seed_rsq(x);
  double g = x*y0, h = 0.5*y0;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  double d = fma(-g, g, x); g = fma(d, h, g);
  d = fma(-g, g, x); g = fma(d, h, g);
  *sq = g; *rc = h + h;
}
static double ulp_err(double got, long double want){
  if (want == 0) return got == 0 ? 0 : 1e9;
  int e; frexpl(want, &e); if (e < -1021) e = -1021; long double u = ldexpl(1.0L, e - 53);
  return (double)fabsl(((long double)got - want)/u);
}
int main(){
  srand48(1);
  double mx_e=0, mx_l=0, mx_s=0, mx_dk=0, mx_q=0, mx_sp=0; long bad_dk=0, bad_s=0, bad_q=0;
  const double k=1e-3, rk = 1.0/k;
  for (long i=0;i<20000000;i++){
    double a = drand48()*(i%4==0?800:(i%4==1?40:(i%4==2?2:0.01)));
    double e = exp_neg(a); double er = ulp_err(e, expl(-(long double)a)); if (er>mx_e) mx_e=er;
    double t = (i%2)? drand48() : exp(-a);
    double l = log1p_01(t); double lr = ulp_err(l, log1pl((long double)t)); if (lr>mx_l) mx_l=lr;
    // softplus tail
    double sp = log1p_01(exp_neg(a)); double spr = ulp_err(sp, log1pl(expl(-(long double)a))); if (spr>mx_sp) mx_sp=spr;
    double x = drand48()*8.0 + 1e-9; double sq, rc; sqrt_rcp(x,&sq,&rc);
    if (sq != sqrt(x)) { bad_s++; double sr=ulp_err(sq, sqrtl((long double)x)); if (sr>mx_s) mx_s=sr; }
    // markstein /k
    double nn = (drand48()-0.5)*6.0; double q = nn*rk; double r = fma(-q,k,nn); double q2 = fma(r, rk, q);
    if (q2 != nn/k) { bad_dk++; }
    // n / dist via rc
    double dist = sq; double q0 = nn*rc; double rr = fma(-q0, dist, nn); double q1 = fma(rr, rc, q0);
    if (q1 != nn/dist) { bad_q++; double qr = ulp_err(q1, (long double)nn/(long double)dist); if (qr>mx_q) mx_q=qr; }
  }
  printf("exp max ulp %.3f  log1p max ulp %.3f softplus-tail %.3f sqrt mismatches %ld (max ulp %.3f) div-k mismatches %ld  n/dist mismatches %ld (max %.3f)\n", mx_e, mx_l, mx_sp, bad_s, mx_s, bad_dk, bad_q, mx_q);
  return 0;
}
