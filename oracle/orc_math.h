/* oracle/orc_math.h — TEST INFRASTRUCTURE ONLY (part of the CPU oracle; see bnmf_oracle.c).
 *
 * Numerical primitives of the "stream spec" (DESIGN.md §4): Philox4x32-10, the
 * u52 uniform, and log / exp / lgamma / digamma / qnorm built ONLY from IEEE-754
 * binary64 +,-,*,/ and sqrt, in a fixed operation order, so that a second
 * implementation of the same spec (the HIP engine) can agree bit for bit.
 * Compile with -ffp-contract=off.  Nothing here calls libm's log/exp.
 *
 * PARITY UNPINNED: the reference (pure R) ships no golden vectors and R is not
 * available in this image, and R's Mersenne-Twister/nmath stream cannot be
 * reproduced by a counter-based generator.  These primitives are pinned against
 * mpmath / scipy in tests/test_oracle_math.py instead.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>   /* sqrt, fabs only */

/* ---------------------------------------------------------------- Philox -- */
/* Philox4x32-R (Salmon et al., SC'11).  ctr = (c0,c1,c2,c3), key = (k0,k1).  Ten rounds for every stream but the
 * count-allocation words (V_Z), which take seven: the smallest round count the authors report as Crush-resistant. */
static inline void orc_philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                    uint32_t k0, uint32_t k1, uint32_t out[4], int rounds) {
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline void orc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
  orc_philox4x32_r(c0, c1, c2, c3, k0, k1, out, 10);
}

/* One stream per (variable, element, iteration): blocks of 4 words, block index
 * is counter word 0.  counter = (block, element, iteration, variable id). */
typedef struct { uint32_t k0, k1, elem, iter, var, blk; } orc_stream;

static inline orc_stream orc_stream_make(uint64_t seed, uint32_t chain, uint32_t var,
                                         uint32_t elem, uint32_t iter) {
  orc_stream s;
  s.k0 = (uint32_t)seed; s.k1 = (uint32_t)(seed >> 32) ^ chain;
  s.elem = elem; s.iter = iter; s.var = var; s.blk = 0;
  return s;
}
static inline void orc_stream_next(orc_stream* s, uint32_t w[4]) {
  orc_philox4x32_10(s->blk, s->elem, s->iter, s->var, s->k0, s->k1, w);
  s->blk++;
}
/* 52-bit uniform strictly inside (0,1): (26 high bits of a | 26 high bits of b) + 1/2, times 2^-52. */
static inline double orc_u52(uint32_t a, uint32_t b) {
  uint64_t x = ((uint64_t)(a >> 6) << 26) | (uint64_t)(b >> 6);
  return ((double)x + 0.5) * 2.220446049250313080847263336181640625e-16;
}

/* ----------------------------------------------------------- bit helpers -- */
static inline uint64_t orc_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double orc_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* ------------------------------------------------------------------- log -- */
/* Natural log, argument reduction x = 2^k * m, m in [sqrt(1/2), sqrt(2)), then
 * log(1+f) = 2s + s*R(s^2), s = f/(2+f) (the classical fdlibm-style scheme,
 * <1 ulp).  x<=0 -> -inf / NaN, NaN -> NaN, +inf -> +inf. */
static inline double orc_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  if (x != x) return x;
  if (x <= 0.0) return (x == 0.0) ? -INFINITY : NAN;
  if (x == INFINITY) return x;
  int k = 0;
  uint64_t u = orc_d2u(x);
  if ((u >> 52) == 0) { x = x * 18014398509481984.0; /* 2^54 */ u = orc_d2u(x); k = -54; }
  uint32_t hx = (uint32_t)(u >> 32);
  uint32_t lx = (uint32_t)u;
  k += (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  uint32_t i = (hx + 0x95f64u) & 0x100000u;
  u = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | lx;
  k += (int)(i >> 20);
  double m = orc_u2d(u);
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double dk = (double)k;
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* log(1 + f) for |f| <= 0.25: orc_log's kernel (its k = 0 branch) applied to f directly */
static inline double orc_log1p_small(double f) {
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return f - (hfsq - s * (hfsq + R));
}

/* ------------------------------------------------------------------- exp -- */
static inline double orc_exp(double x) {
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893383973096) return INFINITY;
  if (x < -745.13321910194110842) return 0.0;
  int k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
  double t = (double)k;
  double hi = x - t * ln2HI;
  double lo = t * ln2LO;
  double r = hi - lo;
  double tt = r * r;
  double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  /* scale by 2^k with exact power-of-two multiplications */
  if (k > 1023) { y = y * 8.98846567431157953865e307; k -= 1023; }        /* 2^1023 */
  if (k < -1022) { y = y * 2.22507385850720138309e-308; k += 1022;         /* 2^-1022 */
    if (k < -1022) { y = y * 2.22507385850720138309e-308; k += 1022; } }
  return y * orc_u2d((uint64_t)(1023 + k) << 52);
}

/* ------------------------------------------------- lgamma and digamma, x>0 -- */
/* Shift x up to xs >= 8 by the recurrences, then Stirling / asymptotic series.
 * lgamma(x) = (xs-.5)log xs - xs + .5log(2pi) + sum B2k/(2k(2k-1)) xs^(1-2k) - log(prod)
 * digamma(x) = log xs - 1/(2xs) - sum B2k/(2k) xs^(-2k) - sum 1/(x+i)          */
static inline void orc_lgamma_digamma(double x, double* lg, double* dg) {
  const double HALF_LOG_2PI = 0.91893853320467274178;
  if (!(x > 0.0)) { *lg = (x == 0.0) ? INFINITY : NAN; *dg = NAN; return; }
  double prod = 1.0, rs = 0.0, xs = x;
  while (xs < 8.0) { prod = prod * xs; rs = rs + 1.0 / xs; xs = xs + 1.0; }
  double lxs = orc_log(xs);
  double w = 1.0 / xs, w2 = w * w;
  double ser = w * (8.33333333333333333333e-02 + w2 * (-2.77777777777777777778e-03 +
               w2 * (7.93650793650793650794e-04 + w2 * (-5.95238095238095238095e-04 +
               w2 * (8.41750841750841750842e-04 + w2 * (-1.91752691752691752692e-03 +
               w2 * 6.41025641025641025641e-03))))));
  *lg = (((xs - 0.5) * lxs - xs) + HALF_LOG_2PI) + ser - orc_log(prod);
  double ds = w2 * (8.33333333333333333333e-02 - w2 * (8.33333333333333333333e-03 -
              w2 * (3.96825396825396825397e-03 - w2 * (4.16666666666666666667e-03 -
              w2 * (7.57575757575757575758e-03 - w2 * (2.10927960927960927961e-02 -
              w2 * 8.33333333333333333333e-02))))));
  *dg = ((lxs - 0.5 * w) - ds) - rs;
}
static inline double orc_lgamma(double x) { double a, b; orc_lgamma_digamma(x, &a, &b); return a; }
static inline double orc_digamma(double x) { double a, b; orc_lgamma_digamma(x, &a, &b); return b; }

/* ----------------------------------------------------------------- qnorm -- */
/* Standard normal quantile, Wichura's AS241 (PPND16), lower tail, p in (0,1). */
static inline double orc_qnorm(double p) {
  /* central part (w = -log(4 p (1 - p)) < 6.25): y F(w), y = 2 p - 1, F a degree-24 polynomial in w - 3.125 (Chebyshev
   * interpolation of sqrt(2) erfinv(y) / y, tools/fit_qnorm.py); tails: AS241's outer branches */
  double q = p - 0.5, r, val;
  {
    const double y = q + q;
    const double w = -orc_log((1.0 - y) * (1.0 + y));
    if (w < 6.25) {
      const double s = w - 3.125;
      double a = -5.081556263217504e-22;
      a = a * s + 4.51702010660485e-21;
      a = a * s + 2.8273805909578654e-20;
      a = a * s + -4.912241660340459e-19;
      a = a * s + 8.591419094975e-19;
      a = a * s + 2.1935565352484576e-17;
      a = a * s + -1.7275519790527177e-16;
      a = a * s + -5.57226748086415e-17;
      a = a * s + 9.222558296724625e-15;
      a = a * s + -5.659687073061811e-14;
      a = a * s + -1.1415427137390895e-13;
      a = a * s + 3.7201267209535326e-12;
      a = a * s + -1.8354802577559388e-11;
      a = a * s + -7.65698113492015e-11;
      a = a * s + 1.4866543569083624e-09;
      a = a * s + -5.8161801676714855e-09;
      a = a * s + -4.111174160550445e-08;
      a = a * s + 5.988894861277318e-07;
      a = a * s + -1.931065027529881e-06;
      a = a * s + -1.9632852863696442e-05;
      a = a * s + 0.00026408204954061674;
      a = a * s + -0.001047511569485617;
      a = a * s + -0.008532899177267343;
      a = a * s + 0.3396349587011389;
      a = a * s + 2.338620710026593;
      return y * a;
    }
  }
  r = (q < 0.0) ? p : (1.0 - p);
  r = sqrt(-orc_log(r));
  if (r <= 5.0) {
    r = r - 1.6;
    val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r + 0.24178072517745061177) * r
               + 1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r
            + 4.6303378461565452959) * r + 1.42343711074968357734)
        / (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r
               + 0.14810397642748007459) * r + 0.68976733498510000455) * r + 1.6763848301838038494) * r
            + 2.05319162663775882187) * r + 1.0);
  } else {
    r = r - 5.0;
    val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r
               + 0.026532189526576123093) * r + 0.29656057182850489123) * r + 1.7848265399172913358) * r
            + 5.4637849111641143699) * r + 6.6579046435011037772)
        / (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r
               + 7.868691311456132591e-4) * r + 0.0148753612908506148525) * r + 0.13692988092273580531) * r
            + 0.59983220655588793769) * r + 1.0);
  }
  return (q < 0.0) ? -val : val;
}

/* log of the standard normal CDF, log Phi(z), via erfc-type continued range
 * split: used only for the truncated-normal log-density normaliser
 * (R/utils.R:134-145 dtruncnorm).  |rel err| ~ 1e-15 for z > -37.            */
/* exp(x^2)*erfc(x) for x >= 0: W. J. Cody (1969) rational Chebyshev approximations */
static inline double orc_erfcx_cody(double y) {
  /* y >= 0 */
  if (y <= 0.46875) {
    /* erf small: erfc = 1 - erf */
    const double a[5] = {3.16112374387056560e00, 1.13864154151050156e02, 3.77485237685302021e02,
                         3.20937758913846947e03, 1.85777706184603153e-1};
    const double b[4] = {2.36012909523441209e01, 2.44024637934444173e02, 1.28261652607737228e03,
                         2.84423683343917062e03};
    double ysq = y * y;
    double xnum = a[4] * ysq, xden = ysq;
    for (int i = 0; i < 3; ++i) { xnum = (xnum + a[i]) * ysq; xden = (xden + b[i]) * ysq; }
    double erf = y * (xnum + a[3]) / (xden + b[3]);
    return (1.0 - erf) * orc_exp(ysq);
  } else if (y <= 4.0) {
    const double c[9] = {5.64188496988670089e-1, 8.88314979438837594e00, 6.61191906371416295e01,
                         2.98635138197400131e02, 8.81952221241769090e02, 1.71204761263407058e03,
                         2.05107837782607147e03, 1.23033935479799725e03, 2.15311535474403846e-8};
    const double d[8] = {1.57449261107098347e01, 1.17693950891312499e02, 5.37181101862009858e02,
                         1.62138957456669019e03, 3.29079923573345963e03, 4.36261909014324716e03,
                         3.43936767414372164e03, 1.23033935480374942e03};
    double xnum = c[8] * y, xden = y;
    for (int i = 0; i < 7; ++i) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
    return (xnum + c[7]) / (xden + d[7]);
  } else {
    const double p[6] = {3.05326634961232344e-1, 3.60344899949804439e-1, 1.25781726111229246e-1,
                         1.60837851487422766e-2, 6.58749161529837803e-4, 1.63153871373020978e-2};
    const double q[5] = {2.56852019228982242e00, 1.87295284992346725e00, 5.27905102951428412e-1,
                         6.05183413124413191e-2, 2.33520497626869185e-3};
    const double sqrpi = 5.6418958354775628695e-1;
    double ysq = 1.0 / (y * y);
    double xnum = p[5] * ysq, xden = ysq;
    for (int i = 0; i < 4; ++i) { xnum = (xnum + p[i]) * ysq; xden = (xden + q[i]) * ysq; }
    double r = ysq * (xnum + p[4]) / (xden + q[4]);
    return (sqrpi - r) / y;
  }
}
/* log Phi(z) */
static inline double orc_log_pnorm(double z) {
  const double SQRT1_2 = 0.70710678118654752440;
  if (z != z) return z;
  if (z >= 0.0) {
    /* Phi = 1 - 0.5 erfc(z/sqrt2) */
    double y = z * SQRT1_2;
    double e = 0.5 * orc_erfcx_cody(y) * orc_exp(-(y * y));
    /* log1p(-e), e in (0, .5], by the u = 1-e correction trick (exact ops only) */
    double u = 1.0 - e;
    if (u == 1.0) return -e;
    return orc_log(u) * (-e) / (u - 1.0);
  } else {
    double y = -z * SQRT1_2;
    /* Phi = 0.5 erfc(y) = 0.5 erfcx(y) exp(-y^2) */
    return orc_log(0.5 * orc_erfcx_cody(y)) - y * y;
  }
}

/* --------------------------------------------------- canonical summation -- */
/* canon_sum(x, L, stride, W): W partial accumulators; accumulator i adds
 * x[i], x[i+W], x[i+2W], ... in that order starting from +0.0; then a halving
 * tree acc[i] += acc[i+h], h = W/2 ... 1.  This is exactly what a W-lane
 * wavefront / workgroup reduction on the GPU does, so fp sums agree bitwise.  */
static inline double orc_canon_sum(const double* x, long L, long stride, int W) {
  double acc[1024];
  for (int i = 0; i < W; ++i) {
    double a = 0.0;
    for (long j = i; j < L; j += W) a = a + x[j * stride];
    acc[i] = a;
  }
  for (int h = W / 2; h >= 1; h >>= 1)
    for (int i = 0; i < h; ++i) acc[i] = acc[i] + acc[i + h];
  return acc[0];
}
#endif
