/* oracle/orc_samplers.h — TEST INFRASTRUCTURE ONLY (part of the CPU oracle).
 *
 * Scalar samplers of the stream spec.  Each consumes whole Philox
 * blocks from a per-(variable, element, iteration) stream in a fixed pattern, so
 * the HIP engine draws identical values independent of its launch geometry.
 * The *distributions* are the ones the reference's R code asks base-R / truncnorm /
 * invgamma / armspp for (call sites cited per function); the algorithms are this
 * repo's own because R's Mersenne-Twister stream cannot be reproduced.
 */
#ifndef ORC_SAMPLERS_H
#define ORC_SAMPLERS_H
#include "orc_math.h"

#define ORC_MAX_ATTEMPTS 2000

/* N(0,1): one block, words 0,1 -> u -> qnorm(u).  (stats::rnorm, R/sample_priors.R:34,48,219,235) */
static inline double orc_rnorm_std(orc_stream* s) {
  uint32_t w[4]; orc_stream_next(s, w);
  return orc_qnorm(orc_u52(w[0], w[1]));
}
/* U(0,1): one block, words 0,1.  (stats::runif, R/sample_Pn.R:243, R/sample_En.R:236) */
static inline double orc_runif(orc_stream* s) {
  uint32_t w[4]; orc_stream_next(s, w);
  return orc_u52(w[0], w[1]);
}
/* Exp(rate): one block.  (stats::rexp, R/sample_Pn.R:21,66) */
static inline double orc_rexp(orc_stream* s, double rate) {
  uint32_t w[4]; orc_stream_next(s, w);
  return -orc_log(orc_u52(w[0], w[1])) / rate;
}

/* Gamma(shape a, rate): Marsaglia & Tsang (2000).  One block per attempt
 * (words 0,1 -> normal by inversion; words 2,3 -> acceptance uniform); for a<1
 * one further block for the U^(1/a) boost.
 * (stats::rgamma, R/sample_Pn.R:117, R/sample_En.R:116, R/sample_priors.R:72-127,285-344) */
static inline double orc_rgamma(orc_stream* s, double a, double rate) {
  if (!(a > 0.0)) return (a == 0.0) ? 0.0 : NAN;
  int boost = a < 1.0;
  double a1 = boost ? a + 1.0 : a;
  double d = a1 - 0.333333333333333333333;
  double c = 1.0 / sqrt(9.0 * d);
  double v = 1.0;
  for (int it = 0; it < ORC_MAX_ATTEMPTS; ++it) {
    uint32_t w[4]; orc_stream_next(s, w);
    double z = orc_qnorm(orc_u52(w[0], w[1]));
    double ua = orc_u52(w[2], w[3]);
    double cz = c * z;
    v = 1.0 + cz;
    if (v <= 0.0) continue;
    v = v * v * v;
    double z2 = z * z;
    if (ua < 1.0 - 0.0331 * (z2 * z2)) break;
    /* accept iff log(ua) < R; (ua - 1) / ua <= log(ua) <= ua - 1 decides all but a sliver without the logarithm */
    double lv = fabs(cz) <= 0.25 ? 3.0 * orc_log1p_small(cz) : orc_log(v);   /* log v = 3 log(1 + c z) */
    double R = 0.5 * z2 + d * ((1.0 - v) + lv);
    double um1 = ua - 1.0;
    if (um1 < R) break;
    if (um1 / ua >= R) continue;
    if (orc_log(ua) < R) break;
  }
  double g = d * v;
  if (boost) {
    uint32_t w[4]; orc_stream_next(s, w);
    g = g * orc_exp(orc_log(orc_u52(w[0], w[1])) / a);
  }
  return g / rate;
}
/* InvGamma(shape, rate) = 1 / Gamma(shape, rate)  (invgamma::rinvgamma, R/sample_priors.R:41,55,247,264; R/sample_params.R:279) */
static inline double orc_rinvgamma(orc_stream* s, double shape, double rate) {
  return 1.0 / orc_rgamma(s, shape, rate);
}

/* Normal truncated to [0, inf): TN(mu, sd; 0, inf).
 * alpha = -mu/sd is the standardised lower bound.  alpha < 0.45: draw normals
 * until z >= alpha (one block per attempt, words 0,1).  Otherwise Robert's (1995)
 * translated-exponential rejection (one block per attempt: words 0,1 -> e, 2,3 -> u).
 * (truncnorm::rtruncnorm, R/sample_Pn.R:14,59,79, R/sample_En.R:14,59,78) */
static inline double orc_rtnorm0(orc_stream* s, double mu, double sd) {
  double alpha = -mu / sd;
  double z = alpha;
  if (alpha < 0.45) {
    for (int it = 0; it < ORC_MAX_ATTEMPTS; ++it) {
      uint32_t w[4]; orc_stream_next(s, w);
      z = orc_qnorm(orc_u52(w[0], w[1]));
      if (z >= alpha) break;
    }
  } else {
    double lam = 0.5 * (alpha + sqrt(alpha * alpha + 4.0));
    for (int it = 0; it < ORC_MAX_ATTEMPTS; ++it) {
      uint32_t w[4]; orc_stream_next(s, w);
      double e = -orc_log(orc_u52(w[0], w[1])) / lam;
      z = alpha + e;
      double t = z - lam;
      double rho = orc_exp(-0.5 * (t * t));
      if (orc_u52(w[2], w[3]) <= rho) break;
    }
  }
  double x = mu + sd * z;
  return x < 0.0 ? 0.0 : x;
}

/* Gamma-shape hyper-parameter draw (armspp::arms target, R/sample_priors.R:356-397):
 *   log f(x) = (c-1) log x - tau x - lgamma(x) on [1e-3, 1e4],
 *   tau = d - log(beta) - log(value).
 * f is log-concave for every c>0 (trigamma(x) >= 1/x^2), so ARMS returns an exact
 * draw; here: exact rejection from a 3-tangent piecewise-exponential hull placed
 * around a cheap mode estimate started from the previous Alpha (placement only
 * affects efficiency, never the distribution).  One block per attempt. */
static inline void orc_alpha_h(double x, double c, double tau, double* h, double* hp) {
  double lg, dg; orc_lgamma_digamma(x, &lg, &dg);
  *h = ((c - 1.0) * orc_log(x) - tau * x) - lg;
  *hp = ((c - 1.0) / x - tau) - dg;
}
static inline double orc_ralpha(orc_stream* s, double c, double tau, double xprev, int* n_attempts) {
  const double L = 1e-3, U = 1e4, DELTA = 1.41421356237309504880;
  double x = xprev;
  if (!(x >= L)) x = L;
  if (x > U) x = U;
  /* (1) cheap mode estimate: safeguarded Newton on H(x) = c/x - tau - log(x+1/2), the
   * derivative of log f with digamma(x) ~ log(x+1/2) - 1/x (right at both ends) */
  for (int it = 0; it < 32; ++it) {
    double Hx = (c / x - tau) - orc_log(x + 0.5);
    double dH = -c / (x * x) - 1.0 / (x + 0.5);
    double xn = x - Hx / dH;
    if (!(xn > 0.1 * x)) xn = 0.1 * x;
    if (xn > 10.0 * x) xn = 10.0 * x;
    if (xn < L) xn = L;
    if (xn > U) xn = U;
    double dx = fabs(xn - x);
    x = xn;
    if (dx <= 1e-3 * x) break;
  }
  double m = x;
  double g = (c / m - tau) - orc_log(m + 0.5);
  double kap = c / (m * m) + 1.0 / (m + 0.5);
  double sc = 1.0 / sqrt(kap);
  if (fabs(g) * sc > 1.0) sc = 1.0 / fabs(g);
  double x1, x2, x3, h1, s1, h2, s2, h3, s3;
  int walked = 0;
  for (int round = 0; round < 4; ++round) {
    /* (2) three design points around m at +-sqrt(2) scale units, inside [L,U] */
    x1 = m - DELTA * sc; x2 = m; x3 = m + DELTA * sc;
    if (x1 < L) x1 = L;
    if (x3 > U) x3 = U;
    if (x2 - x1 < 0.25 * sc) { x1 = L; x2 = L + 0.75 * sc; x3 = L + 2.5 * sc; }        /* mode at/near L */
    else if (x3 - x2 < 0.25 * sc) { x3 = U; x2 = U - 0.75 * sc; x1 = U - 2.5 * sc; }   /* mode at/near U */
    if (x1 < L) x1 = L;
    if (x3 > U) x3 = U;
    if (!(x1 < x2 && x2 < x3)) { x1 = L; x2 = 0.5 * (L + U); x3 = U; }
    orc_alpha_h(x1, c, tau, &h1, &s1);
    orc_alpha_h(x2, c, tau, &h2, &s2);
    orc_alpha_h(x3, c, tau, &h3, &s3);
    /* (3) bracket the mode with exact slopes: walk the window, doubling its step */
    walked = 0;
    for (int it = 0; it < 64 && s3 > 0.0 && x3 < U; ++it) {
      double step = 2.0 * (x3 - x2);
      x1 = x2; h1 = h2; s1 = s2; x2 = x3; h2 = h3; s2 = s3;
      x3 = x3 + step; if (x3 > U) x3 = U;
      orc_alpha_h(x3, c, tau, &h3, &s3);
      walked = 1;
    }
    for (int it = 0; it < 64 && s1 < 0.0 && x1 > L; ++it) {
      double step = 2.0 * (x2 - x1);
      x3 = x2; h3 = h2; s3 = s2; x2 = x1; h2 = h1; s2 = s1;
      x1 = x1 - step; if (x1 < L) x1 = L;
      orc_alpha_h(x1, c, tau, &h1, &s1);
      walked = 1;
    }
    if (!walked) break;
    /* (4) the window moved: re-centre on the secant root of the exact slope between the
     * pair of points that brackets it, with the scale the exact slopes imply */
    double xa, xb, sa, sb;
    if (s2 > 0.0) { xa = x2; sa = s2; xb = x3; sb = s3; } else { xa = x1; sa = s1; xb = x2; sb = s2; }
    if (!(sa > 0.0 && sb < 0.0)) break;            /* mode on the boundary: keep this window */
    m = xa + sa * (xb - xa) / (sa - sb);
    sc = sqrt((xb - xa) / (sa - sb));
    if (!(sc > 0.0)) break;
  }
  /* breakpoints: tangent intersections, clamped between their design points */
  double z1 = 0.5 * (x1 + x2), z2 = 0.5 * (x2 + x3);
  if (s1 - s2 > 1e-14 * (fabs(s1) + fabs(s2))) {
    z1 = (((h2 - h1) - s2 * x2) + s1 * x1) / (s1 - s2);
    if (!(z1 >= x1)) z1 = x1;
    if (z1 > x2) z1 = x2;
  }
  if (s2 - s3 > 1e-14 * (fabs(s2) + fabs(s3))) {
    z2 = (((h3 - h2) - s3 * x3) + s2 * x2) / (s2 - s3);
    if (!(z2 >= x2)) z2 = x2;
    if (z2 > x3) z2 = x3;
  }
  /* segment j: [lo_j, hi_j], tangent T_j(x) = h_j + s_j (x - x_j) */
  double lo[3] = {L, z1, z2}, hi[3] = {z1, z2, U};
  double hh[3] = {h1, h2, h3}, ss[3] = {s1, s2, s3}, xx[3] = {x1, x2, x3};
  double Tlo[3], Thi[3], ref = -INFINITY;
  for (int j = 0; j < 3; ++j) {
    Tlo[j] = hh[j] + ss[j] * (lo[j] - xx[j]);
    Thi[j] = hh[j] + ss[j] * (hi[j] - xx[j]);
    if (Tlo[j] > ref) ref = Tlo[j];
    if (Thi[j] > ref) ref = Thi[j];
  }
  double A[3];
  for (int j = 0; j < 3; ++j) {
    double wj = hi[j] - lo[j];
    double sw = ss[j] * wj;
    if (fabs(sw) < 1e-6) A[j] = orc_exp(Tlo[j] - ref) * wj * (1.0 + 0.5 * sw);
    else A[j] = (orc_exp(Thi[j] - ref) - orc_exp(Tlo[j] - ref)) / ss[j];
    if (!(A[j] > 0.0)) A[j] = 0.0;
  }
  double Atot = (A[0] + A[1]) + A[2];
  double xs = m;
  int it = 0;
  for (; it < ORC_MAX_ATTEMPTS; ++it) {
    uint32_t w[4]; orc_stream_next(s, w);
    double ua = orc_u52(w[0], w[1]) * Atot;
    double u2 = orc_u52(w[2], w[3]);
    int j; double r;
    if (ua < A[0]) { j = 0; r = ua / A[0]; }
    else if (ua < A[0] + A[1]) { j = 1; r = (ua - A[0]) / A[1]; }
    else { j = 2; r = ((ua - A[0]) - A[1]) / A[2]; }
    if (!(r <= 1.0)) r = 1.0;
    double wj = hi[j] - lo[j], sj = ss[j], sw = sj * wj;
    if (fabs(sw) < 1e-6) xs = lo[j] + r * wj;
    else if (sj > 0.0) xs = hi[j] + orc_log(r + (1.0 - r) * orc_exp(-sw)) / sj;
    else xs = lo[j] + orc_log((1.0 - r) + r * orc_exp(sw)) / sj;
    if (xs < lo[j]) xs = lo[j];
    if (xs > hi[j]) xs = hi[j];
    double hx, hpx; orc_alpha_h(xs, c, tau, &hx, &hpx);
    double Tx = hh[j] + sj * (xs - xx[j]);
    if (orc_log(u2) <= hx - Tx) break;
  }
  if (n_attempts) *n_attempts = it + 1;
  return xs;
}

/* ---- Gamma-shape hyper-parameter, fast path (the sampler the sweep uses; same target as orc_ralpha) ----
 * f(x) ~ x^(c-1) e^(-tau x) / Gamma(x) on [1e-3, 1e4]  (armspp::arms call sites R/sample_priors.R:365,391).
 * -lgamma is concave, so its tangent at x0 bounds it from above: f(x) <= const * x^(c-1) e^(-(tau + psi(x0)) x), a
 * Gamma(c, r = tau + psi(x0)) envelope.  x ~ Gamma(c, r) by Marsaglia-Tsang, kept with probability
 * exp(lgamma(x0) + psi(x0)(x - x0) - lgamma(x)) <= 1; both acceptance tests share ONE uniform (accept iff u < p_MT * p_tilt),
 * so an attempt is one Philox block (words 0,1 -> normal, words 2,3 -> uniform).  The tangent point is the grid point (doubles
 * with 6 mantissa bits: lgamma / digamma tabulated once) below two Newton steps from the previous value towards the mode;
 * ANY x0 > 0 with r > 0 gives a valid envelope, a good one gives ~0.95 acceptance at the default hyper-parameters.
 * c <= 1, r <= 0 or 64 rejections in a row fall back to the general 3-tangent sampler orc_ralpha (same stream). */
#define ORC_ALUT_I0 (1013 << 6)        /* 2^-10 <= 1e-3 */
#define ORC_ALUT_N (24 * 64)           /* up to 2^14 > 1e4 */
static double orc_alut[3 * ORC_ALUT_N];   /* lgamma, digamma at the grid point; slope of the chord of lgamma to the next one */
static inline double orc_alut_x(int i) { return orc_u2d((uint64_t)((uint32_t)(i + ORC_ALUT_I0) << 14) << 32); }
static inline int orc_alut_idx(double x) {
  int i = (int)((uint32_t)(orc_d2u(x) >> 32) >> 14) - ORC_ALUT_I0;
  return i < 0 ? 0 : (i > ORC_ALUT_N - 1 ? ORC_ALUT_N - 1 : i);
}
__attribute__((constructor)) static void orc_alut_init(void) {
  for (int i = 0; i < ORC_ALUT_N; ++i) orc_lgamma_digamma(orc_alut_x(i), &orc_alut[3 * i], &orc_alut[3 * i + 1]);
  for (int i = 0; i < ORC_ALUT_N; ++i)
    orc_alut[3 * i + 2] = i + 1 < ORC_ALUT_N ? (orc_alut[3 * (i + 1)] - orc_alut[3 * i]) / (orc_alut_x(i + 1) - orc_alut_x(i)) : orc_alut[3 * i + 1];
}
#define ORC_FAST_ATTEMPTS 64
static inline double orc_ralpha_fast(orc_stream* s, double c, double tau, double xprev, int* n_attempts) {
  const double L = 1e-3, U = 1e4;
  if (!(c > 1.0)) return orc_ralpha(s, c, tau, xprev, n_attempts);
  double x = xprev;
  if (!(x >= L)) x = L;
  if (x > U) x = U;
  const double cm1 = c - 1.0;
  for (int it = 0; it < 12; ++it) {    /* H(x) = (c-1)/x - tau - psi(x), psi from the table, psi'(x) ~ 1/x + 1/x^2 */
    const int i = orc_alut_idx(x);
    const double xg = orc_alut_x(i), psi = orc_alut[3 * i + 1];
    const double inv = 1.0 / xg;
    const double H = (cm1 * inv - tau) - psi;
    const double dH = -cm1 * (inv * inv) - (inv + inv * inv);
    double xn = xg - H / dH;
    if (!(xn > 0.1 * xg)) xn = 0.1 * xg;
    if (xn > 10.0 * xg) xn = 10.0 * xg;
    if (xn < L) xn = L;
    if (xn > U) xn = U;
    const double dx = fabs(xn - xg);
    x = xn;
    if (dx <= 0.03 * xg) break;
  }
  const int i0 = orc_alut_idx(x);
  const double x0 = orc_alut_x(i0), lg0 = orc_alut[3 * i0], psi0 = orc_alut[3 * i0 + 1];
  const double r = tau + psi0;
  /* expected acceptance ~ 1 / sqrt(1 + rho), rho = psi'(x0) var(x): a broad or skewed target (small c) goes to the general sampler */
  const double i0v = 1.0 / x0, tri = i0v + i0v * i0v;
  const double rho = tri / (cm1 * (i0v * i0v) + tri);
  if (!(r > 0.0) || !(rho < 0.35)) return orc_ralpha(s, c, tau, xprev, n_attempts);
  const double d = c - 0.333333333333333333333;
  const double cc = 1.0 / sqrt(9.0 * d);
  const double b0 = lg0 - psi0 * x0;
  for (int it = 0; it < ORC_FAST_ATTEMPTS; ++it) {
    uint32_t w[4]; orc_stream_next(s, w);
    const double z = orc_qnorm(orc_u52(w[0], w[1]));
    const double u = orc_u52(w[2], w[3]);
    const double cz = cc * z;
    double v = 1.0 + cz;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double xs = (d * v) / r;
    if (!(xs >= L && xs <= U)) continue;
    /* lgamma(xs) between its tangent at the grid point below xs and its chord to the next one (lgamma is convex): most
     * attempts are decided without evaluating it */
    const double lv = fabs(cz) <= 0.25 ? 3.0 * orc_log1p_small(cz) : orc_log(v);   /* log v */
    const double a = (0.5 * (z * z) + d * ((1.0 - v) + lv)) + (b0 + psi0 * xs);
    const int ix = orc_alut_idx(xs);
    const double dx = xs - orc_alut_x(ix);
    const int grid = dx >= 0.0 && ix + 1 < ORC_ALUT_N;
    const double tc = a - (orc_alut[3 * ix] + orc_alut[3 * ix + 2] * dx);   /* a - chord   <= a - lgamma(xs) */
    const double tt = a - (orc_alut[3 * ix] + orc_alut[3 * ix + 1] * dx);   /* a - tangent >= a - lgamma(xs) */
    /* (u - 1) / u <= log(u) <= u - 1: the two grid tests without the logarithm when the bounds already decide them */
    const double um1 = u - 1.0;
    int accept;
    if (grid && um1 < tc) accept = 1;
    else if (grid && um1 / u >= tt) accept = 0;
    else {
      const double lu = orc_log(u);
      if (grid && lu < tc) accept = 1;                     /* below a - chord */
      else if (grid && lu >= tt) accept = 0;               /* at / above a - tangent */
      else accept = lu < a - orc_lgamma(xs);
    }
    if (accept) { if (n_attempts) *n_attempts = it + 1; return xs; }
  }
  int na = 0;
  const double xs = orc_ralpha(s, c, tau, xprev, &na);
  if (n_attempts) *n_attempts = ORC_FAST_ATTEMPTS + na;
  return xs;
}

#endif
