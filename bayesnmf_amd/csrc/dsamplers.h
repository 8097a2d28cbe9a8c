// bayesnmf_amd/csrc/dsamplers.h — per-lane scalar samplers of the stream spec.
// Each sampler consumes whole Philox blocks of its own (variable, element, iteration) stream,
// so a draw never depends on which lane, wave or workgroup produced it.
//   Gamma        : Marsaglia-Tsang, normal by inversion    (stats::rgamma call sites: R/sample_Pn.R:117,
//                  R/sample_En.R:116, R/sample_priors.R:72-127,285-344)
//   TN(mu,sd;0,inf): normal rejection / Robert's exponential rejection (truncnorm::rtruncnorm,
//                  R/sample_Pn.R:14,59,79; R/sample_En.R:14,59,78)
//   Alpha        : exact 3-tangent rejection for the log-concave ARMS target of
//                  R/sample_priors.R:356-397
#pragma once
#include "dmath.h"

namespace bnmf {

constexpr int MAX_ATTEMPTS = 2000;

BNMF_DEV double rnorm_std(Stream& s) { const u32x4 w = s.next(); return dqnorm(u52(w.x, w.y)); }
BNMF_DEV double runif(Stream& s) { const u32x4 w = s.next(); return u52(w.x, w.y); }
BNMF_DEV double rexp(Stream& s, double rate) { const u32x4 w = s.next(); return -dlog(u52(w.x, w.y)) / rate; }

#ifdef ZSPROF
constexpr int DRPROF_W = 4096;
__device__ unsigned long long g_drprof[8 * DRPROF_W];   // diagnostics: section ticks of k_draw's E waves, [wave][8]; the last row's [4]: lanes that left ralpha_fast for the general sampler
#ifdef ZSLIGHT   /* stamps only: no counters inside the samplers (they cost the profile build a factor of eight) */
#define FB_COUNT
#define PCOUNT(w, l)
#else
#define FB_COUNT atomicAdd(&g_drprof[8 * (DRPROF_W - 1) + 4], 1ull)
// wave-level pass counters (first active lane counts) and lane-level attempt counters, last row: [0] Newton wave-iterations,
// [1] Alpha wave-passes, [2] Alpha lane-attempts, [3] Newton lane-iterations, [5] rgamma wave-passes, [6] rgamma lane-attempts
#define PCOUNT(w, l) do { if ((int)(threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_drprof[8 * (DRPROF_W - 1) + (w)], 1ull); \
                          atomicAdd(&g_drprof[8 * (DRPROF_W - 1) + (l)], 1ull); } while (0)
#endif
#else
#define FB_COUNT
#define PCOUNT(w, l)
#endif
// RETRY_PRIO (k_draw): a wavefront that goes round the rejection loop again raises its issue priority for the rest of the draw — it is
// the wave its workgroup (and the kernel) will end with, and the SIMD's other waves are a pass ahead of it.  Timing only.
template <bool RETRY_PRIO = false>
BNMF_DEV double rgamma(Stream& s, double a, double rate) {
  if (!(a > 0.0)) return (a == 0.0) ? 0.0 : BNMF_NAN;
  const bool boost = a < 1.0;
  const double a1 = boost ? a + 1.0 : a;
  const double d = a1 - 0.333333333333333333333;
  const double c = 1.0 / dsqrt(9.0 * d);
  double v = 1.0;
  for (int it = 0; it < MAX_ATTEMPTS; ++it) {
    PCOUNT(5, 6);
    if (RETRY_PRIO && it == 1) __builtin_amdgcn_s_setprio(3);
    const u32x4 w = s.next();
    const double z = dqnorm(u52(w.x, w.y));
    const double ua = u52(w.z, w.w);
    const double cz = c * z;
    v = 1.0 + cz;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double z2 = z * z;
    if (ua < 1.0 - 0.0331 * (z2 * z2)) break;
    // accept iff log(ua) < R.  (ua - 1) / ua <= log(ua) <= ua - 1 decides all but a sliver of width ~ (1 - ua)^2 without the
    // logarithm (a wavefront pays for it whenever one lane needs it)
    const double lv = dabs(cz) <= 0.25 ? 3.0 * dlog1p_small(cz) : dlog(v);   // log v
    const double R = 0.5 * z2 + d * ((1.0 - v) + lv);
    const double um1 = ua - 1.0;
    if (um1 < R) break;
    if (um1 / ua >= R) continue;
    if (dlog(ua) < R) break;
  }
  if (RETRY_PRIO) __builtin_amdgcn_s_setprio(0);
  double g = d * v;
  if (boost) {
    const u32x4 w = s.next();
    g = g * dexp(dlog(u52(w.x, w.y)) / a);
  }
  return g / rate;
}
BNMF_DEV double rinvgamma(Stream& s, double shape, double rate) { return 1.0 / rgamma(s, shape, rate); }

BNMF_DEV double rtnorm0(Stream& s, double mu, double sd) {
  const double alpha = -mu / sd;
  double z = alpha;
  if (alpha < 0.45) {
    for (int it = 0; it < MAX_ATTEMPTS; ++it) {
      const u32x4 w = s.next();
      z = dqnorm(u52(w.x, w.y));
      if (z >= alpha) break;
    }
  } else {
    const double lam = 0.5 * (alpha + dsqrt(alpha * alpha + 4.0));
    for (int it = 0; it < MAX_ATTEMPTS; ++it) {
      const u32x4 w = s.next();
      const double e = -dlog(u52(w.x, w.y)) / lam;
      z = alpha + e;
      const double t = z - lam;
      const double rho = dexp(-0.5 * (t * t));
      if (u52(w.z, w.w) <= rho) break;
    }
  }
  const double x = mu + sd * z;
  return x < 0.0 ? 0.0 : x;
}

// rtnorm0 with the parts of its first attempt that need neither mu nor sd taken out: the block's first uniform through the
// normal quantile and through the logarithm, and its second uniform.  The sequential sweeps of the MH / Normal models draw one
// truncated normal per factor inside a chain of dependent steps; these three values per factor are made before the chain starts
// (all factors at once, one per lane).  rtnorm0_pre(s, rt_pre(...), mu, sd) performs rtnorm0's operations in rtnorm0's order.
struct RtPre { double z, lu, u2; };
BNMF_DEV RtPre rt_pre(uint32_t k0, uint32_t k1, uint32_t var, uint32_t elem, uint32_t iter) {
  const u32x4 w = philox4x32_10(0u, elem, iter, var, k0, k1);
  const double u1 = u52(w.x, w.y);
  return RtPre{dqnorm(u1), dlog(u1), u52(w.z, w.w)};
}
BNMF_DEV double rtnorm0_pre(Stream& s /* of the same (variable, element, iteration), at block 0 */, const RtPre& p, double mu, double sd) {
  const double alpha = -mu / sd;
  double z = alpha;
  s.blk = 1;
  if (alpha < 0.45) {
    z = p.z;
    if (!(z >= alpha))
      for (int it = 1; it < MAX_ATTEMPTS; ++it) {
        const u32x4 w = s.next();
        z = dqnorm(u52(w.x, w.y));
        if (z >= alpha) break;
      }
  } else {
    const double lam = 0.5 * (alpha + dsqrt(alpha * alpha + 4.0));
    const double e0 = -p.lu / lam;
    z = alpha + e0;
    const double t0 = z - lam;
    const double rho0 = dexp(-0.5 * (t0 * t0));
    if (!(p.u2 <= rho0))
      for (int it = 1; it < MAX_ATTEMPTS; ++it) {
        const u32x4 w = s.next();
        const double e = -dlog(u52(w.x, w.y)) / lam;
        z = alpha + e;
        const double t = z - lam;
        const double rho = dexp(-0.5 * (t * t));
        if (u52(w.z, w.w) <= rho) break;
      }
  }
  const double x = mu + sd * z;
  return x < 0.0 ? 0.0 : x;
}

// log f(x) = (c-1) log x - tau x - lgamma(x) and its derivative
BNMF_DEV void alpha_h(double x, double c, double tau, double& h, double& hp) {
  double lg, dg;
  lgamma_digamma<true>(x, lg, dg);
  h = ((c - 1.0) * dlog(x) - tau * x) - lg;
  hp = ((c - 1.0) / x - tau) - dg;
}

BNMF_DEV double ralpha(Stream& s, double c, double tau, double xprev, int* n_attempts = nullptr) {
  const double L = 1e-3, U = 1e4, DELTA = 1.41421356237309504880;
  double x = xprev;
  if (!(x >= L)) x = L;
  if (x > U) x = U;
  for (int it = 0; it < 32; ++it) {
    const double Hx = (c / x - tau) - dlog(x + 0.5);
    const double dH = -c / (x * x) - 1.0 / (x + 0.5);
    double xn = x - Hx / dH;
    if (!(xn > 0.1 * x)) xn = 0.1 * x;
    if (xn > 10.0 * x) xn = 10.0 * x;
    if (xn < L) xn = L;
    if (xn > U) xn = U;
    const double dx = dabs(xn - x);
    x = xn;
    if (dx <= 1e-3 * x) break;
  }
  double m = x;
  const double g = (c / m - tau) - dlog(m + 0.5);
  const double kap = c / (m * m) + 1.0 / (m + 0.5);
  double sc = 1.0 / dsqrt(kap);
  if (dabs(g) * sc > 1.0) sc = 1.0 / dabs(g);
  double x1, x2, x3, h1, s1, h2, s2, h3, s3;
  for (int round = 0; round < 4; ++round) {
    x1 = m - DELTA * sc; x2 = m; x3 = m + DELTA * sc;
    if (x1 < L) x1 = L;
    if (x3 > U) x3 = U;
    if (x2 - x1 < 0.25 * sc) { x1 = L; x2 = L + 0.75 * sc; x3 = L + 2.5 * sc; }
    else if (x3 - x2 < 0.25 * sc) { x3 = U; x2 = U - 0.75 * sc; x1 = U - 2.5 * sc; }
    if (x1 < L) x1 = L;
    if (x3 > U) x3 = U;
    if (!(x1 < x2 && x2 < x3)) { x1 = L; x2 = 0.5 * (L + U); x3 = U; }
    alpha_h(x1, c, tau, h1, s1);
    alpha_h(x2, c, tau, h2, s2);
    alpha_h(x3, c, tau, h3, s3);
    bool walked = false;
    for (int it = 0; it < 64 && s3 > 0.0 && x3 < U; ++it) {
      const double step = 2.0 * (x3 - x2);
      x1 = x2; h1 = h2; s1 = s2; x2 = x3; h2 = h3; s2 = s3;
      x3 = x3 + step; if (x3 > U) x3 = U;
      alpha_h(x3, c, tau, h3, s3);
      walked = true;
    }
    for (int it = 0; it < 64 && s1 < 0.0 && x1 > L; ++it) {
      const double step = 2.0 * (x2 - x1);
      x3 = x2; h3 = h2; s3 = s2; x2 = x1; h2 = h1; s2 = s1;
      x1 = x1 - step; if (x1 < L) x1 = L;
      alpha_h(x1, c, tau, h1, s1);
      walked = true;
    }
    if (!walked) break;
    double xa, xb, sa, sb;
    if (s2 > 0.0) { xa = x2; sa = s2; xb = x3; sb = s3; } else { xa = x1; sa = s1; xb = x2; sb = s2; }
    if (!(sa > 0.0 && sb < 0.0)) break;
    m = xa + sa * (xb - xa) / (sa - sb);
    sc = dsqrt((xb - xa) / (sa - sb));
    if (!(sc > 0.0)) break;
  }
  double z1 = 0.5 * (x1 + x2), z2 = 0.5 * (x2 + x3);
  if (s1 - s2 > 1e-14 * (dabs(s1) + dabs(s2))) {
    z1 = (((h2 - h1) - s2 * x2) + s1 * x1) / (s1 - s2);
    if (!(z1 >= x1)) z1 = x1;
    if (z1 > x2) z1 = x2;
  }
  if (s2 - s3 > 1e-14 * (dabs(s2) + dabs(s3))) {
    z2 = (((h3 - h2) - s3 * x3) + s2 * x2) / (s2 - s3);
    if (!(z2 >= x2)) z2 = x2;
    if (z2 > x3) z2 = x3;
  }
  // three segments [L,z1],[z1,z2],[z2,U]; kept in named scalars (no runtime-indexed arrays)
  const double lo0 = L, hi0 = z1, lo1 = z1, hi1 = z2, lo2 = z2, hi2 = U;
  const double Tlo0 = h1 + s1 * (lo0 - x1), Thi0 = h1 + s1 * (hi0 - x1);
  const double Tlo1 = h2 + s2 * (lo1 - x2), Thi1 = h2 + s2 * (hi1 - x2);
  const double Tlo2 = h3 + s3 * (lo2 - x3), Thi2 = h3 + s3 * (hi2 - x3);
  double ref = -BNMF_INF;
  if (Tlo0 > ref) ref = Tlo0;
  if (Thi0 > ref) ref = Thi0;
  if (Tlo1 > ref) ref = Tlo1;
  if (Thi1 > ref) ref = Thi1;
  if (Tlo2 > ref) ref = Tlo2;
  if (Thi2 > ref) ref = Thi2;
  auto area = [&](double Tlo, double Thi, double sj, double wj) -> double {
    const double sw = sj * wj;
    double A;
    if (dabs(sw) < 1e-6) A = dexp(Tlo - ref) * wj * (1.0 + 0.5 * sw);
    else A = (dexp(Thi - ref) - dexp(Tlo - ref)) / sj;
    if (!(A > 0.0)) A = 0.0;
    return A;
  };
  const double A0 = area(Tlo0, Thi0, s1, hi0 - lo0);
  const double A1 = area(Tlo1, Thi1, s2, hi1 - lo1);
  const double A2 = area(Tlo2, Thi2, s3, hi2 - lo2);
  const double Atot = (A0 + A1) + A2;
  double xs = m;
  int it = 0;
  for (; it < MAX_ATTEMPTS; ++it) {
    const u32x4 w = s.next();
    const double ua = u52(w.x, w.y) * Atot;
    const double u2 = u52(w.z, w.w);
    double r, lo, hi, sj, hj, xj;
    if (ua < A0) { r = ua / A0; lo = lo0; hi = hi0; sj = s1; hj = h1; xj = x1; }
    else if (ua < A0 + A1) { r = (ua - A0) / A1; lo = lo1; hi = hi1; sj = s2; hj = h2; xj = x2; }
    else { r = ((ua - A0) - A1) / A2; lo = lo2; hi = hi2; sj = s3; hj = h3; xj = x3; }
    if (!(r <= 1.0)) r = 1.0;
    const double wj = hi - lo, sw = sj * wj;
    if (dabs(sw) < 1e-6) xs = lo + r * wj;
    else if (sj > 0.0) xs = hi + dlog(r + (1.0 - r) * dexp(-sw)) / sj;
    else xs = lo + dlog((1.0 - r) + r * dexp(sw)) / sj;
    if (xs < lo) xs = lo;
    if (xs > hi) xs = hi;
    double hx, hpx;
    alpha_h(xs, c, tau, hx, hpx);
    const double Tx = hj + sj * (xs - xj);
    if (dlog(u2) <= hx - Tx) break;
  }
  if (n_attempts) *n_attempts = it + 1;
  return xs;
}


// ---- Gamma-shape hyper-parameter, fast path (what the hyper sweep calls; the CPU checker restates it op for op) ----
// -lgamma is concave: its tangent at x0 gives a Gamma(c, tau + psi(x0)) envelope of f(x) ~ x^(c-1) e^(-tau x) / Gamma(x);
// x ~ Gamma(c, r) by Marsaglia-Tsang, kept with probability exp(lgamma(x0) + psi(x0)(x - x0) - lgamma(x)); one uniform for
// both acceptance tests, one Philox block per attempt.  x0 = grid point (doubles with 6 mantissa bits; lgamma / digamma
// tabulated once per device in g_alut) below two Newton steps from the previous value.  c <= 1, r <= 0 or 64 rejections in
// a row: the general 3-tangent sampler ralpha, on the same stream.
constexpr int ALUT_I0 = 1013 << 6;     // 2^-10 <= 1e-3
constexpr int ALUT_N = 24 * 64;        // up to 2^14 > 1e4
__device__ double g_alut[3 * ALUT_N];   // lgamma, digamma at the grid point; slope of the chord of lgamma to the next one
BNMF_DEV double alut_x(int i) { return __longlong_as_double((long long)((uint64_t)((uint32_t)(i + ALUT_I0) << 14) << 32)); }
BNMF_DEV int alut_idx(double x) {
  const int i = (int)((uint32_t)((uint64_t)__double_as_longlong(x) >> 32) >> 14) - ALUT_I0;
  return i < 0 ? 0 : (i > ALUT_N - 1 ? ALUT_N - 1 : i);
}
__global__ void k_alut_fill(int pass) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ALUT_N) return;
  if (pass == 0) { double lg, dg; lgamma_digamma<true>(alut_x(i), lg, dg); g_alut[3 * i] = lg; g_alut[3 * i + 1] = dg; }
  else g_alut[3 * i + 2] = i + 1 < ALUT_N ? (g_alut[3 * (i + 1)] - g_alut[3 * i]) / (alut_x(i + 1) - alut_x(i)) : g_alut[3 * i + 1];
}
constexpr int FAST_ATTEMPTS = 64;
// What an attempt of the fast path needs beside its Philox block: the Marsaglia-Tsang constants of Gamma(c, r) and the tangent of
// lgamma at the grid point x0.
struct AlphaFast { double d, cc, r, b0, psi0; };
// The part of ralpha_fast in front of its attempts: Newton steps from the previous value towards the mode (psi from the table), the
// grid point x0, the envelope.  false: the element takes the general sampler (c <= 1, r <= 0, a broad or skewed target).
BNMF_DEV bool ralpha_setup(double c, double tau, double xprev, const double* lut, AlphaFast& p) {
  const double L = 1e-3, U = 1e4;
  if (!(c > 1.0)) return false;
  double x = xprev;
  if (!(x >= L)) x = L;
  if (x > U) x = U;
  const double cm1 = c - 1.0;
  for (int it = 0; it < 12; ++it) {    // H(x) = (c-1)/x - tau - psi(x), psi from the table, psi'(x) ~ 1/x + 1/x^2
    PCOUNT(0, 3);
    const int i = alut_idx(x);
    const double xg = alut_x(i), psi = lut[3 * i + 1];
    const double inv = 1.0 / xg;
    const double H = (cm1 * inv - tau) - psi;
    const double dH = -cm1 * (inv * inv) - (inv + inv * inv);
    double xn = xg - H / dH;
    if (!(xn > 0.1 * xg)) xn = 0.1 * xg;
    if (xn > 10.0 * xg) xn = 10.0 * xg;
    if (xn < L) xn = L;
    if (xn > U) xn = U;
    const double dx = dabs(xn - xg);
    x = xn;
    if (dx <= 0.03 * xg) break;
  }
  const int i0 = alut_idx(x);
  const double x0 = alut_x(i0), lg0 = lut[3 * i0], psi0 = lut[3 * i0 + 1];
  const double r = tau + psi0;
  // expected acceptance ~ 1 / sqrt(1 + rho), rho = psi'(x0) var(x): a broad or skewed target (small c) goes to the general sampler
  const double i0v = 1.0 / x0, tri = i0v + i0v * i0v;
  const double rho = tri / (cm1 * (i0v * i0v) + tri);
  if (!(r > 0.0) || !(rho < 0.35)) return false;
  p.d = c - 0.333333333333333333333;
  p.cc = 1.0 / dsqrt(9.0 * p.d);
  p.r = r;
  p.b0 = lg0 - psi0 * x0;
  p.psi0 = psi0;
  return true;
}
// Attempt number `att` of an element = Philox block `att` of its (variable, element, iteration) stream: a pure function of the
// element and the attempt number, so any lane may evaluate it.  true: accepted, xs is the draw.
BNMF_DEV bool ralpha_attempt(uint32_t k0, uint32_t k1, uint32_t var, uint32_t elem, uint32_t iter, uint32_t att, const AlphaFast& p,
                             const double* lut, double& xs_out) {
  const double L = 1e-3, U = 1e4;
  PCOUNT(1, 2);
  const u32x4 w = philox4x32_10(att, elem, iter, var, k0, k1);
  const double z = dqnorm(u52(w.x, w.y));
  const double u = u52(w.z, w.w);
  const double cz = p.cc * z;
  double v = 1.0 + cz;
  if (v <= 0.0) return false;
  v = v * v * v;
  const double xs = (p.d * v) / p.r;
  if (!(xs >= L && xs <= U)) return false;
  // lgamma(xs) between its tangent at the grid point below xs and its chord to the next one (lgamma is convex): most
  // attempts are decided without evaluating it
  const double lv = dabs(cz) <= 0.25 ? 3.0 * dlog1p_small(cz) : dlog(v);   // log v
  const double a = (0.5 * (z * z) + p.d * ((1.0 - v) + lv)) + (p.b0 + p.psi0 * xs);
  const int ix = alut_idx(xs);
  const double dx = xs - alut_x(ix);
  const bool grid = dx >= 0.0 && ix + 1 < ALUT_N;
  const double tc = a - (lut[3 * ix] + lut[3 * ix + 2] * dx);   // a - chord   <= a - lgamma(xs)
  const double tt = a - (lut[3 * ix] + lut[3 * ix + 1] * dx);   // a - tangent >= a - lgamma(xs)
  // (u - 1) / u <= log(u) <= u - 1: the two grid tests without the logarithm when the bounds already decide them
  const double um1 = u - 1.0;
  bool accept;
  if (grid && um1 < tc) accept = true;
  else if (grid && um1 / u >= tt) accept = false;
  else {
    const double lu = dlog(u);
    if (grid && lu < tc) accept = true;                  // below a - chord
    else if (grid && lu >= tt) accept = false;           // at / above a - tangent
    else accept = lu < a - dlgamma(xs);
  }
  xs_out = xs;
  return accept;
}
// lut: the table (g_alut, or a workgroup's copy of it in LDS: k_draw)
BNMF_DEV double ralpha_fast(Stream& s, double c, double tau, double xprev, int* n_attempts = nullptr, const double* lut = g_alut) {
  AlphaFast p;
  if (!ralpha_setup(c, tau, xprev, lut, p)) { FB_COUNT; return ralpha(s, c, tau, xprev, n_attempts); }
  for (int it = 0; it < FAST_ATTEMPTS; ++it) {
    double xs;
    const bool accept = ralpha_attempt(s.k0, s.k1, s.var, s.elem, s.iter, s.blk, p, lut, xs);
    ++s.blk;
    if (accept) { if (n_attempts) *n_attempts = it + 1; return xs; }
  }
  FB_COUNT;
  int na = 0;
  const double xs = ralpha(s, c, tau, xprev, &na);
  if (n_attempts) *n_attempts = FAST_ATTEMPTS + na;
  return xs;
}

// ---- ralpha_fast for a whole wavefront (k_draw): the same draws, a bounded number of passes ----
// An element's draw is its FIRST accepted attempt, and attempt k is a pure function of (element, k).  With one lane per element
// and ~0.94 acceptance nearly every wavefront ran a second pass for its three or four rejected lanes, one in five a third (2.15
// wave-passes for 1.06 attempts per element, and the kernel ended with its unluckiest wave).  Here the lanes whose element is
// decided evaluate the LATER attempts of the undecided ones: with np elements pending each gets the next 64 / np attempts (rounded
// down to a power of two) evaluated side by side, and takes the lowest-numbered accepted one — the attempt the sequential loop would
// have stopped at.  Same bits, two passes (a third with probability ~1e-17 per element).
// Every lane of the wave must call this (act = the lane has an element); lanes go to the general sampler exactly when ralpha_fast does.
BNMF_DEV double ralpha_fast_wave(bool act, uint32_t k0, uint32_t k1, uint32_t var, uint32_t elem, uint32_t iter, double c, double tau, double xprev,
                                 const double* lut) {
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  AlphaFast p{1.0, 1.0, 1.0, 0.0, 0.0};
  int mode = 0;                                          // 0: no element, 1: fast path, 2: general sampler from block 0, 3: ... from block FAST_ATTEMPTS
  if (act) mode = ralpha_setup(c, tau, xprev, lut, p) ? 1 : 2;
  double res = 0.0;
  bool pend = false;
  if (mode == 1) pend = !ralpha_attempt(k0, k1, var, elem, iter, 0u, p, lut, res);
  unsigned long long pm = __builtin_amdgcn_ballot_w64(pend);
  uint32_t nk = 1;
  while (pm != 0ull) {                                   // wave-uniform: every lane takes part in the exchanges below
    __builtin_amdgcn_s_setprio(3);                       // the wave is a pass behind the others of its SIMD (timing only)
    const int np = __builtin_popcountll(pm);
    const int lg = 31 - __builtin_clz((unsigned)(64 / np));          // attempts per pending element this round: 2^lg <= 64 / np
    const int per = 1 << lg;
    const int slot = lane >> lg, off = lane & (per - 1);
    int src = 0;                                         // the slot-th pending lane
    { unsigned long long m = pm; int rk = 0; while (m) { const int b = __builtin_ctzll(m); m &= m - 1ull; src = (slot == rk) ? b : src; ++rk; } }
    AlphaFast q;
    q.d = __shfl(p.d, src); q.cc = __shfl(p.cc, src); q.r = __shfl(p.r, src); q.b0 = __shfl(p.b0, src); q.psi0 = __shfl(p.psi0, src);
    const uint32_t el = (uint32_t)__shfl((int)elem, src);
    const uint32_t att = nk + (uint32_t)off;
    double xh = 0.0;
    bool ok = false;
    if (slot < np && att < (uint32_t)FAST_ATTEMPTS) ok = ralpha_attempt(k0, k1, var, el, iter, att, q, lut, xh);
    const unsigned long long am = __builtin_amdgcn_ballot_w64(ok);
    // the pending lane of rank rho owns the verdicts of lanes [rho 2^lg, (rho + 1) 2^lg)
    const int rho = __builtin_popcountll(pm & ((1ull << lane) - 1ull));
    const unsigned long long mine = pend ? (am >> (rho << lg)) & (per == 64 ? ~0ull : ((1ull << per) - 1ull)) : 0ull;
    const int win = mine ? (rho << lg) + __builtin_ctzll(mine) : lane;
    const double xw = __shfl(xh, win);
    if (mine) { res = xw; pend = false; }
    nk += (uint32_t)per;
    if (pend && nk >= (uint32_t)FAST_ATTEMPTS) { mode = 3; pend = false; }
    pm = __builtin_amdgcn_ballot_w64(pend);
  }
  __builtin_amdgcn_s_setprio(0);
  if (mode >= 2) {
    FB_COUNT;
    Stream s(k0, k1, var, elem, iter);
    s.blk = mode == 3 ? (uint32_t)FAST_ATTEMPTS : 0u;
    res = ralpha(s, c, tau, xprev, nullptr);
  }
  return res;
}

}  // namespace bnmf
