// bayesnmf_amd/csrc/dmath.h — gfx950 device numerics of the stream spec.
//
// Philox4x32-10, the u52 uniform and fp64 log / exp / lgamma / digamma / qnorm / log Phi
// written only with IEEE +,-,*,/ and sqrt in a fixed order (build with -ffp-contract=off),
// so that results are independent of launch geometry and identical on every device.
// Accuracy: <= 1 ulp (log, exp), <= 6e-15 abs/rel (lgamma, digamma), 1e-15 rel (qnorm).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bnmf {

#define BNMF_DEV __device__ __forceinline__

// ---------------------------------------------------------------- Philox4x32 (Salmon et al., SC'11), R rounds
// Ten rounds (the authors' default) for every stream but the count-allocation words, which take seven: the smallest round
// count the authors report as Crush-resistant (it passes BigCrush), and 97 % of all the words an iteration draws.
struct u32x4 { uint32_t x, y, z, w; };

template <int R>
BNMF_DEV u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;   // v_mad_u64_u32
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);   // a ^ b ^ c in one v_bitop3_b32 (gfx950)
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}
BNMF_DEV u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) { return philox4x32<10>(c0, c1, c2, c3, k0, k1); }
BNMF_DEV u32x4 philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) { return philox4x32<7>(c0, c1, c2, c3, k0, k1); }

// counter = (block, element, iteration, variable id); key = (seed_lo, seed_hi ^ chain)
struct Stream {
  uint32_t k0, k1, elem, iter, var, blk;
  BNMF_DEV Stream(uint32_t k0_, uint32_t k1_, uint32_t var_, uint32_t elem_, uint32_t iter_)
      : k0(k0_), k1(k1_), elem(elem_), iter(iter_), var(var_), blk(0) {}
  BNMF_DEV u32x4 next() { u32x4 w = philox4x32_10(blk, elem, iter, var, k0, k1); ++blk; return w; }
};

BNMF_DEV double u52(uint32_t a, uint32_t b) {
  const uint64_t x = ((uint64_t)(a >> 6) << 26) | (uint64_t)(b >> 6);
  return ((double)x + 0.5) * 2.220446049250313080847263336181640625e-16;
}

BNMF_DEV double dsqrt(double x) { return __dsqrt_rn(x); }
BNMF_DEV double dabs(double x) { return __builtin_fabs(x); }
#define BNMF_INF (__builtin_inf())
#define BNMF_NAN (__builtin_nan(""))

// ---------------------------------------------------------------------- log
BNMF_DEV double dlog(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  if (x != x) return x;
  if (x <= 0.0) return (x == 0.0) ? -BNMF_INF : BNMF_NAN;
  if (x == BNMF_INF) return x;
  int k = 0;
  uint64_t u = (uint64_t)__double_as_longlong(x);
  if ((u >> 52) == 0) { x = x * 18014398509481984.0; u = (uint64_t)__double_as_longlong(x); k = -54; }
  uint32_t hx = (uint32_t)(u >> 32);
  const uint32_t lx = (uint32_t)u;
  k += (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  const uint32_t i = (hx + 0x95f64u) & 0x100000u;
  u = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | lx;
  k += (int)(i >> 20);
  const double m = __longlong_as_double((long long)u);
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// log(1 + f) for |f| <= 0.25: dlog's kernel (its k = 0 branch) applied to f directly — no exponent extraction, no special cases.
// The Marsaglia-Tsang test needs log((1 + c z)^3) = 3 log(1 + c z) with c z small whenever the shape is not.
BNMF_DEV double dlog1p_small(double f) {
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  return f - (hfsq - s * (hfsq + R));
}

// dlog for a positive, finite, NORMAL argument: the same operations as dlog (hence the same bits) without the
// special-case branches, so that independent evaluations can be interleaved by the compiler
BNMF_DEV double dlog_fin(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t u = (uint64_t)__double_as_longlong(x);
  uint32_t hx = (uint32_t)(u >> 32);
  const uint32_t lx = (uint32_t)u;
  int k = (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  const uint32_t i = (hx + 0x95f64u) & 0x100000u;
  u = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | lx;
  k += (int)(i >> 20);
  const double m = __longlong_as_double((long long)u);
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// ---------------------------------------------------------------------- exp
BNMF_DEV double dexp(double x) {
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893383973096) return BNMF_INF;
  if (x < -745.13321910194110842) return 0.0;
  int k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
  const double t = (double)k;
  const double hi = x - t * ln2HI;
  const double lo = t * ln2LO;
  const double r = hi - lo;
  const double tt = r * r;
  const double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (k > 1023) { y = y * 8.98846567431157953865e307; k -= 1023; }
  if (k < -1022) { y = y * 2.22507385850720138309e-308; k += 1022;
    if (k < -1022) { y = y * 2.22507385850720138309e-308; k += 1022; } }
  return y * __longlong_as_double((long long)((uint64_t)(1023 + k) << 52));
}

// ------------------------------------------------------ lgamma / digamma, x > 0
template <bool WANT_DG>
BNMF_DEV void lgamma_digamma(double x, double& lg, double& dg) {
  const double HALF_LOG_2PI = 0.91893853320467274178;
  if (!(x > 0.0)) { lg = (x == 0.0) ? BNMF_INF : BNMF_NAN; dg = BNMF_NAN; return; }
  double prod = 1.0, rs = 0.0, xs = x;
  while (xs < 8.0) { prod = prod * xs; if (WANT_DG) rs = rs + 1.0 / xs; xs = xs + 1.0; }
  const double lxs = dlog(xs);
  const double w = 1.0 / xs, w2 = w * w;
  const double ser = w * (8.33333333333333333333e-02 + w2 * (-2.77777777777777777778e-03 +
                     w2 * (7.93650793650793650794e-04 + w2 * (-5.95238095238095238095e-04 +
                     w2 * (8.41750841750841750842e-04 + w2 * (-1.91752691752691752692e-03 +
                     w2 * 6.41025641025641025641e-03))))));
  lg = (((xs - 0.5) * lxs - xs) + HALF_LOG_2PI) + ser - dlog(prod);
  if (WANT_DG) {
    const double ds = w2 * (8.33333333333333333333e-02 - w2 * (8.33333333333333333333e-03 -
                      w2 * (3.96825396825396825397e-03 - w2 * (4.16666666666666666667e-03 -
                      w2 * (7.57575757575757575758e-03 - w2 * (2.10927960927960927961e-02 -
                      w2 * 8.33333333333333333333e-02))))));
    dg = ((lxs - 0.5 * w) - ds) - rs;
  }
}
BNMF_DEV double dlgamma(double x) { double a, b; lgamma_digamma<false>(x, a, b); return a; }
BNMF_DEV double ddigamma(double x) { double a, b; lgamma_digamma<true>(x, a, b); return b; }

// -------------------------------------------------------------------- qnorm
// Central part (99.9 % of the uniforms: w = -log(4 p (1 - p)) < 6.25): Phi^-1(p) = y F(w), y = 2 p - 1, F a degree-24 polynomial in
// w - 3.125 (the form of Giles' erfinv approximation; coefficients from a Chebyshev interpolation of sqrt(2) erfinv(y) / y at
// 60 digits, tools/fit_qnorm.py; 3.5e-16 against mpmath) — one logarithm and a Horner chain, where AS241's two inner branches
// (rational | logarithm, root, rational) were both executed by nearly every wavefront.  Tails: AS241's outer branches.
BNMF_DEV double dqnorm(double p) {
  const double q = p - 0.5;
  double r, val;
  {
    const double y = q + q;
    const double w = -dlog_fin((1.0 - y) * (1.0 + y));   // argument in [2^-51, 1]: positive, normal, finite — the same bits as dlog
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
  r = dsqrt(-dlog(r));
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

// ------------------------------------------- exp(y^2) erfc(y), y >= 0 (Cody 1969) and log Phi
BNMF_DEV double derfcx(double y) {
  if (y <= 0.46875) {
    const double a0 = 3.16112374387056560e00, a1 = 1.13864154151050156e02, a2 = 3.77485237685302021e02,
                 a3 = 3.20937758913846947e03, a4 = 1.85777706184603153e-1;
    const double b0 = 2.36012909523441209e01, b1 = 2.44024637934444173e02, b2 = 1.28261652607737228e03,
                 b3 = 2.84423683343917062e03;
    const double ysq = y * y;
    double xnum = a4 * ysq, xden = ysq;
    xnum = (xnum + a0) * ysq; xden = (xden + b0) * ysq;
    xnum = (xnum + a1) * ysq; xden = (xden + b1) * ysq;
    xnum = (xnum + a2) * ysq; xden = (xden + b2) * ysq;
    const double erf = y * (xnum + a3) / (xden + b3);
    return (1.0 - erf) * dexp(ysq);
  } else if (y <= 4.0) {
    const double c[9] = {5.64188496988670089e-1, 8.88314979438837594e00, 6.61191906371416295e01,
                         2.98635138197400131e02, 8.81952221241769090e02, 1.71204761263407058e03,
                         2.05107837782607147e03, 1.23033935479799725e03, 2.15311535474403846e-8};
    const double d[8] = {1.57449261107098347e01, 1.17693950891312499e02, 5.37181101862009858e02,
                         1.62138957456669019e03, 3.29079923573345963e03, 4.36261909014324716e03,
                         3.43936767414372164e03, 1.23033935480374942e03};
    double xnum = c[8] * y, xden = y;
#pragma unroll
    for (int i = 0; i < 7; ++i) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
    return (xnum + c[7]) / (xden + d[7]);
  } else {
    const double p[6] = {3.05326634961232344e-1, 3.60344899949804439e-1, 1.25781726111229246e-1,
                         1.60837851487422766e-2, 6.58749161529837803e-4, 1.63153871373020978e-2};
    const double q[5] = {2.56852019228982242e00, 1.87295284992346725e00, 5.27905102951428412e-1,
                         6.05183413124413191e-2, 2.33520497626869185e-3};
    const double sqrpi = 5.6418958354775628695e-1;
    const double ysq = 1.0 / (y * y);
    double xnum = p[5] * ysq, xden = ysq;
#pragma unroll
    for (int i = 0; i < 4; ++i) { xnum = (xnum + p[i]) * ysq; xden = (xden + q[i]) * ysq; }
    const double r = ysq * (xnum + p[4]) / (xden + q[4]);
    return (sqrpi - r) / y;
  }
}
BNMF_DEV double dlog_pnorm(double z) {
  const double SQRT1_2 = 0.70710678118654752440;
  if (z != z) return z;
  if (z >= 0.0) {
    const double y = z * SQRT1_2;
    const double e = 0.5 * derfcx(y) * dexp(-(y * y));
    const double u = 1.0 - e;
    if (u == 1.0) return -e;
    return dlog(u) * (-e) / (u - 1.0);
  } else {
    const double y = -z * SQRT1_2;
    return dlog(0.5 * derfcx(y)) - y * y;
  }
}

// ----------------------------------------------------------- canonical reductions
// canon_sum(x, L, W): accumulator i adds x[i], x[i+W], ... in order from +0.0, then the
// halving tree acc[i] += acc[i+h], h = W/2..1.  W = 64: one wavefront; larger W: a workgroup.
BNMF_DEV double down32(double v) {                        // lane l < 32 gets lane l + 32's value
  const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)b[1] << 32) | (unsigned)a[1]);
}
// The same halving tree (lane i adds lane i + h for h = 32, 16, 8, 4, 2, 1: same operands, same bits as the __shfl_down
// form) without the LDS crossbar: gfx950's v_permlane32_swap / v_permlane16_swap bring lanes i + 32 / i + 16 down, DPP
// row shifts do the rest.  A tree is ~20 VALU instructions instead of 12 dependent ds_bpermute round trips.
BNMF_DEV double wave_tree64(double v) {   // lane 0 gets the W=64 halving tree of v over the wave
  {
    const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = v + __longlong_as_double(((long long)b[1] << 32) | (unsigned)a[1]);
  }
  {
    const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = v + __longlong_as_double(((long long)b[1] << 32) | (unsigned)a[1]);
  }
#define BNMF_TREE_STEP(CTRL)                                                                                       \
  {                                                                                                                \
    int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32);                              \
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true); \
    v = v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);                                            \
  }
  BNMF_TREE_STEP(0x108) BNMF_TREE_STEP(0x104) BNMF_TREE_STEP(0x102) BNMF_TREE_STEP(0x101)   // row_shl:8, 4, 2, 1
#undef BNMF_TREE_STEP
  return v;
}
// lane 0's value to every lane through the scalar unit (v_readlane: no LDS crossbar); all lanes of the wave must be active
BNMF_DEV double wave_bcast0(double v) {
  const int lo = __builtin_amdgcn_readlane((int)__double_as_longlong(v), 0), hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), 0);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// block tree over NT (<=1024, power of two) values through LDS `buf` (NT doubles); result in buf[0]
template <int NT>
BNMF_DEV double block_tree(double v, double* buf, int tid) {
  buf[tid] = v;
  __syncthreads();
#pragma unroll
  for (int h = NT / 2; h >= 64; h >>= 1) {
    if (tid < h) buf[tid] = buf[tid] + buf[tid + h];
    __syncthreads();
  }
  double r = 0.0;
  if (tid < 64) { r = buf[tid]; r = wave_tree64(r); if (NT < 128) {} }
  return r;   // valid on thread 0
}

}  // namespace bnmf
