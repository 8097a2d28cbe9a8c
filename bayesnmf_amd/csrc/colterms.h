// bayesnmf_amd/csrc/colterms.h — the per-column metric terms of the Poisson models (RMSE, log-likelihood and KL partial sums of a
// column of M against Mhat) for the allocation kernels that leave Mhat behind instead of summing the terms themselves (zalloc_sort.h).
#pragma once
#include "dmath.h"

namespace bnmf {

constexpr int ZS_MCOL = 2;           // columns per wavefront at a time (independent chains in flight: the work is latency-bound)
// ---- the per-column metric terms from the Mhat the allocation kernel left (compute_metrics_ R/utils.R:412-455, get_loglik_ :62-112, padded_KL_ :467-471) ----
// Columns g0 and g0 + 1 by one wavefront, lane = row: squared error, Poisson log-likelihood term (Mhat clipped at 1e-6, :100) and KL term
// of every cell, summed over the rows in the canonical W = 64 order (accumulator l adds rows l, l + 64, ... from +0.0, then the halving
// tree).  The operations of rounds 3-4's metric tasks on the same values: the same bits.  A last pass of at most 32 rows (K = 96: rows
// 64..95) is shared by the two columns, lanes 0..31 / 32..63, and column 1's terms come down with v_permlane32_swap to the lanes whose
// accumulators they belong to — the same additions in the same order without half a wave idling through a pass.
struct CtArgs { const double* mh; const int32_t* M; const double *lgfact, *logm; double *sse, *ll, *kl; int K, G, maxM; };
BNMF_DEV void colterms_pair(const CtArgs& a, int g0, int lane) {
  const int K = a.K, KR = (K + 63) >> 6;
  const int gA = g0, gB = min(g0 + 1, a.G - 1);
  double a_sse[ZS_MCOL], a_ll[ZS_MCOL], a_kl[ZS_MCOL];
#pragma unroll
  for (int j = 0; j < ZS_MCOL; ++j) { a_sse[j] = 0.0; a_ll[j] = 0.0; a_kl[j] = 0.0; }
  const int KL = K - ((KR - 1) << 6);
  const bool split = ZS_MCOL == 2 && KL <= 32;
  for (int r = 0; r < KR - (split ? 1 : 0); ++r) {
    const int kk = (r << 6) + lane;
    if (kk < K) {
      int m[ZS_MCOL];
      double lgf[ZS_MCOL], lgm[ZS_MCOL], c[ZS_MCOL];
#pragma unroll
      for (int j = 0; j < ZS_MCOL; ++j) {
        const size_t at = (size_t)kk + (size_t)K * (size_t)(j ? gB : gA);
        m[j] = a.M[at];
        c[j] = a.mh[at];
        const int mi = m[j] < 0 ? 0 : (m[j] > a.maxM ? a.maxM : m[j]);
        lgf[j] = a.lgfact[mi]; lgm[j] = a.logm[mi];
      }
#pragma unroll
      for (int j = 0; j < ZS_MCOL; ++j) {
        const double dd = c[j] - (double)m[j];
        const double mh = c[j] < 1e-6 ? 1e-6 : c[j];
        const double lmh = dlog(mh);
        const double mt = m[j] < 1 ? 1e-6 : (double)m[j];
        a_sse[j] = a_sse[j] + dd * dd;
        a_ll[j] = a_ll[j] + (((double)m[j] * lmh - mh) - lgf[j]);
        a_kl[j] = a_kl[j] + mt * (lgm[j] - lmh);
      }
    }
  }
  if (split) {
    const int jc = lane >> 5, rl = lane & 31;
    const int kk = ((KR - 1) << 6) + rl;
    double tsse = 0.0, tll = 0.0, tkl = 0.0;
    if (rl < KL) {
      const size_t at = (size_t)kk + (size_t)K * (size_t)(jc ? gB : gA);
      const int m = a.M[at];
      const double c = a.mh[at];
      const int mi = m < 0 ? 0 : (m > a.maxM ? a.maxM : m);
      const double lgf = a.lgfact[mi], lgm = a.logm[mi];
      const double dd = c - (double)m;
      const double mh = c < 1e-6 ? 1e-6 : c;
      const double lmh = dlog(mh);
      const double mt = m < 1 ? 1e-6 : (double)m;
      tsse = dd * dd;
      tll = ((double)m * lmh - mh) - lgf;
      tkl = mt * (lgm - lmh);
    }
    const double u0 = down32(tsse), u1 = down32(tll), u2 = down32(tkl);
    if (lane < 32 && rl < KL) {
      a_sse[0] = a_sse[0] + tsse; a_ll[0] = a_ll[0] + tll; a_kl[0] = a_kl[0] + tkl;
      constexpr int J1 = ZS_MCOL > 1 ? 1 : 0;
      if (ZS_MCOL > 1) { a_sse[J1] = a_sse[J1] + u0; a_ll[J1] = a_ll[J1] + u1; a_kl[J1] = a_kl[J1] + u2; }
    }
  }
#pragma unroll
  for (int j = 0; j < ZS_MCOL; ++j) {
    const double r0 = wave_tree64(a_sse[j]), r1 = wave_tree64(a_ll[j]), r2 = wave_tree64(a_kl[j]);
    if (lane == 0 && g0 + j < a.G) { const int g = g0 + j; a.sse[g] = r0; a.ll[g] = r1; a.kl[g] = r2; }
  }
}
// a launch of its own (the first sweeps of a chain, the two-kernel sweep, the end of a bnmf_run call): four column pairs per workgroup
constexpr int CT_T = 256;
__global__ __launch_bounds__(CT_T) void k_colterms(CtArgs a) {
  const int p = (int)blockIdx.x * (CT_T / 64) + ((int)threadIdx.x >> 6);
  if (2 * p < a.G) colterms_pair(a, 2 * p, (int)threadIdx.x & 63);
}

}  // namespace bnmf
