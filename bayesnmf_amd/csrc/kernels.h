// bayesnmf_amd/csrc/kernels.h — hand-written gfx950 kernels of the Gibbs sweep.
//
// One iteration (Poisson likelihood, no MH; R/bayesNMF_sampler.R:273-285) is four launches:
//   k_pside   grid N       : Esum[n] -> hyper sweep of column n -> P[,n] -> Psum[n], log-prior
//   k_eside   grid NG/256  : hyper sweep of E elements -> E -> log-prior partials
//   k_zalloc  grid ~2k     : thresholds in LDS -> categorical allocation of every count ->
//                            ZsumK, ZsumG (+Z) and the per-column RMSE/KL/log-lik terms
//   k_finalize grid 1      : canonical reductions -> metrics row
// All fp64; all cross-lane sums use the canonical orders of dmath.h so results are bitwise
// independent of scheduling.  No MFMA: the path is sampling + reductions.
#pragma once
#include "dsamplers.h"
#include "../../include/bnmf.h"

namespace bnmf {

struct HRef { const double* p; int stride; };   // hyper-prior matrix or broadcast scalar
BNMF_DEV double hy(const HRef& h, int e) { return h.p[(size_t)e * h.stride]; }

struct Dev {
  int K, G, N;
  int prior, likelihood, MH, learning_rank, rank_method, save_Z;
  uint32_t k0, k1;
  int maxM;
  const int32_t* M;
  double *P, *E, *A;
  int* R;
  int32_t *ZsumK, *ZsumG, *Z;
  double *Alpha_p, *Beta_p, *Alpha_e, *Beta_e, *Mu_p, *Sig_p, *Mu_e, *Sig_e, *Lam_p, *Lam_e;
  HRef hA_p, hB_p, hC_p, hD_p, hM_p, hS_p, hA_e, hB_e, hC_e, hD_e, hM_e, hS_e;
  double *Esum, *Psum, *lpPn, *lpE_part, *colsse, *colll, *colkl;
  const double *lgfact, *logm;       // LUTs over m = 0..maxM: lgamma(m+1), log(max(m,1e-6))
  const double* temperature; long n_temperature;
  double* metrics;                   // device rows [row][BNMF_NMETRIC]
};

BNMF_DEV double clamp_tiny(double v) { return (v < 1e-300) ? 1e-300 : v; }

// ---- hyper sweep of one element: R/sample_priors.R:150-200 (element-wise conditionals) ----
template <int SIDE>
BNMF_DEV void hyper_elem(const Dev& d, int e, uint32_t t, double v) {
  if (d.prior == BNMF_GAMMA) {
    const HRef &hA = SIDE ? d.hA_e : d.hA_p, &hB = SIDE ? d.hB_e : d.hB_p;
    const HRef &hC = SIDE ? d.hC_e : d.hC_p, &hD = SIDE ? d.hD_e : d.hD_p;
    double* Al = SIDE ? d.Alpha_e : d.Alpha_p;
    double* Be = SIDE ? d.Beta_e : d.Beta_p;
    Stream s(d.k0, d.k1, SIDE ? BNMF_V_BETA_E : BNMF_V_BETA_P, (uint32_t)e, t);
    const double al_old = Al[e];
    const double b = rgamma(s, hy(hA, e) + al_old, hy(hB, e) + v);        // sample_Beta_*  :323-345
    Be[e] = b;
    const double tau = (hy(hD, e) - dlog(clamp_tiny(b))) - dlog(clamp_tiny(v));
    Stream s2(d.k0, d.k1, SIDE ? BNMF_V_ALPHA_E : BNMF_V_ALPHA_P, (uint32_t)e, t);
    Al[e] = ralpha(s2, hy(hC, e), tau, al_old);                             // sample_Alpha_* :356-397
  } else if (d.prior == BNMF_EXPONENTIAL) {
    const HRef &hA = SIDE ? d.hA_e : d.hA_p, &hB = SIDE ? d.hB_e : d.hB_p;
    double* La = SIDE ? d.Lam_e : d.Lam_p;
    Stream s(d.k0, d.k1, SIDE ? BNMF_V_LAMBDA_E : BNMF_V_LAMBDA_P, (uint32_t)e, t);
    La[e] = rgamma(s, hy(hA, e) + 1.0, hy(hB, e) + v);                      // sample_Lambda_* :284-308
  } else {
    const HRef &hM = SIDE ? d.hM_e : d.hM_p, &hS = SIDE ? d.hS_e : d.hS_p;
    const HRef &hA = SIDE ? d.hA_e : d.hA_p, &hB = SIDE ? d.hB_e : d.hB_p;
    double* Mu = SIDE ? d.Mu_e : d.Mu_p;
    double* Sg = SIDE ? d.Sig_e : d.Sig_p;
    const double sg = Sg[e];
    const double num = hy(hM, e) / hy(hS, e) + v / sg;
    const double den = 1.0 / hy(hS, e) + 1.0 / sg;
    Stream s(d.k0, d.k1, SIDE ? BNMF_V_MU_E : BNMF_V_MU_P, (uint32_t)e, t);
    const double mu = num / den + (1.0 / den) * rnorm_std(s);               // sd = 1/denom (quirk) :214-236
    Mu[e] = mu;
    const double dl = v - mu;
    const double rate = (SIDE ? hy(hA, e) : hy(hB, e)) + (dl * dl) / 2.0;   // A_e for B_e (quirk) :263-270
    Stream s2(d.k0, d.k1, SIDE ? BNMF_V_SIGSQ_E : BNMF_V_SIGSQ_P, (uint32_t)e, t);
    Sg[e] = rinvgamma(s2, hy(hA, e) + 0.5, rate);
  }
}
// prior draw of an element of P / E: R/sample_Pn.R:12-30, R/sample_En.R:12-30
template <int SIDE>
BNMF_DEV double prior_draw(const Dev& d, int e, uint32_t t) {
  Stream s(d.k0, d.k1, SIDE ? BNMF_V_E : BNMF_V_P, (uint32_t)e, t);
  if (d.prior == BNMF_GAMMA) return rgamma(s, (SIDE ? d.Alpha_e : d.Alpha_p)[e], (SIDE ? d.Beta_e : d.Beta_p)[e]);
  if (d.prior == BNMF_EXPONENTIAL) return rexp(s, (SIDE ? d.Lam_e : d.Lam_p)[e]);
  return rtnorm0(s, (SIDE ? d.Mu_e : d.Mu_p)[e], dsqrt((SIDE ? d.Sig_e : d.Sig_p)[e]));
}
// log prior density of an element: R/utils.R:132-175
template <int SIDE>
BNMF_DEV double prior_logdens(const Dev& d, int e, double x) {
  if (d.prior == BNMF_GAMMA) {
    const double al = (SIDE ? d.Alpha_e : d.Alpha_p)[e], be = (SIDE ? d.Beta_e : d.Beta_p)[e];
    return ((al * dlog(be) - dlgamma(al)) + (al - 1.0) * dlog(x)) - be * x;
  }
  if (d.prior == BNMF_EXPONENTIAL) {
    const double la = (SIDE ? d.Lam_e : d.Lam_p)[e];
    return dlog(la) - la * x;
  }
  const double mu = (SIDE ? d.Mu_e : d.Mu_p)[e], sg = dsqrt((SIDE ? d.Sig_e : d.Sig_p)[e]);
  const double zz = (x - mu) / sg;
  return ((-0.91893853320467274178 - dlog(sg)) - 0.5 * (zz * zz)) - dlog_pnorm(mu / sg);
}

// ---- k_pside: one workgroup of 1024 lanes per factor n ----
// sample_Pn_poisson R/sample_Pn.R:98-120 (dispatch :11-42) with the P-side hyper sweep fused in.
constexpr int PS_T = 1024;
__global__ __launch_bounds__(PS_T) void k_pside(Dev d, uint32_t t, int from_prior, int do_hyper) {
  // all LDS in one dynamic, 16-byte aligned region: buf[PS_T] | bc[2] | Pn[K] | lp[K]
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  double* buf = (double*)dyn;
  double* bc = buf + PS_T;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int K = d.K, G = d.G, N = d.N;
  double Esum = 0.0;
  if (!from_prior) {
    double acc = 0.0;
    for (int g = tid; g < G; g += PS_T) acc = acc + d.E[n + (size_t)N * g];
    const double r = block_tree<PS_T>(acc, buf, tid);
    if (tid == 0) { bc[0] = r; d.Esum[n] = r; }
    __syncthreads();
    Esum = bc[0];
  }
  const double a_n = d.A[n];
  // column n of P lives in LDS while Psum / log-prior are reduced canonically over k
  double* Pn = bc + 2;                // [K]
  double* lp = Pn + K;                // [K]
  for (int k = tid; k < K; k += PS_T) {
    const int e = k + K * n;
    if (do_hyper) hyper_elem<0>(d, e, t, d.P[e]);
    double x;
    if (from_prior || a_n == 0.0) x = prior_draw<0>(d, e, t);
    else {
      double shape, rate;
      if (d.prior == BNMF_GAMMA) { shape = d.Alpha_p[e] + (double)d.ZsumG[e]; rate = d.Beta_p[e] + a_n * Esum; }
      else { shape = 1.0 + (double)d.ZsumG[e]; rate = d.Lam_p[e] + a_n * Esum; }
      Stream s(d.k0, d.k1, BNMF_V_P, (uint32_t)e, t);
      x = rgamma(s, shape, rate);
    }
    d.P[e] = x;
    d.ZsumG[e] = 0;                    // consumed; k_zalloc accumulates the next one
    Pn[k] = x;
    lp[k] = prior_logdens<0>(d, e, x);
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  if (wave < 2) {
    const double* src = wave ? lp : Pn;
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) acc = acc + src[k];
    acc = wave_tree64(acc);
    if (lane == 0) (wave ? d.lpPn : d.Psum)[n] = acc;
  }
}

// ---- k_eside: one lane per element (n,g) of E, flat column-major index e = n + N g ----
// sample_En_poisson R/sample_En.R:97-119 with the E-side hyper sweep fused in.
constexpr int ES_T = 256;
__global__ __launch_bounds__(ES_T) void k_eside(Dev d, uint32_t t, int from_prior, int do_hyper) {
  __shared__ double buf[ES_T];
  const int tid = threadIdx.x;
  const long NE = (long)d.N * d.G;
  const long e = (long)blockIdx.x * ES_T + tid;
  double lp = 0.0;
  if (e < NE) {
    const int n = (int)(e % d.N);
    const double a_n = d.A[n];
    if (do_hyper) hyper_elem<1>(d, (int)e, t, d.E[e]);
    double x;
    if (from_prior || a_n == 0.0) x = prior_draw<1>(d, (int)e, t);
    else {
      double shape, rate;
      if (d.prior == BNMF_GAMMA) { shape = d.Alpha_e[e] + (double)d.ZsumK[e]; rate = d.Beta_e[e] + a_n * d.Psum[n]; }
      else { shape = 1.0 + (double)d.ZsumK[e]; rate = d.Lam_e[e] + a_n * d.Psum[n]; }
      Stream s(d.k0, d.k1, BNMF_V_E, (uint32_t)e, t);
      x = rgamma(s, shape, rate);
    }
    d.E[e] = x;
    lp = prior_logdens<1>(d, (int)e, x);
  }
  const double r = block_tree<ES_T>(lp, buf, tid);
  if (tid == 0) d.lpE_part[blockIdx.x] = r;
}

// ---- k_zalloc: the hot kernel ----
// sample_Zkg R/sample_params.R:253-265 for every cell, fused with Mhat (R/utils.R:29-49) and the
// per-cell RMSE / KL / Poisson log-lik terms (R/utils.R:62-112, :412-471).
// A workgroup takes passes of CB columns x K rows (<= 256 cells).  Phase 1: one lane per cell
// builds the cell's cumulative thresholds thr[cell][n] = floor(cum_n * 2^32 / Mhat) in LDS.
// Phase 2: the pass's counts are flattened into "quads" (4 counts = one Philox block of the
// cell's stream) and dealt round-robin to the 256 lanes, so lanes stay balanced whatever the
// count distribution; each count binary-searches its cell's thresholds and bumps zacc[cell][n]
// with an LDS atomic.  Phase 3: ZsumK of the pass's columns = column sums of zacc minus their
// previous value; ZsumG is flushed once per workgroup with global integer atomics (exact,
// order-independent).
constexpr int ZT = 256;
template <bool SAVE_Z>
__global__ __launch_bounds__(ZT) void k_zalloc(Dev d, uint32_t t, int NP, int CB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  const int ncell = K * CB;
  uint32_t* thr = (uint32_t*)smem;                       // [ncell][NP]
  uint32_t* zacc = thr + (size_t)ncell * NP;             // [ncell][NP]
  uint32_t* zprev = zacc + (size_t)ncell * NP;           // [ncell][NP] (SAVE_Z only)
  double* red = (double*)(zprev + (SAVE_Z ? (size_t)ncell * NP : 0));   // [3][ZT]
  int* qoff = (int*)(red + 3 * ZT);                      // [ZT+1]
  int* mcnt = qoff + ZT + 4;                             // [ZT]
  int* nlastA = mcnt + ZT;                               // [ZT]
  int* totold = nlastA + ZT;                             // [CB*N]
  int* wsum = totold + CB * N;                           // [4]
  for (int i = tid; i < ncell * NP; i += ZT) { zacc[i] = 0; if (SAVE_Z) zprev[i] = 0; }
  for (int i = tid; i < CB * N; i += ZT) totold[i] = 0;
  __syncthreads();
  const int npass = (G + CB - 1) / CB;
  for (int pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const int g0 = pass * CB;
    // ---------------- phase 1: thresholds + per-cell metric terms
    int q = 0;
    {
      double sse = 0.0, ll = 0.0, kl = 0.0;
      int m = 0, nl = -1;
      if (tid < ncell) {
        const int cb = tid / K, kk = tid - cb * K, g = g0 + cb;
        if (g < G) {
          const double* Eg = d.E + (size_t)N * g;
          double c = 0.0;
          for (int n = 0; n < N; ++n) {
            const double p = (d.P[kk + (size_t)K * n] * d.A[n]) * Eg[n];
            c = c + p;
            if (p > 0.0) nl = n;
          }
          m = d.M[kk + (size_t)K * g];
          if (c > 0.0 && m > 0 && nl >= 0) {
            const double scale = 4294967296.0 / c;
            double cc = 0.0;
            uint32_t* row = thr + (size_t)tid * NP;
            for (int n = 0; n < N - 1; ++n) {
              const double p = (d.P[kk + (size_t)K * n] * d.A[n]) * Eg[n];
              cc = cc + p;
              const double tt = cc * scale;
              row[n] = (tt >= 4294967295.0) ? 0xFFFFFFFFu : (uint32_t)tt;
            }
            q = (m + 3) >> 2;
          }
          const double dd = c - (double)m;
          sse = dd * dd;
          const double mh = c < 1e-6 ? 1e-6 : c;
          const double lmh = dlog(mh);
          const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
          const double mt = m < 1 ? 1e-6 : (double)m;
          ll = ((double)m * lmh - mh) - d.lgfact[mi];
          kl = mt * (d.logm[mi] - lmh);
        }
      }
      red[tid] = sse; red[ZT + tid] = ll; red[2 * ZT + tid] = kl;
      mcnt[tid] = q > 0 ? m : 0;
      nlastA[tid] = nl;
    }
    // exclusive scan of q over the 256 lanes -> qoff
    int incl = q;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    qoff[tid] = base + incl - q;
    if (tid == ZT - 1) qoff[ZT] = base + incl;
    __syncthreads();
    const int Q = qoff[ZT];
    // ---------------- phase 2: allocate counts
    for (int qi = tid; qi < Q; qi += ZT) {
      int lo = 0, len = ZT;                       // upper_bound(qoff[0..ZT), qi) - 1
      while (len > 0) { const int half = len >> 1; if (qoff[lo + half] <= qi) { lo += half + 1; len -= half + 1; } else len = half; }
      const int cell = lo - 1;
      const int cb = cell / K, kk = cell - cb * K, g = g0 + cb;
      const int j0 = (qi - qoff[cell]) << 2;
      const int nd = min(4, mcnt[cell] - j0);
      const u32x4 w = philox4x32_10((uint32_t)(j0 >> 2), (uint32_t)(kk + (size_t)K * g), t, BNMF_V_Z, d.k0, d.k1);
      const uint32_t* row = thr + (size_t)cell * NP;
      uint32_t* zrow = zacc + (size_t)cell * NP;
      const int nl = nlastA[cell];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < nd) {
          const uint32_t u = j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w;
          int b = 0, bl = N - 1;                  // upper_bound(row[0..N-1), u)
          while (bl > 0) { const int half = bl >> 1; if (row[b + half] <= u) { b += half + 1; bl -= half + 1; } else bl = half; }
          if (b > nl) b = nl;
          atomicAdd(&zrow[b], 1u);
        }
      }
    }
    __syncthreads();
    // ---------------- phase 3: ZsumK of this pass's columns (+ Z), canonical metric sums
    for (int i = tid; i < CB * N; i += ZT) {
      const int cb = i / N, n = i - cb * N, g = g0 + cb;
      if (g < G) {
        int tot = 0;
        for (int kk = 0; kk < K; ++kk) tot += (int)zacc[(size_t)(cb * K + kk) * NP + n];
        d.ZsumK[n + (size_t)N * g] = tot - totold[i];
        totold[i] = tot;
      }
    }
    if (SAVE_Z) {
      for (int i = tid; i < ncell * N; i += ZT) {       // i = kk + K*(n + N*cb): coalesced Z store
        const int kk = i % K, r = i / K, n = r % N, cb = r / N, g = g0 + cb;
        if (g < G) {
          const size_t a = (size_t)(cb * K + kk) * NP + n;
          const uint32_t z = zacc[a];
          d.Z[kk + (size_t)K * (n + (size_t)N * g)] = (int32_t)(z - zprev[a]);
          zprev[a] = z;
        }
      }
    }
    for (int job = wave; job < CB * 3; job += ZT / 64) {
      const int cb = job / 3, mtr = job - cb * 3, g = g0 + cb;
      if (g < G) {
        const double* src = red + mtr * ZT + cb * K;
        double acc = 0.0;
        for (int kk = lane; kk < K; kk += 64) acc = acc + src[kk];
        acc = wave_tree64(acc);
        if (lane == 0) (mtr == 0 ? d.colsse : mtr == 1 ? d.colll : d.colkl)[g] = acc;
      }
    }
    __syncthreads();
  }
  // flush ZsumG: integer atomics are exact, so the result is order-independent
  for (int i = tid; i < K * N; i += ZT) {
    const int kk = i % K, n = i / K;
    uint32_t v = 0;
    for (int cb = 0; cb < CB; ++cb) v += zacc[(size_t)(cb * K + kk) * NP + n];
    if (v) atomicAdd(&d.ZsumG[kk + (size_t)K * n], (int32_t)v);
  }
}

// ---- k_finalize: canonical reductions over columns -> one metrics row ----
// compute_metrics_ R/utils.R:412-455, update_sample_metrics_ :339-348
constexpr int FN_T = 1024;
__global__ __launch_bounds__(FN_T) void k_finalize(Dev d, uint32_t t, int row, int nblkE) {
  __shared__ double buf[FN_T];
  __shared__ double res[4];
  const int tid = threadIdx.x;
  const double* srcs[4] = {d.colsse, d.colll, d.colkl, d.lpE_part};
  const int lens[4] = {d.G, d.G, d.G, nblkE};
  for (int j = 0; j < 4; ++j) {
    double acc = 0.0;
    for (int i = tid; i < lens[j]; i += FN_T) acc = acc + srcs[j][i];
    const double r = block_tree<FN_T>(acc, buf, tid);
    if (tid == 0) res[j] = r;
    __syncthreads();
  }
  if (tid == 0) {
    double lpP = 0.0, sumA = 0.0;
    for (int n = 0; n < d.N; ++n) { lpP = lpP + d.lpPn[n]; sumA = sumA + d.A[n]; }
    const double sse = res[0], ll = res[1], kl = res[2], lpE = res[3];
    const double n_params = sumA * (double)(d.G + d.K);
    double T = 1.0;
    if (d.n_temperature > 0) {
      long i = (long)t - 1;
      if (i < 0) i = 0;
      if (i >= d.n_temperature) i = d.n_temperature - 1;
      T = d.temperature[i];
    }
    double* o = d.metrics + (size_t)row * BNMF_NMETRIC;
    o[0] = (double)t;
    o[1] = dsqrt(sse / ((double)d.K * (double)d.G));
    o[2] = kl;
    o[3] = ll;
    o[4] = ll + (lpP + lpE);
    o[5] = n_params;
    o[6] = -2.0 * ll + n_params * dlog((double)d.G);
    o[7] = sumA;
    o[8] = T;
    o[9] = BNMF_NAN;
    o[10] = BNMF_NAN;
  }
}

// ---- constructor draws of the prior parameters from the hyper-priors ----
// init_prior_params_ R/sample_priors.R:15-141 (all three families are rgamma(shape, rate) draws
// for the Gamma / Exponential priors).  redraw[n] != 0: column n (P side) / row n (E side) missing.
template <int SIDE>
__global__ void k_init_gamma(Dev d, double* x, HRef hs, HRef hr, uint32_t var, const int* redraw) {
  const long len = SIDE ? (long)d.N * d.G : (long)d.K * d.N;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  const int n = SIDE ? (int)(e % d.N) : (int)(e / d.K);
  if (!redraw[n]) return;
  Stream s(d.k0, d.k1, var, (uint32_t)e, 0u);
  x[e] = rgamma(s, hy(hs, (int)e), hy(hr, (int)e));
}

// LUTs lgamma(m+1), log(max(m,1e-6)) for m = 0..maxM
__global__ void k_luts(double* lgfact, double* logm, int maxM) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m > maxM) return;
  lgfact[m] = dlgamma((double)m + 1.0);
  logm[m] = dlog(m < 1 ? 1e-6 : (double)m);
}

// ---- probes for the parity tests ----
__global__ void k_test_math(int fn, const double* in, double* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  double r;
  switch (fn) {
    case 0: r = dlog(x); break;
    case 1: r = dexp(x); break;
    case 2: r = dlgamma(x); break;
    case 3: r = ddigamma(x); break;
    case 4: r = dqnorm(x); break;
    case 5: r = dlog_pnorm(x); break;
    case 6: r = dsqrt(x); break;
    case 7: r = 1.0 / x; break;
    default: r = BNMF_NAN;
  }
  out[i] = r;
}
__global__ void k_test_sampler(int which, uint32_t k0, uint32_t k1, uint32_t var, uint32_t elem0, uint32_t iter,
                               const double* a, const double* b, const double* c, double* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Stream s(k0, k1, var, elem0 + (uint32_t)i, iter);
  double r;
  switch (which) {
    case 0: r = rgamma(s, a[i], b[i]); break;
    case 1: r = rtnorm0(s, a[i], b[i]); break;
    case 2: r = rnorm_std(s); break;
    case 3: r = ralpha(s, a[i], b[i], c[i]); break;
    case 4: r = runif(s); break;
    case 5: r = rexp(s, a[i]); break;
    default: r = BNMF_NAN;
  }
  out[i] = r;
}
__global__ void k_test_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  const u32x4 w = philox4x32_10(c0, c1, c2, c3, k0, k1);
  out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = w.w;
}

}  // namespace bnmf
