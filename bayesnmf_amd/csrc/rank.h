// bayesnmf_amd/csrc/rank.h — rank learning: sample_R and sample_An (R/sample_params.R:101-241).
//
// A[n] is updated for n = 1..N in order; each update needs two full Poisson log-likelihoods,
// one with A[n] forced to 0 and one with A[n] forced to 1, each from a fresh
// Mhat = P diag(A^j) E (get_loglik_/get_Mhat_, R/utils.R:29-112) — so one k_rank_ll pass over
// all cells and one k_rank_decide per factor.  Sums are canonical (64-strided over k inside a
// wave, then W = 1024 over g), hence the Bernoulli decisions are bit-identical to the oracle.
#pragma once

namespace bnmf {

BNMF_DEV double prior_prob_1(double R, double N) {   // compute_prior_prob_1 :178-187
  double p = R / N;
  if (p < 0.4 / N) p = 0.4 / N;
  if (p > 1.0 - 0.4 / N) p = 1.0 - 0.4 / N;
  return p;
}
BNMF_DEV double temp_at(const Dev& d, uint32_t t) {
  if (d.n_temperature <= 0) return 1.0;
  long i = (long)t - 1;
  if (i < 0) i = 0;
  if (i >= d.n_temperature) i = d.n_temperature - 1;
  return d.temperature[i];
}

// sample_R :217-241 (one lane; N+1 weights)
__global__ void k_rank_R(Dev d, uint32_t t, int from_prior) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int N = d.N;
  Stream s(d.k0, d.k1, BNMF_V_R, 0u, t);
  const double u = runif(s);
  if (from_prior) { int r = (int)(u * (double)(N + 1)); if (r > N) r = N; *d.R = r; return; }
  const double T = temp_at(d, t);
  double sumA = 0.0;
  for (int n = 0; n < N; ++n) sumA = sumA + d.A[n];
  double tot = 0.0;
  for (int r = 0; r <= N; ++r) {
    const double p1 = prior_prob_1((double)r, (double)N);
    tot = tot + dexp(T * (sumA * dlog(p1) + ((double)N - sumA) * dlog(1.0 - p1)));
  }
  const double target = u * tot;
  double cum = 0.0;
  int pick = N;
  for (int r = 0; r <= N; ++r) {
    const double p1 = prior_prob_1((double)r, (double)N);
    cum = cum + dexp(T * (sumA * dlog(p1) + ((double)N - sumA) * dlog(1.0 - p1)));
    if (target < cum) { pick = r; break; }
  }
  *d.R = pick;
}
// sample_An(from_prior = TRUE) :102-106
__global__ void k_rank_Aprior(Dev d, uint32_t t) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= d.N) return;
  const double pi1 = prior_prob_1((double)*d.R, (double)d.N);
  Stream s(d.k0, d.k1, BNMF_V_A, (uint32_t)n, t);
  d.A[n] = (runif(s) < pi1) ? 1.0 : 0.0;
}

// log-likelihood columns with A[n] forced to 0 / 1: one wave per column, lane = row
constexpr int RK_T = 256;
__global__ __launch_bounds__(RK_T) void k_rank_ll(Dev d, int n, double* col0, double* col1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  double* e0 = (double*)smem + (size_t)wave * 2 * N;   // [N] A0[j]... stored as the two scaled columns
  double* a01 = e0 + N;                                  // [N] A values with entry n forced (0 -> e0 uses a0)
  const int gw = blockIdx.x * (RK_T / 64) + wave, nw = gridDim.x * (RK_T / 64);
  const int KR = (K + 63) >> 6;
  for (int g = gw; g < G; g += nw) {
    for (int j = lane; j < N; j += 64) { e0[j] = d.E[j + (size_t)N * g]; a01[j] = d.A[j]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < KR; ++r) {
      const int kk = (r << 6) + lane;
      if (kk < K) {
        double c0 = 0.0, c1 = 0.0;
        for (int j = 0; j < N; ++j) {
          const double pe = d.P[kk + (size_t)K * j];
          const double e = e0[j];
          const double a0 = (j == n) ? 0.0 : a01[j], a1 = (j == n) ? 1.0 : a01[j];
          c0 = c0 + (pe * a0) * e;
          c1 = c1 + (pe * a1) * e;
        }
        const int m = d.M[kk + (size_t)K * g];
        const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
        const double lg = d.lgfact[mi];
        const double h0 = c0 < 1e-6 ? 1e-6 : c0, h1 = c1 < 1e-6 ? 1e-6 : c1;
        if (d.likelihood == BNMF_NORMAL) {
          const double sd = dsqrt(d.sigmasq[g]);
          const double z0 = ((double)m - c0) / sd, z1 = ((double)m - c1) / sd;
          s0 = s0 + ((-0.91893853320467274178 - dlog(sd)) - 0.5 * (z0 * z0));
          s1 = s1 + ((-0.91893853320467274178 - dlog(sd)) - 0.5 * (z1 * z1));
        } else {
          s0 = s0 + (((double)m * dlog(h0) - h0) - lg);    // canonical: lane l adds rows l, l+64, ...
          s1 = s1 + (((double)m * dlog(h1) - h1) - lg);
        }
      }
    }
    s0 = wave_tree64(s0); s1 = wave_tree64(s1);
    if (lane == 0) { col0[g] = s0; col1[g] = s1; }
    __builtin_amdgcn_wave_barrier();
  }
}

// canonical sums over columns, then the tempered Bernoulli of sample_An :108-166
__global__ __launch_bounds__(RT) void k_rank_decide(Dev d, uint32_t t, int n, const double* col0, const double* col1) {
  __shared__ double buf[RT];
  __shared__ double res[2];
  const int tid = threadIdx.x;
  const double r0 = canon1024_by256(col0, d.G, 1, buf, tid);
  if (tid == 0) res[0] = r0;
  __syncthreads();
  const double r1 = canon1024_by256(col1, d.G, 1, buf, tid);
  if (tid != 0) return;
  const int N = d.N, K = d.K, G = d.G;
  const double ll0 = res[0], ll1 = r1;
  const double pi1 = prior_prob_1((double)*d.R, (double)N);
  const double T = temp_at(d, t);
  const double a_old = d.A[n];
  double sumA = 0.0;
  for (int j = 0; j < N; ++j) sumA = sumA + d.A[j];
  const double sumA0 = sumA - a_old, sumA1 = sumA0 + 1.0;
  double s0 = ll0, s1 = ll1;
  if (d.rank_method == BNMF_SBFI) {
    const double lg = dlog((double)G);
    s0 = ll0 - (sumA0 * (double)(G + K)) * lg / 2.0;
    s1 = ll1 - (sumA1 * (double)(G + K)) * lg / 2.0;
  }
  const double lp0 = dlog(1.0 - pi1) + T * s0;
  const double lp1 = dlog(pi1) + T * s1;
  const double hi = lp0 > lp1 ? lp0 : lp1, lo = lp0 > lp1 ? lp1 : lp0;
  const double lse = hi + dlog(1.0 + dexp(lo - hi));            // sumLog :199-206
  double p = dexp(lp1 - lse);
  if (p != p) {                                                   // overflow clamp :136-162
    if (lp1 != lp1 && lp0 != lp0) p = 0.5; else if (lp1 != lp1) p = 0.0; else if (lp0 != lp0) p = 1.0;
    else if (lp1 > lp0) p = 1.0; else if (lp1 < lp0) p = 0.0; else p = 0.5;
  }
  Stream s(d.k0, d.k1, BNMF_V_A, (uint32_t)n, t);
  d.A[n] = (runif(s) < p) ? 1.0 : 0.0;
}

}  // namespace bnmf
