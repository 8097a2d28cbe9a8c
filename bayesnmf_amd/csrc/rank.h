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

// ---- the rank sweep: sample_An for n = 1..N in ONE persistent launch ----
// Every decision needs a sum over ALL cells, so the factors are separated by a grid-wide barrier (hand-written:
// agent-scope release / monotonic counter / acquire, bounded spin; the grid is sized to be co-resident, one
// workgroup per CU).  Stream spec: Mhat fresh at the start of the sweep, then maintained per cell; for factor n only
// the alternative state alt = Mhat -/+ P[k,n] E[n,g] is evaluated, the log-likelihood of the current state is carried.
// Per-column sums (64-strided over k + tree) go to colbuf[parity][g]; after the barrier every workgroup reduces them
// canonically (W = 1024 over g) and takes the same tempered Bernoulli decision.
constexpr int RK_T = 512;
constexpr int RK_W = RK_T / 64;
constexpr int RK_MAXC = 8;                                // columns per wave kept in registers (REG variant)
constexpr unsigned RK_SPIN_LIMIT = 1u << 24;

BNMF_DEV double rank_cell_ll(const Dev& d, int m, double c, double sg) {
  if (d.likelihood == BNMF_NORMAL) {
    const double sd = dsqrt(sg);
    const double z = ((double)m - c) / sd;
    return (-0.91893853320467274178 - dlog(sd)) - 0.5 * (z * z);
  }
  const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
  const double h = c < 1e-6 ? 1e-6 : c;
  return ((double)m * dlog(h) - h) - d.lgfact[mi];
}
// grid barrier number `phase` (1, 2, ...): every workgroup arrives once per phase.  Returns false on time-out.
BNMF_DEV bool rank_grid_sync(unsigned* counter, unsigned phase, unsigned nwg, int* err, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have left
  __syncthreads();
  __shared__ int ok_s;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = phase * nwg;
    unsigned spins = 0;
    bool ok = true;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > RK_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
    }
    if (!ok) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok_s = ok ? 1 : 0;
  }
  __syncthreads();
  return ok_s != 0;
}
// canonical W = 1024 sum of x[0..L) by a 512-lane workgroup: lane i owns accumulators i and i + 512
BNMF_DEV double canon1024_by512(const double* x, long L, double* buf, int tid) {
  double a0 = 0.0, a1 = 0.0;
  for (long i = tid; i < L; i += 1024) {
    a0 = a0 + x[i];
    if (i + 512 < L) a1 = a1 + x[i + 512];
  }
  a0 = a0 + a1;                                           // tree level h = 512
  const double r = block_tree<RK_T>(a0, buf, tid);        // valid on thread 0
  __syncthreads();
  return r;
}
// tempered Bernoulli of sample_An :108-166 from the two log-likelihoods; A still holds the old value of factor n
BNMF_DEV double rank_decide(const Dev& d, uint32_t t, int n, double ll0, double ll1, const double* A) {
  const int N = d.N, K = d.K, G = d.G;
  const double pi1 = prior_prob_1((double)*d.R, (double)N);
  const double T = temp_at(d, t);
  const double a_old = A[n];
  double sumA = 0.0;
  for (int j = 0; j < N; ++j) sumA = sumA + A[j];
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
  return (runif(s) < p) ? 1.0 : 0.0;
}

// REG: the wave's cells (<= RK_MAXC columns x 2 row passes, K <= 128) stay in registers for the whole sweep;
// otherwise Mhat lives in the global scratch mhg[k + K g].
template <bool REG>
__global__ __launch_bounds__(RK_T) void k_rank_sweep(Dev d, uint32_t t, double* colbuf /* [2][G] */, unsigned* counter, int* err, double* mhg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double buf[RK_T];
  __shared__ double bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  const int KR = (K + 63) >> 6;
  const int wg = blockIdx.x * RK_W + wave, Wt = gridDim.x * RK_W;
  const unsigned nwg = gridDim.x;
  // the workgroup's own copy of A: every workgroup takes every decision itself, so A is never read across
  // workgroups inside the launch (the global A is written for the kernels that follow)
  double* Ash = (double*)smem;                           // [N]
  for (int j = tid; j < N; j += RK_T) Ash[j] = d.A[j];
  __syncthreads();
  double mh[REG ? RK_MAXC : 1][2];
  int mm[REG ? RK_MAXC : 1][2];
  // ---- phase 0: fresh Mhat and the log-likelihood of the current state
  double* col = colbuf;
  if (REG) {
#pragma unroll
    for (int c = 0; c < RK_MAXC; ++c) {
      const int g = wg + c * Wt;
      if (g < G) {
        const double sg = d.likelihood == BNMF_NORMAL ? d.sigmasq[g] : 1.0;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int kk = (r << 6) + lane;
          mh[c][r] = 0.0; mm[c][r] = 0;
          if (kk < K) {
            double cc = 0.0;
            for (int j = 0; j < N; ++j) cc = cc + (d.P[kk + (size_t)K * j] * Ash[j]) * d.E[j + (size_t)N * g];
            const int m = d.M[kk + (size_t)K * g];
            mh[c][r] = cc; mm[c][r] = m;
            acc = acc + rank_cell_ll(d, m, cc, sg);
          }
        }
        acc = wave_tree64(acc);
        if (lane == 0) col[g] = acc;
      }
    }
  } else {
    for (int g = wg; g < G; g += Wt) {
      const double sg = d.likelihood == BNMF_NORMAL ? d.sigmasq[g] : 1.0;
      double acc = 0.0;
      for (int r = 0; r < KR; ++r) {
        const int kk = (r << 6) + lane;
        if (kk < K) {
          double cc = 0.0;
          for (int j = 0; j < N; ++j) cc = cc + (d.P[kk + (size_t)K * j] * Ash[j]) * d.E[j + (size_t)N * g];
          mhg[kk + (size_t)K * g] = cc;
          acc = acc + rank_cell_ll(d, d.M[kk + (size_t)K * g], cc, sg);
        }
      }
      acc = wave_tree64(acc);
      if (lane == 0) col[g] = acc;
    }
  }
  unsigned phase = 1;
  if (!rank_grid_sync(counter, phase, nwg, err, tid)) return;
  double ll_cur = canon1024_by512(col, G, buf, tid);
  if (tid == 0) bc[0] = ll_cur;
  __syncthreads();
  ll_cur = bc[0];
  // ---- factors in order
  for (int n = 0; n < N; ++n) {
    const double a_old = Ash[n];
    col = colbuf + (size_t)((n + 1) & 1) * G;
    const double* Pn = d.P + (size_t)K * n;
    if (REG) {
      const double p0 = lane < K ? Pn[lane] : 0.0, p1 = 64 + lane < K ? Pn[64 + lane] : 0.0;
#pragma unroll
      for (int c = 0; c < RK_MAXC; ++c) {
        const int g = wg + c * Wt;
        if (g < G) {
          const double sg = d.likelihood == BNMF_NORMAL ? d.sigmasq[g] : 1.0;
          const double en = d.E[n + (size_t)N * g];
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int kk = (r << 6) + lane;
            if (kk < K) {
              const double tt = (r ? p1 : p0) * en;
              const double alt = (a_old == 1.0) ? mh[c][r] - tt : mh[c][r] + tt;
              acc = acc + rank_cell_ll(d, mm[c][r], alt, sg);
            }
          }
          acc = wave_tree64(acc);
          if (lane == 0) col[g] = acc;
        }
      }
    } else {
      for (int g = wg; g < G; g += Wt) {
        const double sg = d.likelihood == BNMF_NORMAL ? d.sigmasq[g] : 1.0;
        const double en = d.E[n + (size_t)N * g];
        double acc = 0.0;
        for (int r = 0; r < KR; ++r) {
          const int kk = (r << 6) + lane;
          if (kk < K) {
            const double tt = Pn[kk] * en;
            const double cur = mhg[kk + (size_t)K * g];
            const double alt = (a_old == 1.0) ? cur - tt : cur + tt;
            acc = acc + rank_cell_ll(d, d.M[kk + (size_t)K * g], alt, sg);
          }
        }
        acc = wave_tree64(acc);
        if (lane == 0) col[g] = acc;
      }
    }
    ++phase;
    if (!rank_grid_sync(counter, phase, nwg, err, tid)) return;
    const double ll_alt = canon1024_by512(col, G, buf, tid);
    if (tid == 0) {
      const double ll0 = (a_old == 1.0) ? ll_alt : ll_cur, ll1 = (a_old == 1.0) ? ll_cur : ll_alt;
      bc[0] = rank_decide(d, t, n, ll0, ll1, Ash);
      bc[1] = ll_alt;
    }
    __syncthreads();
    const double a_new = bc[0];
    if (a_new != a_old) {
      ll_cur = bc[1];
      if (REG) {
        const double p0 = lane < K ? Pn[lane] : 0.0, p1 = 64 + lane < K ? Pn[64 + lane] : 0.0;
#pragma unroll
        for (int c = 0; c < RK_MAXC; ++c) {
          const int g = wg + c * Wt;
          if (g < G) {
            const double en = d.E[n + (size_t)N * g];
#pragma unroll
            for (int r = 0; r < 2; ++r) { const double tt = (r ? p1 : p0) * en; mh[c][r] = (a_old == 1.0) ? mh[c][r] - tt : mh[c][r] + tt; }
          }
        }
      } else {
        for (int g = wg; g < G; g += Wt) {
          const double en = d.E[n + (size_t)N * g];
          for (int r = 0; r < KR; ++r) {
            const int kk = (r << 6) + lane;
            if (kk < K) { const double tt = Pn[kk] * en; const double cur = mhg[kk + (size_t)K * g]; mhg[kk + (size_t)K * g] = (a_old == 1.0) ? cur - tt : cur + tt; }
          }
        }
      }
    }
    if (tid == 0) { Ash[n] = a_new; if (blockIdx.x == 0) d.A[n] = a_new; }
    __syncthreads();
  }
}

}  // namespace bnmf
