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

// sample_R :217-241: the N+1 weights are evaluated one per lane, then added and scanned in r order by lane 0
__global__ __launch_bounds__(64) void k_rank_R(Dev d, uint32_t t, int from_prior) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* w = (double*)smem;                             // [N+1]
  if (blockIdx.x != 0) return;
  const int N = d.N, lane = threadIdx.x;
  Stream s(d.k0, d.k1, BNMF_V_R, 0u, t);
  const double u = runif(s);
  if (from_prior) { if (lane == 0) { int r = (int)(u * (double)(N + 1)); if (r > N) r = N; *d.R = r; } return; }
  const double T = temp_at(d, t);
  double sumA = 0.0;
  for (int n = 0; n < N; ++n) sumA = sumA + d.A[n];
  for (int r = lane; r <= N; r += 64) {
    const double p1 = prior_prob_1((double)r, (double)N);
    w[r] = dexp(T * (sumA * dlog(p1) + ((double)N - sumA) * dlog(1.0 - p1)));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane != 0) return;
  double tot = 0.0;
  for (int r = 0; r <= N; ++r) tot = tot + w[r];
  const double target = u * tot;
  double cum = 0.0;
  int pick = N;
  for (int r = 0; r <= N; ++r) { cum = cum + w[r]; if (target < cum) { pick = r; break; } }
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
// tagged-granule all-gather, bounded spin; the grid is sized to be co-resident: 512-lane workgroups, one per CU,
// which also keeps the waves' compute time even).  Stream spec: Mhat fresh at the start of the sweep, then maintained per cell; for factor n only
// the alternative state alt = Mhat -/+ P[k,n] E[n,g] is evaluated, the log-likelihood of the current state is carried.
// Per-column sums (64-strided over k + tree) are added in blocks of 8 consecutive columns (one wave owns whole
// blocks) into blkbuf[parity][b]; after the barrier every workgroup reduces the block sums canonically (W = 1024)
// and takes the same tempered Bernoulli decision.
constexpr int RK_T = 512;
constexpr int RK_W = RK_T / 64;
constexpr int RK_MAXC = 8;                                // columns per block = columns per wave kept in registers (REG variant)
constexpr unsigned RK_SPIN_LIMIT = 1u << 22;

BNMF_DEV double rank_cell_ll(const Dev& d, int m, double c, double sg, double lgf) {
  if (d.likelihood == BNMF_NORMAL) {
    const double sd = dsqrt(sg);
    const double z = ((double)m - c) / sd;
    return (-0.91893853320467274178 - dlog(sd)) - 0.5 * (z * z);
  }
  const double h = c < 1e-6 ? 1e-6 : c;
  return ((double)m * dlog(h) - h) - lgf;
}
template <bool NORMAL>
BNMF_DEV double rank_cell_ll_t(int m, double c, double sg, double lgf) {   // rank_cell_ll without the run-time model branch
  if (NORMAL) {
    const double sd = dsqrt(sg);
    const double z = ((double)m - c) / sd;
    return (-0.91893853320467274178 - dlog(sd)) - 0.5 * (z * z);
  }
  const double h = c < 1e-6 ? 1e-6 : c;
  return ((double)m * dlog(h) - h) - lgf;
}
BNMF_DEV double rank_lgf(const Dev& d, int m) { return d.lgfact[m < 0 ? 0 : (m > d.maxM ? d.maxM : m)]; }
// All-gather of the block sums without a separate barrier: a value is published as two 8-byte granules
// {tag, low word}, {tag, high word} (one agent-scope relaxed store each: write-through, never torn), tag = a number
// unique to (iteration, phase).  Every workgroup sweeps all granules with agent-scope relaxed loads (which bypass
// its L1) until every tag matches, and rebuilds the doubles in LDS: the data is its own flag, no fences needed.
// Bounded: a time-out sets *err and makes every workgroup leave.
BNMF_DEV void rank_publish(unsigned long long* gran, int b, unsigned tag, double v) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(gran + 2 * (size_t)b, ((unsigned long long)tag << 32) | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(gran + 2 * (size_t)b + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A lane's granules are requested GB blocks at a time (one round trip to the memory side for all of them instead of one per
// block).  rank_request issues the loads of the lane's first GB blocks; rank_gather takes them (pre != nullptr) or loads them
// itself, polls every block whose tags do not match yet, and handles the blocks beyond the first GB in the same way.
constexpr int RK_GB = 3;
struct RankPre { unsigned long long g0[RK_GB], g1[RK_GB]; };
BNMF_DEV void rank_request(const unsigned long long* gran, int NB, int tid, RankPre& pre) {
#pragma unroll
  for (int i = 0; i < RK_GB; ++i) {
    const int b = tid + i * RK_T;
    pre.g0[i] = pre.g1[i] = 0ull;                         // tag 0 is never used: an unrequested block fails the match and is polled
    if (b < NB) {
      pre.g0[i] = __hip_atomic_load(gran + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pre.g1[i] = __hip_atomic_load(gran + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
BNMF_DEV bool rank_gather(const unsigned long long* gran, int NB, unsigned tag, double* vals /* LDS [NB] */, int* err, int tid, const RankPre* pre = nullptr) {
  __shared__ int bad_s;
  if (tid == 0) bad_s = 0;
  __syncthreads();
  bool bad = false;
  for (int b0 = tid; b0 < NB && !bad; b0 += RK_GB * RK_T) {
    RankPre cur;
    if (pre && b0 == tid) cur = *pre; else rank_request(gran + 2 * (size_t)(b0 - tid), NB - (b0 - tid), tid, cur);
#pragma unroll
    for (int i = 0; i < RK_GB; ++i) {
      const int b = b0 + i * RK_T;
      if (b >= NB || bad) continue;
      unsigned long long h0 = cur.g0[i], h1 = cur.g1[i];
      unsigned spins = 0;
      while ((unsigned)(h0 >> 32) != tag || (unsigned)(h1 >> 32) != tag) {
        if (spins) __builtin_amdgcn_s_sleep(1);
        if (++spins > RK_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { bad = true; break; }
        h0 = __hip_atomic_load(gran + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h1 = __hip_atomic_load(gran + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!bad) vals[b] = __longlong_as_double((long long)((h0 & 0xFFFFFFFFull) | (h1 << 32)));
    }
  }
  if (bad) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad_s = 1; }
  __syncthreads();
  return bad_s == 0;
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
// tempered Bernoulli of sample_An :108-166 from the two log-likelihoods (the sweep's constants are hoisted by the caller)
struct RankConst { double l1mp, lpi, lgG, T; };
BNMF_DEV double rank_decide(const Dev& d, double u /* the factor's uniform: block 0 of stream (BNMF_V_A, n, t) */, double ll0, double ll1, double a_old, double sumA, const RankConst& rc) {
  const int K = d.K, G = d.G;
  const double sumA0 = sumA - a_old, sumA1 = sumA0 + 1.0;
  double s0 = ll0, s1 = ll1;
  if (d.rank_method == BNMF_SBFI) {
    s0 = ll0 - (sumA0 * (double)(G + K)) * rc.lgG / 2.0;
    s1 = ll1 - (sumA1 * (double)(G + K)) * rc.lgG / 2.0;
  }
  const double lp0 = rc.l1mp + rc.T * s0;
  const double lp1 = rc.lpi + rc.T * s1;
  const double hi = lp0 > lp1 ? lp0 : lp1, lo = lp0 > lp1 ? lp1 : lp0;
  const double lse = hi + dlog(1.0 + dexp(lo - hi));            // sumLog :199-206
  double p = dexp(lp1 - lse);
  if (p != p) {                                                   // overflow clamp :136-162
    if (lp1 != lp1 && lp0 != lp0) p = 0.5; else if (lp1 != lp1) p = 0.0; else if (lp0 != lp0) p = 1.0;
    else if (lp1 > lp0) p = 1.0; else if (lp1 < lp0) p = 0.0; else p = 0.5;
  }
  return (u < p) ? 1.0 : 0.0;
}

// REG (K <= 96): one block of 8 columns per wave, its cells in registers for the whole sweep: rows 0..63 of every
// column (slot 0: lane = row), and rows 64..95 of TWO columns per register (slot 1: lanes 0..31 column 2p, lanes 32..63
// column 2p + 1, row 64 + (lane & 31)), so that no lane idles on a half-empty second pass; the upper half's terms are
// brought down with v_permlane32_swap before they are added to their column's accumulators (lane i = rows i, i + 64, as
// the canonical sum demands).  Otherwise a wave walks its blocks b = w, w + Wt, ... and Mhat lives in the global scratch.
template <bool REG, bool NORMAL>
__global__ __launch_bounds__(RK_T, 2) void k_rank_sweep(Dev d, uint32_t t, unsigned long long* granbuf /* [4][2 NB] */, int NB, int* err, double* mhg, unsigned long long* dbg,
                                                         int row /* metrics row, or -1 */, double* recA, double* recR) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double buf[RK_T];
  __shared__ double bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  const int KR = (K + 63) >> 6;
  // block (REG) / first block (otherwise) of this wave: wave-major, so that a grid wider than NB / RK_W spreads the
  // blocks over all CUs (about one busy wave per SIMD instead of two on 60 % of the CUs)
  const int wg = wave * gridDim.x + blockIdx.x, Wt = gridDim.x * RK_W;
  constexpr bool normal = NORMAL;
  // tags of this launch, unique over the chain: tag0 + 1 = log-likelihood of the current state, tag0 + 2 + 2n + redo = the
  // alternative of factor n (redo = 1: evaluated again after factor n-1 flipped, see the factor loop)
  const unsigned tag0 = t * (unsigned)(2 * N + 4);
  // the workgroup's own copy of A: every workgroup takes every decision itself, so A is never read across
  // workgroups inside the launch (the global A is written for the kernels that follow)
  double* Ash = (double*)smem;                           // [N]
  double* vals = Ash + N;                                // [NB] gathered block sums
  double* wR = vals + NB;                                // [N+1] weights of sample_R
  double* uA = wR + (N + 1);                             // [N] the factors' uniforms (sample_An's rbinom), drawn up front by N lanes
  __shared__ int Rsh;
  for (int j = tid; j < N; j += RK_T) { Ash[j] = d.A[j]; Stream sa(d.k0, d.k1, BNMF_V_A, (uint32_t)j, t); uA[j] = runif(sa); }
  __syncthreads();
  // sample_R :217-241 (was a launch of its own): every workgroup draws the same R from the same stream; the N+1 weights
  // are evaluated one per lane of wave 0, then added and scanned in r order by its lane 0
  if (wave == 0) {
    Stream s(d.k0, d.k1, BNMF_V_R, 0u, t);
    const double u = runif(s);
    const double T = temp_at(d, t);
    double sA = 0.0;
    for (int n = 0; n < N; ++n) sA = sA + Ash[n];
    for (int r = lane; r <= N; r += 64) {
      const double p1 = prior_prob_1((double)r, (double)N);
      wR[r] = dexp(T * (sA * dlog(p1) + ((double)N - sA) * dlog(1.0 - p1)));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      double tot = 0.0;
      for (int r = 0; r <= N; ++r) tot = tot + wR[r];
      const double target = u * tot;
      double cum = 0.0;
      int pick = N;
      for (int r = 0; r <= N; ++r) { cum = cum + wR[r]; if (target < cum) { pick = r; break; } }
      Rsh = pick;
      if (blockIdx.x == 0) *d.R = pick;
    }
  }
  constexpr int RK_P = RK_MAXC / 2;                       // column pairs of the second row slot
  double mh0[REG ? RK_MAXC : 1], mh1[REG ? RK_P : 1], sgc[(REG && NORMAL) ? RK_MAXC : 1];
  double lg0[(REG && !NORMAL) ? RK_MAXC : 1], lg1[(REG && !NORMAL) ? RK_P : 1];   // lgamma(M + 1) of the wave's cells
  int mm0[REG ? RK_MAXC : 1], mm1[REG ? RK_P : 1];
  const int half = lane >> 5, row1 = 64 + (lane & 31);    // slot 1: this lane's column of the pair and its row
  const bool lowv1 = lane < 32 && 64 + lane < K;          // lanes that own an accumulator with a second row
  // ---- phase 0: fresh Mhat and the log-likelihood of the current state
  unsigned long long* gran = granbuf;                    // buffer 0 of 4
  const unsigned phase = 1;
  if (REG) {
    double accv[RK_MAXC];
    auto fresh = [&](int kk, int g, double sg, double& mhv, int& mv, double& lgv) {   // Mhat, M, lgamma(M+1) and the cell's term
      double cc = 0.0;
      for (int j = 0; j < N; ++j) cc = cc + (d.P[kk + (size_t)K * j] * Ash[j]) * d.E[j + (size_t)N * g];
      const int m = d.M[kk + (size_t)K * g];
      const double lgf = rank_lgf(d, m);
      mhv = cc; mv = m; lgv = lgf;
      return rank_cell_ll(d, m, cc, sg, lgf);
    };
#pragma unroll
    for (int c = 0; c < RK_MAXC; ++c) {                   // slot 0: rows 0..63 (cells beyond G / K: harmless values, never added)
      const int g = wg * RK_MAXC + c;
      if (NORMAL) sgc[c] = 1.0;
      mh0[c] = 0.0; mm0[c] = 0; accv[c] = 0.0;
      double lgv = 0.0;
      if (g < G) {
        const double sg = normal ? d.sigmasq[g] : 1.0;
        if (NORMAL) sgc[c] = sg;
        if (lane < K) accv[c] = accv[c] + fresh(lane, g, sg, mh0[c], mm0[c], lgv);
      }
      if (!NORMAL) lg0[NORMAL ? 0 : c] = lgv;
    }
#pragma unroll
    for (int p = 0; p < RK_P; ++p) {                      // slot 1: rows 64..95 of columns 2p (lanes 0..31) and 2p + 1 (lanes 32..63)
      const int g = wg * RK_MAXC + 2 * p + half;
      mh1[p] = 0.0; mm1[p] = 0;
      double lgv = 0.0, v = 0.0;
      if (g < G && row1 < K) v = fresh(row1, g, normal ? d.sigmasq[g] : 1.0, mh1[p], mm1[p], lgv);
      if (!NORMAL) lg1[NORMAL ? 0 : p] = lgv;
      const double w = down32(v);
      if (lowv1 && wg * RK_MAXC + 2 * p < G) accv[2 * p] = accv[2 * p] + v;
      if (lowv1 && wg * RK_MAXC + 2 * p + 1 < G) accv[2 * p + 1] = accv[2 * p + 1] + w;
    }
    double bs = 0.0;
#pragma unroll
    for (int c = 0; c < RK_MAXC; ++c) { const double tr = wave_tree64(accv[c]); if (wg * RK_MAXC + c < G) bs = bs + tr; }   // lane 0: block sum, columns in ascending order
    if (lane == 0 && wg < NB) rank_publish(gran, wg, tag0 + phase, bs);
  } else {
    for (int b = wg; b < NB; b += Wt) {
      double bs = 0.0;
      for (int c = 0; c < RK_MAXC; ++c) {
        const int g = b * RK_MAXC + c;
        if (g >= G) break;
        const double sg = normal ? d.sigmasq[g] : 1.0;
        double acc = 0.0;
        for (int r = 0; r < KR; ++r) {
          const int kk = (r << 6) + lane;
          if (kk < K) {
            double cc = 0.0;
            for (int j = 0; j < N; ++j) cc = cc + (d.P[kk + (size_t)K * j] * Ash[j]) * d.E[j + (size_t)N * g];
            mhg[kk + (size_t)K * g] = cc;
            const int m = d.M[kk + (size_t)K * g];
            acc = acc + rank_cell_ll(d, m, cc, sg, rank_lgf(d, m));
          }
        }
        bs = bs + wave_tree64(acc);
      }
      if (lane == 0) rank_publish(gran, b, tag0 + phase, bs);
    }
  }
  // ---- factors in order, one step ahead of the decisions.  The alternative log-likelihood of factor n+1 depends on the
  // decision for factor n only through Mhat, and most decisions leave A[n] as it was.  So a workgroup evaluates and publishes
  // the alternative of factor n+1 BEFORE it gathers the block sums of factor n: by the time it has finished, every other
  // workgroup's sums of factor n (published one step earlier) have arrived, and the gather is a read instead of a wait for the
  // slowest publisher.  When factor n does flip, Mhat is updated and the alternative of n+1 is evaluated and published again
  // under the `redo` tag; every workgroup takes the same decision, so all of them know which tag to gather.  Same values,
  // same order of operations, same draws as the one-factor-at-a-time sweep.
  // Four granule buffers: a workgroup in step n writes buffer (n+2)&3 while the slowest one may still read (n-1)&3 .. (n+1)&3.
  auto gbuf = [&](int n) { return granbuf + (size_t)((n + 1) & 3) * 2 * NB; };
  auto tagof = [&](int n, unsigned redo) { return tag0 + 2u + 2u * (unsigned)n + redo; };
  // REG: column n of P and row n of E (this wave's 8 columns) are requested one factor ahead of their use
  double np0 = 0.0, np1 = 0.0, nen[REG ? RK_MAXC : 1];
  auto prefetch = [&](int n) {
    const double* Pq = d.P + (size_t)K * n;
    np0 = lane < K ? Pq[lane] : 0.0; np1 = row1 < K ? Pq[row1] : 0.0;   // slot 1: both half-waves hold rows 64 + (lane & 31)
#pragma unroll
    for (int c = 0; c < (REG ? RK_MAXC : 1); ++c) { const int g = wg * RK_MAXC + c; nen[c] = g < G ? d.E[n + (size_t)N * g] : 0.0; }
  };
  double p0 = 0.0, p1 = 0.0, en_[REG ? RK_MAXC : 1];      // REG: the factor whose alternative is evaluated next
  auto take_prefetched = [&](int n_next) {                 // current <- prefetched, request factor n_next
    p0 = np0; p1 = np1;
#pragma unroll
    for (int c = 0; c < (REG ? RK_MAXC : 1); ++c) en_[c] = nen[c];
    if (n_next < N) prefetch(n_next);
  };
  // evaluate the alternative of factor f (A[f] flipped) on the current Mhat and publish this wave's block sums
  auto publish_alt = [&](int f, unsigned redo) {
    const double a_f = Ash[f];
    unsigned long long* gr = gbuf(f);
    const unsigned tg = tagof(f, redo);
    if (REG) {
      // straight-line code: the 16 cell terms and then the 8 column trees are independent chains the scheduler can
      // interleave (per-column / per-cell branches kept them apart: 5.3 us of pure latency per factor); cells beyond K and
      // columns beyond G hold harmless values and are never added
      double accv[RK_MAXC];
#pragma unroll
      for (int c = 0; c < RK_MAXC; ++c) {                 // slot 0
        const double tt = p0 * en_[c];
        const double alt = (a_f == 1.0) ? mh0[c] - tt : mh0[c] + tt;
        const double ll = rank_cell_ll_t<NORMAL>(mm0[c], alt, NORMAL ? sgc[NORMAL ? c : 0] : 1.0, NORMAL ? 0.0 : lg0[NORMAL ? 0 : c]);
        accv[c] = (lane < K) ? 0.0 + ll : 0.0;
      }
#pragma unroll
      for (int p = 0; p < RK_P; ++p) {                    // slot 1: two columns per register
        const double tt = p1 * (half ? en_[2 * p + 1] : en_[2 * p]);
        const double alt = (a_f == 1.0) ? mh1[p] - tt : mh1[p] + tt;
        const double sg = NORMAL ? (half ? sgc[NORMAL ? 2 * p + 1 : 0] : sgc[NORMAL ? 2 * p : 0]) : 1.0;
        const double v = rank_cell_ll_t<NORMAL>(mm1[p], alt, sg, NORMAL ? 0.0 : lg1[NORMAL ? 0 : p]);
        const double w = down32(v);
        accv[2 * p] = lowv1 ? accv[2 * p] + v : accv[2 * p];
        accv[2 * p + 1] = lowv1 ? accv[2 * p + 1] + w : accv[2 * p + 1];
      }
#pragma unroll
      for (int c = 0; c < RK_MAXC; ++c) accv[c] = wave_tree64(accv[c]);
      double bs = 0.0;
#pragma unroll
      for (int c = 0; c < RK_MAXC; ++c) bs = (wg * RK_MAXC + c < G) ? bs + accv[c] : bs;   // lane 0: block sum, columns in ascending order
      if (lane == 0 && wg < NB) rank_publish(gr, wg, tg, bs);
    } else {
      const double* Pf = d.P + (size_t)K * f;
      for (int b = wg; b < NB; b += Wt) {
        double bs = 0.0;
        for (int c = 0; c < RK_MAXC; ++c) {
          const int g = b * RK_MAXC + c;
          if (g >= G) break;
          const double sg = normal ? d.sigmasq[g] : 1.0;
          const double en = d.E[f + (size_t)N * g];
          double acc = 0.0;
          for (int r = 0; r < KR; ++r) {
            const int kk = (r << 6) + lane;
            if (kk < K) {
              const double tt = Pf[kk] * en;
              const double cur = mhg[kk + (size_t)K * g];
              const double alt = (a_f == 1.0) ? cur - tt : cur + tt;
              const int m = d.M[kk + (size_t)K * g];
              acc = acc + rank_cell_ll(d, m, alt, sg, rank_lgf(d, m));
            }
          }
          bs = bs + wave_tree64(acc);
        }
        if (lane == 0) rank_publish(gr, b, tg, bs);
      }
    }
  };
  // Mhat <- Mhat -/+ the term of factor f (its A flipped from a_was): the same operations as the alternative just evaluated
  auto flip_mhat = [&](int f, double a_was) {
    const double* Pf = d.P + (size_t)K * f;
    if (REG) {
      const double q0 = lane < K ? Pf[lane] : 0.0, q1 = row1 < K ? Pf[row1] : 0.0;
#pragma unroll
      for (int c = 0; c < RK_MAXC; ++c) {
        const int g = wg * RK_MAXC + c;
        const double tt = q0 * (g < G ? d.E[f + (size_t)N * g] : 0.0);
        mh0[c] = (a_was == 1.0) ? mh0[c] - tt : mh0[c] + tt;
      }
#pragma unroll
      for (int p = 0; p < RK_P; ++p) {
        const int g = wg * RK_MAXC + 2 * p + half;
        const double tt = q1 * (g < G ? d.E[f + (size_t)N * g] : 0.0);
        mh1[p] = (a_was == 1.0) ? mh1[p] - tt : mh1[p] + tt;
      }
    } else {
      for (int b = wg; b < NB; b += Wt)
        for (int c = 0; c < RK_MAXC; ++c) {
          const int g = b * RK_MAXC + c;
          if (g >= G) break;
          const double en = d.E[f + (size_t)N * g];
          for (int r = 0; r < KR; ++r) {
            const int kk = (r << 6) + lane;
            if (kk < K) { const double tt = Pf[kk] * en; const double cur = mhg[kk + (size_t)K * g]; mhg[kk + (size_t)K * g] = (a_was == 1.0) ? cur - tt : cur + tt; }
          }
        }
    }
  };
  if (REG) { prefetch(0); take_prefetched(1); }
  publish_alt(0, 0u);                                      // alternative of factor 0: needs no decision
  if (!rank_gather(gran, NB, tag0 + phase, vals, err, tid)) return;
  double ll_cur = canon1024_by512(vals, NB, buf, tid);  // valid on thread 0, which carries it
  RankConst rc{};
  double sumA = 0.0;
  if (tid == 0) {
    const double pi1 = prior_prob_1((double)Rsh, (double)N);   // written by this thread (wave 0, lane 0) above
    rc.l1mp = dlog(1.0 - pi1); rc.lpi = dlog(pi1); rc.lgG = dlog((double)G); rc.T = temp_at(d, t);
    for (int j = 0; j < N; ++j) sumA = sumA + Ash[j];
  }
  unsigned redo = 0;                                       // which publication of factor n's alternative is the valid one
  for (int n = 0; n < N; ++n) {
    const double a_old = Ash[n];
#define RKSTAMP(i) if (dbg && tid == 0 && n < 16) dbg[(blockIdx.x * 16 + n) * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
    RKSTAMP(0);
    RankPre pre;                                           // factor n's sums were published a step ago: their loads fly under the
    rank_request(gbuf(n), NB, tid, pre);                   // evaluation below (after a flip they are not there yet and are polled)
    if (n + 1 < N) {                                       // one step ahead: factor n+1 on the Mhat as it is
      if (REG) take_prefetched(n + 2);
      publish_alt(n + 1, 0u);
    }
    RKSTAMP(1);
    if (!rank_gather(gbuf(n), NB, tagof(n, redo), vals, err, tid, &pre)) return;
    RKSTAMP(2);
    const double ll_alt = canon1024_by512(vals, NB, buf, tid);
    if (tid == 0) {
      const double ll0 = (a_old == 1.0) ? ll_alt : ll_cur, ll1 = (a_old == 1.0) ? ll_cur : ll_alt;
      const double a_new = rank_decide(d, uA[n], ll0, ll1, a_old, sumA, rc);
      if (a_new != a_old) { ll_cur = ll_alt; sumA = (sumA - a_old) + a_new; }
      bc[0] = a_new;
    }
    RKSTAMP(3);
    __syncthreads();
    const double a_new = bc[0];
    redo = 0;
    if (a_new != a_old) {                                  // the step ahead was taken on a stale Mhat: again
      flip_mhat(n, a_old);                                 // (general variant: a lane re-reads only the Mhat cells it wrote itself)
      if (n + 1 < N) { publish_alt(n + 1, 1u); redo = 1; }
    }
    if (tid == 0) { Ash[n] = a_new; if (blockIdx.x == 0) d.A[n] = a_new; }
    __syncthreads();
    RKSTAMP(4);
  }
  // what k_sumA did in a launch of its own (Gibbs sweep): A, R into the ring, sum(A) into the raw metrics row
  if (blockIdx.x == 0 && row >= 0) {
    if (recA) for (int j = tid; j < N; j += RK_T) recA[j] = Ash[j];
    if (tid == 0) {
      if (recR) *recR = (double)Rsh;
      double sA = 0.0;
      for (int j = 0; j < N; ++j) sA = sA + Ash[j];
      d.raw[(size_t)row * 8 + 5] = sA;
    }
  }
}

}  // namespace bnmf
