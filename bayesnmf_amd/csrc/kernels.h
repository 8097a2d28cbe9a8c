// bayesnmf_amd/csrc/kernels.h — hand-written gfx950 kernels of the Gibbs sweep.
//
// One iteration (Poisson likelihood, no MH; R/bayesNMF_sampler.R:273-285), see DESIGN.md §5 for the stream structure:
//   k_pdraw   grid N       : P[,n] -> Psum[n] (and, at init, the log-prior of column n)
//   k_edraw   grid NG/256  : E (and, at init, the log-prior partials)
//   k_draw    the two above in one launch (steady state of large fixed-rank problems)
//   k_zalloc* one wave per column (zalloc_reg.h) / one workgroup per 32-row chunk (zalloc_tile.h): thresholds ->
//                            categorical allocation of every count -> ZsumK, ZsumG (+Z), per-column RMSE/KL/log-lik terms
//   k_side / k_side_lp (side streams, beside k_zalloc): hyper sweep of the next iteration, Esum, log-priors of this one
//   k_reduce  grid 4       : canonical reductions; k_compose once per run -> metrics rows
// All fp64; all cross-lane sums use the canonical orders of dmath.h so results are bitwise
// independent of scheduling.  No MFMA: the path is sampling + reductions.
#pragma once
#include "dsamplers.h"
#include "colterms.h"
#include "../../include/bnmf.h"

namespace bnmf {

struct HRef { const double* p; int stride; };   // hyper-prior matrix or broadcast scalar
BNMF_DEV double hy(const HRef& h, int e) { return h.p[(size_t)e * h.stride]; }

struct Dev {
  int K, G, N;
  int prior, likelihood, MH, learning_rank, rank_method, save_Z;
  int zsumk_accum;      // the allocation kernel accumulates ZsumK across row chunks (k_zalloc_tile): k_edraw zeroes what it has consumed
  uint32_t k0, k1;
  int maxM;
  const int32_t* M;
  const int32_t* Mt;    // [G][K] transpose of M (MH / Normal models: lanes walk the columns of one row)
  double* Et;           // [G][N] transpose of E, refreshed by k_mh_nz before the P-side updates
  double *P, *E, *A;
  int* R;
  int32_t *ZsumK, *ZsumG, *Z;
  double *Alpha_p, *Beta_p, *Alpha_e, *Beta_e, *Mu_p, *Sig_p, *Mu_e, *Sig_e, *Lam_p, *Lam_e;
  HRef hA_p, hB_p, hC_p, hD_p, hM_p, hS_p, hA_e, hB_e, hC_e, hD_e, hM_e, hS_e;
  double* sigmasq; HRef hAlphaS, hBetaS;   // Normal likelihood: sigmasq_g ~ InvGamma(Alpha_g + K/2, Beta_g + ss/2)
  double *Esum, *Psum, *lpPn, *lpE_part, *colsse, *colll, *colkl;
  const double *lgfact, *logm;       // LUTs over m = 0..maxM: lgamma(m+1), log(max(m,1e-6))
  const double* temperature; long n_temperature;
  double* metrics;                   // device rows [row][BNMF_NMETRIC]
  double* raw;                       // device rows [row][8]: sse, ll, kl, lpE, lpP, sumA (k_reduce -> k_compose)
  size_t lenP, lenE;                 // K*N, N*G: the prior-parameter arrays hold 2 slots, slot(t) = t & 1
};
// record_sample fused into the producers (R/bayesNMF_sampler.R:651-672): the ring slot of the iteration whose values a
// kernel writes, one pointer per recorded array (null = not recorded / window 0).  pp[0..1]: P-side prior parameters in
// the order (Alpha|Mu|Lambda, Beta|Sigmasq), pp[2..3]: E-side.
struct RecDst { double *P, *E, *A, *R; double* pp[4]; };
// prior parameters of iteration t live in slot t&1, so the hyper sweep of iteration t+1 can run
// (on the side stream) while iteration t's values are still being read
template <int SIDE> BNMF_DEV double* slot(const Dev& d, double* base, uint32_t t) { return base + (size_t)(t & 1u) * (SIDE ? d.lenE : d.lenP); }

BNMF_DEV double clamp_tiny(double v) { return (v < 1e-300) ? 1e-300 : v; }

// ---- hand-off of the side kernels' results to k_pdraw WITHOUT a cross-stream event on the main stream ----
// (a stream-wait is a barrier packet that costs ~16 us after k_zalloc even when its event has long fired).  The side
// kernels store what the next draws read with write-through agent-scope stores; every workgroup then drains its stores
// and bumps a counter, and the LAST one publishes the iteration number in a flag.  k_pdraw (next kernel on the main
// stream) polls the flags from one lane per workgroup, bounded, then takes ONE agent-scope acquire.
struct SideDone { unsigned* counter; unsigned* flag; unsigned nblk; unsigned epoch; };
struct SideWait { const unsigned* f0; const unsigned* f1; unsigned epoch; int* err; };
BNMF_DEV void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BNMF_DEV void side_done(const SideDone& sd, int tid) {
  if (!sd.flag) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have left
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(sd.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == sd.nblk - 1) {
      __hip_atomic_store(sd.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sd.flag, sd.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
template <bool FENCE = true>
BNMF_DEV void side_wait(const SideWait& sw, int tid) {
  if (!sw.f0) return;
  if (tid == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(sw.f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sw.epoch ||
           __hip_atomic_load(sw.f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sw.epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) { __hip_atomic_store(sw.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    if (FENCE) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
}
BNMF_DEV double ld_ag(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef ZSPROF
#define DRSTAMP(i) const unsigned long long st##i = __builtin_amdgcn_s_memtime()
#else
#define DRSTAMP(i)
#endif
// ---- hyper sweep of one element: R/sample_priors.R:150-200 (element-wise conditionals) ----
// PRE: `pre` is the Gamma(shape, 1) part of the element's first draw, made by hyper_pre before v was known (k_draw: before the
// wait for P); rgamma(s, shape, rate) is that value divided by the rate, so the result is the same bits.  lut: ralpha_fast's table.
// It also carries the element's hyper-prior values and previous Alpha: loaded before the wait, not behind it.
struct HyperPre { double g, hB, hC, hD, al_old; };
template <int SIDE, bool RETRY_PRIO = false>
BNMF_DEV HyperPre hyper_pre(const Dev& d, int e, uint32_t t) {
  const HRef &hA = SIDE ? d.hA_e : d.hA_p, &hB = SIDE ? d.hB_e : d.hB_p;
  HyperPre p{0.0, hy(hB, e), 0.0, 0.0, 0.0};
  if (d.prior == BNMF_GAMMA) {
    const HRef &hC = SIDE ? d.hC_e : d.hC_p, &hD = SIDE ? d.hD_e : d.hD_p;
    p.hC = hy(hC, e); p.hD = hy(hD, e);
    p.al_old = slot<SIDE>(d, SIDE ? d.Alpha_e : d.Alpha_p, t - 1)[e];
    Stream s(d.k0, d.k1, SIDE ? BNMF_V_BETA_E : BNMF_V_BETA_P, (uint32_t)e, t);
    p.g = rgamma<RETRY_PRIO>(s, hy(hA, e) + p.al_old, 1.0);
    return p;
  }
  Stream s(d.k0, d.k1, SIDE ? BNMF_V_LAMBDA_E : BNMF_V_LAMBDA_P, (uint32_t)e, t);
  p.g = rgamma<RETRY_PRIO>(s, hy(hA, e) + 1.0, 1.0);
  return p;
}
template <int SIDE, bool PRE = false>
BNMF_DEV void hyper_elem(const Dev& d, int e, uint32_t t, double v, double* rec0 = nullptr, double* rec1 = nullptr, const HyperPre& pre = HyperPre{},
                         const double* lut = g_alut) {
  if (d.prior == BNMF_GAMMA) {
    const HRef &hA = SIDE ? d.hA_e : d.hA_p, &hB = SIDE ? d.hB_e : d.hB_p;
    const HRef &hC = SIDE ? d.hC_e : d.hC_p, &hD = SIDE ? d.hD_e : d.hD_p;
    double* Al = SIDE ? d.Alpha_e : d.Alpha_p;
    double* Be = SIDE ? d.Beta_e : d.Beta_p;
    Stream s(d.k0, d.k1, SIDE ? BNMF_V_BETA_E : BNMF_V_BETA_P, (uint32_t)e, t);
    const double al_old = PRE ? pre.al_old : slot<SIDE>(d, Al, t - 1)[e];
    const double b = PRE ? pre.g / (pre.hB + v) : rgamma(s, hy(hA, e) + al_old, hy(hB, e) + v);        // sample_Beta_*  :323-345
    st_wt(&slot<SIDE>(d, Be, t)[e], b);
    if (rec1) rec1[e] = b;
    const double tau = ((PRE ? pre.hD : hy(hD, e)) - dlog(clamp_tiny(b))) - dlog(clamp_tiny(v));
    Stream s2(d.k0, d.k1, SIDE ? BNMF_V_ALPHA_E : BNMF_V_ALPHA_P, (uint32_t)e, t);
    const double al = ralpha_fast(s2, PRE ? pre.hC : hy(hC, e), tau, al_old, nullptr, lut);   // sample_Alpha_* :356-397
    st_wt(&slot<SIDE>(d, Al, t)[e], al);
    if (rec0) rec0[e] = al;
  } else if (d.prior == BNMF_EXPONENTIAL) {
    const HRef &hA = SIDE ? d.hA_e : d.hA_p, &hB = SIDE ? d.hB_e : d.hB_p;
    double* La = SIDE ? d.Lam_e : d.Lam_p;
    Stream s(d.k0, d.k1, SIDE ? BNMF_V_LAMBDA_E : BNMF_V_LAMBDA_P, (uint32_t)e, t);
    const double la = PRE ? pre.g / (pre.hB + v) : rgamma(s, hy(hA, e) + 1.0, hy(hB, e) + v);            // sample_Lambda_* :284-308
    st_wt(&slot<SIDE>(d, La, t)[e], la);
    if (rec0) rec0[e] = la;
  } else {
    const HRef &hM = SIDE ? d.hM_e : d.hM_p, &hS = SIDE ? d.hS_e : d.hS_p;
    const HRef &hA = SIDE ? d.hA_e : d.hA_p, &hB = SIDE ? d.hB_e : d.hB_p;
    double* Mu = SIDE ? d.Mu_e : d.Mu_p;
    double* Sg = SIDE ? d.Sig_e : d.Sig_p;
    const double sg = slot<SIDE>(d, Sg, t - 1)[e];
    const double num = hy(hM, e) / hy(hS, e) + v / sg;
    const double den = 1.0 / hy(hS, e) + 1.0 / sg;
    Stream s(d.k0, d.k1, SIDE ? BNMF_V_MU_E : BNMF_V_MU_P, (uint32_t)e, t);
    const double mu = num / den + (1.0 / den) * rnorm_std(s);               // sd = 1/denom (quirk) :214-236
    st_wt(&slot<SIDE>(d, Mu, t)[e], mu);
    if (rec0) rec0[e] = mu;
    const double dl = v - mu;
    const double rate = (SIDE ? hy(hA, e) : hy(hB, e)) + (dl * dl) / 2.0;   // A_e for B_e (quirk) :263-270
    Stream s2(d.k0, d.k1, SIDE ? BNMF_V_SIGSQ_E : BNMF_V_SIGSQ_P, (uint32_t)e, t);
    const double sgn = rinvgamma(s2, hy(hA, e) + 0.5, rate);
    st_wt(&slot<SIDE>(d, Sg, t)[e], sgn);
    if (rec1) rec1[e] = sgn;
  }
}
// prior draw of an element of P / E: R/sample_Pn.R:12-30, R/sample_En.R:12-30
template <int SIDE>
BNMF_DEV double prior_draw(const Dev& d, int e, uint32_t t) {
  Stream s(d.k0, d.k1, SIDE ? BNMF_V_E : BNMF_V_P, (uint32_t)e, t);
  if (d.prior == BNMF_GAMMA) return rgamma(s, slot<SIDE>(d, SIDE ? d.Alpha_e : d.Alpha_p, t)[e], slot<SIDE>(d, SIDE ? d.Beta_e : d.Beta_p, t)[e]);
  if (d.prior == BNMF_EXPONENTIAL) return rexp(s, slot<SIDE>(d, SIDE ? d.Lam_e : d.Lam_p, t)[e]);
  return rtnorm0(s, slot<SIDE>(d, SIDE ? d.Mu_e : d.Mu_p, t)[e], dsqrt(slot<SIDE>(d, SIDE ? d.Sig_e : d.Sig_p, t)[e]));
}
// log prior density of an element under iteration t's prior parameters: R/utils.R:132-175
template <int SIDE>
BNMF_DEV double prior_logdens(const Dev& d, int e, double x, uint32_t t) {
  if (d.prior == BNMF_GAMMA) {
    const double al = slot<SIDE>(d, SIDE ? d.Alpha_e : d.Alpha_p, t)[e], be = slot<SIDE>(d, SIDE ? d.Beta_e : d.Beta_p, t)[e];
    return ((al * dlog(be) - dlgamma(al)) + (al - 1.0) * dlog(x)) - be * x;
  }
  if (d.prior == BNMF_EXPONENTIAL) {
    const double la = slot<SIDE>(d, SIDE ? d.Lam_e : d.Lam_p, t)[e];
    return dlog(la) - la * x;
  }
  const double mu = slot<SIDE>(d, SIDE ? d.Mu_e : d.Mu_p, t)[e], sg = dsqrt(slot<SIDE>(d, SIDE ? d.Sig_e : d.Sig_p, t)[e]);
  const double zz = (x - mu) / sg;
  return ((-0.91893853320467274178 - dlog(sg)) - 0.5 * (zz * zz)) - dlog_pnorm(mu / sg);
}

// canonical W=1024 sum of x[0], x[stride], ... (L terms) by one 256-lane workgroup: lane i owns
// accumulators i, i+256, i+512, i+768, folds the first two tree levels locally, then a 256-tree.
constexpr int RT = 256;
BNMF_DEV double canon1024_by256(const double* x, long L, long stride, double* buf, int tid) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  // five rounds at a time: their twenty loads are requested before the first is added (one round per round trip made the reductions
  // of G = 10,000 terms ten dependent round trips: 15 us for k_reduce, twice in a row at the end of every bnmf_run).  The additions and
  // their order are the same: accumulator c adds its terms in ascending order, an absent term adds nothing.
  constexpr int RB = 5;
  for (long i0 = tid; i0 < L; i0 += 1024 * RB) {
    double v[RB][4]; bool ok[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) { const long i = i0 + 1024L * r + 256L * c; ok[r][c] = i < L; v[r][c] = ok[r][c] ? x[i * stride] : 0.0; }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (ok[r][0]) a0 = a0 + v[r][0];
      if (ok[r][1]) a1 = a1 + v[r][1];
      if (ok[r][2]) a2 = a2 + v[r][2];
      if (ok[r][3]) a3 = a3 + v[r][3];
    }
  }
  a0 = a0 + a2; a1 = a1 + a3;          // tree level h = 512
  a0 = a0 + a1;                         // tree level h = 256
  return block_tree<RT>(a0, buf, tid);  // valid on thread 0
}

// ---- k_side: everything of iteration t that depends only on P_{t-1}, E_{t-1} ----
// Runs on the side stream concurrently with k_zalloc of iteration t-1.
//   blocks [0, N)            : Esum[n] = canonical sum_g E[n,g]   (rate of P's Gamma, R/sample_Pn.R:103-106)
//   blocks [N, N+nbP)        : hyper sweep of the P-side prior parameters (R/sample_priors.R:150-200)
//   blocks [N+nbP, ...)      : hyper sweep of the E-side prior parameters
BNMF_DEV void side_body(const Dev& d, uint32_t t, int nbP, int blk, int nblocks, const RecDst& rec, const SideDone& sd, double* buf, int tid) {
  // The small launches (P part, Esum: a few dozen workgroups) share the CUs with k_zalloc, whose older waves win the
  // instruction arbitration: without a raised priority these few young waves starve (Esum took 56 us for a 10,000-term
  // reduction) although the event the next k_pdraw waits for hangs on them.  Too few waves to slow k_zalloc down.
  if (nblocks <= 64) __builtin_amdgcn_s_setprio(3);
  if (blk < d.N) {
    const double r = canon1024_by256(d.E + blk, d.G, d.N, buf, tid);
    if (tid == 0) st_wt(&d.Esum[blk], r);
  } else if (blk < d.N + nbP) {
    const long e = (long)(blk - d.N) * RT + tid;
    if (e < (long)d.lenP) hyper_elem<0>(d, (int)e, t, d.P[e], rec.pp[0], rec.pp[1]);
  } else {
    const long e = (long)(blk - d.N - nbP) * RT + tid;
    if (e < (long)d.lenE) hyper_elem<1>(d, (int)e, t, d.E[e], rec.pp[2], rec.pp[3]);
  }
  side_done(sd, tid);
}
// <= 152 registers: a wave of this kernel then still fits beside three waves of the allocation kernel on a SIMD (3 x 120 + 152 = 512;
// round 3 capped it at 128 and paid 28 bytes of scratch per lane)
__global__ __launch_bounds__(RT) __attribute__((amdgpu_num_vgpr(152))) void k_side(Dev d, uint32_t t, int nbP, int blk0, RecDst rec, SideDone sd) {
  __shared__ double buf[RT];
  side_body(d, t, nbP, blockIdx.x + blk0, gridDim.x, rec, sd, buf, threadIdx.x);   // one launch (blk0 = 0) or one launch per part
}

// publishes an epoch in a flag that side_wait polls: for results an EARLIER kernel of the same stream has stored (the stream's order and
// the kernel boundary put them in front of the flag)
__global__ void k_raise_flag(unsigned* flag, unsigned epoch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- k_pdraw: one workgroup of 128 lanes per factor n ----
// sample_Pn_poisson R/sample_Pn.R:98-120 (dispatch :11-42); Psum[n] and the log-prior of column n
// are reduced canonically (W = 64) over k.
constexpr int PD_T = 128;
__global__ __launch_bounds__(PD_T) void k_pdraw(Dev d, uint32_t t, int from_prior, int with_lp, RecDst rec, SideWait sw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  side_wait(sw, threadIdx.x);
  double* Pn = (double*)dyn;          // [K]
  double* lp = Pn + d.K;              // [K]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int K = d.K;
  const double a_n = d.A[n];
  // Esum may have been published while this kernel was already polling: an agent-scope load (never the scalar cache)
  const double Esum = from_prior ? 0.0 : __hip_atomic_load(&d.Esum[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int k = tid; k < K; k += PD_T) {
    const int e = k + K * n;
    double x;
    if (from_prior == 2) x = d.P[e];                    // user-supplied initial value kept verbatim
    else if (from_prior || a_n == 0.0) x = prior_draw<0>(d, e, t);
    else {
      double shape, rate;
      if (d.prior == BNMF_GAMMA) { shape = slot<0>(d, d.Alpha_p, t)[e] + (double)d.ZsumG[e]; rate = slot<0>(d, d.Beta_p, t)[e] + a_n * Esum; }
      else { shape = 1.0 + (double)d.ZsumG[e]; rate = slot<0>(d, d.Lam_p, t)[e] + a_n * Esum; }
      Stream s(d.k0, d.k1, BNMF_V_P, (uint32_t)e, t);
      x = rgamma(s, shape, rate);
    }
    d.P[e] = x;
    if (rec.P) rec.P[e] = x;
    d.ZsumG[e] = 0;                    // consumed; k_zalloc accumulates the next one
    Pn[k] = x;
    if (with_lp) lp[k] = prior_logdens<0>(d, e, x, t);
  }
  if (rec.A && tid == 0) rec.A[n] = a_n;                  // fixed rank: A and R never change after the constructor
  if (rec.R && tid == 0 && n == 0) *rec.R = (double)*d.R;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  if (wave && !with_lp) return;                           // the log-prior is then computed off the critical path (k_lpp)
  const double* src = wave ? lp : Pn;
  double acc = 0.0;
  for (int k = lane; k < K; k += 64) acc = acc + src[k];
  acc = wave_tree64(acc);
  if (lane == 0) (wave ? d.lpPn : d.Psum)[n] = acc;
}

// ---- k_edraw: one lane per element (n,g) of E, flat column-major index e = n + N g ----
// sample_En_poisson R/sample_En.R:97-119; log-prior partial per 256-element block (canonical tree)
constexpr int ES_T = 256;
__global__ __launch_bounds__(ES_T) void k_edraw(Dev d, uint32_t t, int from_prior, int with_lp, double* recE) {
  __shared__ double buf[ES_T];
  const int tid = threadIdx.x;
  const long e = (long)blockIdx.x * ES_T + tid;
  double lp = 0.0;
  if (e < (long)d.lenE) {
    const int n = (int)(e % d.N);
    const double a_n = d.A[n];
    double x;
    if (from_prior == 2) x = d.E[e];                    // user-supplied initial value kept verbatim
    else if (from_prior || a_n == 0.0) x = prior_draw<1>(d, (int)e, t);
    else {
      double shape, rate;
      if (d.prior == BNMF_GAMMA) { shape = slot<1>(d, d.Alpha_e, t)[e] + (double)d.ZsumK[e]; rate = slot<1>(d, d.Beta_e, t)[e] + a_n * d.Psum[n]; }
      else { shape = 1.0 + (double)d.ZsumK[e]; rate = slot<1>(d, d.Lam_e, t)[e] + a_n * d.Psum[n]; }
      Stream s(d.k0, d.k1, BNMF_V_E, (uint32_t)e, t);
      x = rgamma(s, shape, rate);
    }
    d.E[e] = x;
    if (recE) recE[e] = x;
    if (d.zsumk_accum) d.ZsumK[e] = 0;               // consumed; k_zalloc_tile accumulates the next one
    if (with_lp) lp = prior_logdens<1>(d, (int)e, x, t);
  }
  if (!with_lp) return;                                 // the log-prior is then computed off the critical path (k_lpe)
  const double r = block_tree<ES_T>(lp, buf, tid);
  if (tid == 0) d.lpE_part[blockIdx.x] = r;
}

// ---- k_draw: k_pdraw and k_edraw of the steady-state fixed-rank sweep in ONE launch (large problems: api.hip gate_enabled) ----
// Workgroups [0, N): factor n's column of P (k_pdraw's work), Psum[n] stored write-through, the last of them raises the flag pd.
// Workgroups [N, ..): E, DW lanes each: the Gamma(shape, 1) part of the draw needs nothing from P; then one lane waits for the
// flag pd and the workgroup divides by the rate — the same operations in the same order as rgamma(shape, rate).
// A column of P is OWNED by whoever first writes this launch's sequence number into its word of own[] (one atomic exchange): its
// workgroup b < N when it starts — or an E workgroup that has waited for the flag for a while and finds the word still old.
// In-order dispatch makes the second a path never taken (the P workgroups have marked their columns long before an E workgroup
// gets to its wait), but forward progress no longer rests on it: if E workgroups ever held every slot before a P workgroup had
// started (round 3: "assumption, stated, not enforced"), they would draw the columns themselves instead of timing out, and the
// late P workgroup finds its column taken and leaves.  Which workgroup draws a column changes no bit of it (the elements' own
// streams).  seq grows by one per launch of this kernel on the handle, so the words never need a reset.  no_p (tests):
// workgroups [0, N) leave without taking their columns.
// The kernel waits for nothing outside itself: the allocation kernel before it has waited (one lane, at its end) for the hyper
// sweep of this iteration.
constexpr int DW = 1024;
// The E workgroups also run the E-side hyper sweep of iteration t + 1 (hyper_elem<1>: element-wise, it needs nothing but the
// element's own E_t): beside the allocation kernel that sweep got one wave per SIMD and the leftover issue slots (85 us for
// ~20 us of work, and the allocation kernel's gate waited for it); here it runs at full occupancy.  rec_next: ring slots of
// t + 1 for the prior parameters; ed: the E-side flag of the hand-off protocol (k_side raises it when it does this work).
__global__ __launch_bounds__(DW) void k_draw(Dev d, uint32_t t, RecDst rec, SideDone pd, SideWait pw, RecDst rec_next, SideDone ed, unsigned* own /* [N] */, unsigned seq, int no_p) {
  __shared__ double Pn[DW];
  __shared__ double lutS[3 * ALUT_N];                     // E workgroups: ralpha_fast's table (its look-ups are dependent loads inside the Newton and attempt loops)
  __shared__ int jobS;
  const int tid = threadIdx.x, N = d.N, K = d.K;
  const int BW = blockDim.x;                              // <= DW: chosen by the host so that the E workgroups fill the CUs once
  // column n of P by this workgroup (sample_Pn_poisson R/sample_Pn.R:98-120), Psum[n], and its share of the flag pd
  auto p_column = [&](int n) __attribute__((always_inline)) {
    const double a_n = d.A[n];
    const double Esum = d.Esum[n];
    for (int k = tid; k < K; k += BW) {
      const int e = k + K * n;
      double x;
      if (a_n == 0.0) x = prior_draw<0>(d, e, t);
      else {
        double shape, rate;
        if (d.prior == BNMF_GAMMA) { shape = slot<0>(d, d.Alpha_p, t)[e] + (double)d.ZsumG[e]; rate = slot<0>(d, d.Beta_p, t)[e] + a_n * Esum; }
        else { shape = 1.0 + (double)d.ZsumG[e]; rate = slot<0>(d, d.Lam_p, t)[e] + a_n * Esum; }
        Stream s(d.k0, d.k1, BNMF_V_P, (uint32_t)e, t);
        x = rgamma(s, shape, rate);
      }
      d.P[e] = x;
      if (rec.P) rec.P[e] = x;
      d.ZsumG[e] = 0;
      if (k < DW) Pn[k] = x;
    }
    if (rec.A && tid == 0) rec.A[n] = a_n;
    if (rec.R && tid == 0 && n == 0) *rec.R = (double)*d.R;
    __syncthreads();
    if (tid < 64) {
      double acc = 0.0;
      for (int k = tid; k < K; k += 64) acc = acc + (k < DW ? Pn[k] : d.P[k + K * n]);
      acc = wave_tree64(acc);
      if (tid == 0) st_wt(&d.Psum[n], acc);
    }
    side_done(pd, tid);
  };
  auto take = [&](int n) __attribute__((always_inline)) -> bool {                        // one lane: true if column n is now this workgroup's
    return __hip_atomic_exchange(own + n, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq;
  };
  if ((int)blockIdx.x < N) {
    if (no_p) return;
    if (tid == 0) jobS = take((int)blockIdx.x) ? (int)blockIdx.x : -1;
    __syncthreads();
    if (jobS >= 0) p_column(jobS);
    return;
  }
  // ---------------- E workgroups.  Round 5: no workgroup barrier after the head of the kernel.  Stamps of the waves (tools/drstamps.py) showed
  // the barrier behind the wait for P costing every wave 2.4 us (the workgroup's slowest wave: a second pass of a rejection loop), and
  // the kernel ending 6 us behind its median wave (Alpha: 2, 3 or 4 passes).  Now a wave waits for P by itself, the Alpha draws of a wave
  // take two passes (ralpha_fast_wave), and a column of P that nobody has taken is drawn by ONE WAVE (p_column_wave).
  const int lane = tid & 63;
  // column n of P by one wavefront: p_column's operations in p_column's order (lane l takes rows l, l + 64, ...; Psum: the canonical W = 64 sum)
  auto p_column_wave = [&](int n) __attribute__((always_inline)) {
    const double a_n = d.A[n];
    const double Esum = d.Esum[n];
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) {
      const int e = k + K * n;
      double x;
      if (a_n == 0.0) x = prior_draw<0>(d, e, t);
      else {
        double shape, rate;
        if (d.prior == BNMF_GAMMA) { shape = slot<0>(d, d.Alpha_p, t)[e] + (double)d.ZsumG[e]; rate = slot<0>(d, d.Beta_p, t)[e] + a_n * Esum; }
        else { shape = 1.0 + (double)d.ZsumG[e]; rate = slot<0>(d, d.Lam_p, t)[e] + a_n * Esum; }
        Stream s(d.k0, d.k1, BNMF_V_P, (uint32_t)e, t);
        x = rgamma(s, shape, rate);
      }
      d.P[e] = x;
      if (rec.P) rec.P[e] = x;
      d.ZsumG[e] = 0;
      acc = acc + x;
    }
    if (rec.A && lane == 0) rec.A[n] = a_n;
    if (rec.R && lane == 0 && n == 0) *rec.R = (double)*d.R;
    acc = wave_tree64(acc);
    if (lane == 0) st_wt(&d.Psum[n], acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have left (side_done's hand-off, for one wave)
    if (lane == 0 && pd.flag) {
      const unsigned old = __hip_atomic_fetch_add(pd.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == pd.nblk - 1) {
        __hip_atomic_store(pd.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pd.flag, pd.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  // lane 0 of the wave waits for the flag pd; after a while it looks for a column of P that nobody has taken (see the head of the
  // kernel).  Returns the column this wave has to draw, or -1: P is complete.
  auto wait_p = [&]() __attribute__((always_inline)) -> int {
    int job = -1;
    if (lane == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(pw.f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pw.epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) { __hip_atomic_store(pw.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((spins & 7u) == 0u) {
          for (int c = 0; c < N && job < 0; ++c)
            if (__hip_atomic_load(own + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq && take(c)) job = c;
          if (job >= 0) break;
        }
      }
    }
    return __builtin_amdgcn_readfirstlane(job);
  };
  const long e = (long)((int)blockIdx.x - N) * BW + tid;
  const bool live = e < (long)d.lenE;
  const bool gam = d.prior == BNMF_GAMMA;
  // the Alpha table: requested at the head of the kernel together with the element's own values; the one barrier of the E role
  if (gam) for (int i = tid; i < 3 * ALUT_N; i += BW) lutS[i] = g_alut[i];
  __syncthreads();
  // the E role of the wave; returns a column of P to draw first (then the role is run again from its start: the same streams, the same
  // bits — so that no draw has to be kept in registers across the column's code), or -1 when done
  auto e_role = [&]() __attribute__((always_inline)) -> int {
    DRSTAMP(0);
#ifdef ZSLIGHT
    unsigned long long stA = 0;
#endif
    double x = 0.0, a_n = 0.0;
    bool scaled = false;                                 // x is Gamma(shape, 1) and still has to be divided by the rate
    double base = 0.0;                                   // the rate without its Psum term
    HyperPre hpre{};                                     // Gamma(shape, 1) part of the hyper sweep's first draw and the element's hyper-prior values (need nothing from this iteration)
    int n = 0;
    if (live) {
      n = (int)(e % N);
      a_n = d.A[n];
      if (a_n == 0.0) x = prior_draw<1>(d, (int)e, t);
      else {
        double shape;
        if (gam) { shape = slot<1>(d, d.Alpha_e, t)[e] + (double)d.ZsumK[e]; base = slot<1>(d, d.Beta_e, t)[e]; }
        else { shape = 1.0 + (double)d.ZsumK[e]; base = slot<1>(d, d.Lam_e, t)[e]; }
        Stream s(d.k0, d.k1, BNMF_V_E, (uint32_t)e, t);
        x = rgamma<true>(s, shape, 1.0);
        scaled = true;
      }
#ifdef ZSLIGHT
      stA = __builtin_amdgcn_s_memtime();
#endif
      hpre = hyper_pre<1, true>(d, (int)e, t + 1);
    }
    DRSTAMP(1);
    const int job = wait_p();
    if (job >= 0) return job;
    DRSTAMP(2);
    double tau = 0.0;
    if (live) {
      if (scaled) {
        const double rate = base + a_n * ld_ag(&d.Psum[n]);
        // rgamma(shape, rate) ends in `g / rate`; rgamma(shape, 1.0) returned g / 1.0 = g
        x = x / rate;
      }
      d.E[e] = x;
      if (rec.E) rec.E[e] = x;
      if (d.zsumk_accum) d.ZsumK[e] = 0;
    }
    DRSTAMP(3);
    // the E-side hyper sweep of t + 1 (hyper_elem<1, true>'s operations; the Alpha draws of the wave side by side)
    if (gam) {
      if (live) {
        const double b = hpre.g / (hpre.hB + x);                                                    // sample_Beta_En  :339-345
        st_wt(&slot<1>(d, d.Beta_e, t + 1)[e], b);
        if (rec_next.pp[3]) rec_next.pp[3][e] = b;
        tau = (hpre.hD - dlog(clamp_tiny(b))) - dlog(clamp_tiny(x));
      }
      const double al = ralpha_fast_wave(live, d.k0, d.k1, BNMF_V_ALPHA_E, (uint32_t)e, t + 1, hpre.hC, tau, hpre.al_old, lutS);   // sample_Alpha_Eng :382-397
      if (live) {
        st_wt(&slot<1>(d, d.Alpha_e, t + 1)[e], al);
        if (rec_next.pp[2]) rec_next.pp[2][e] = al;
      }
    } else if (live) hyper_elem<1, true>(d, (int)e, t + 1, x, rec_next.pp[2], rec_next.pp[3], hpre, lutS);
#ifdef ZSPROF
    DRSTAMP(4);
    if (live && lane == 0 && (e >> 6) < DRPROF_W - 1) {
      unsigned long long* o = &g_drprof[8 * (e >> 6)];
#ifdef ZSLIGHT
      o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4; o[5] = stA; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = 1ull;   // absolute stamps of the last launch
#else
      o[0] += st1 - st0; o[1] += st2 - st1; o[2] += st3 - st2; o[3] += st4 - st3; o[5] += st4 - st0; o[6] = st0; o[7] += 1ull;
#endif
    }
#endif
    return -1;
  };
  int job = e_role();
  if (__builtin_expect(job >= 0, 0)) {                    // never taken under in-order dispatch
    do { p_column_wave(job); job = wait_p(); } while (job >= 0);
    e_role();                                             // P is complete: runs through
  }
  side_done(ed, tid);
}

// log-prior of column n of P_t under iteration t's prior parameters: canonical W = 64 over k, as in k_pdraw
BNMF_DEV void lpp_body(const Dev& d, uint32_t t, int n, int lane) {
  __builtin_amdgcn_s_setprio(3);                        // N one-wave workgroups beside k_zalloc (see k_side)
  const int K = d.K;
  double acc = 0.0;
  for (int k = lane; k < K; k += 64) { const int e = k + K * n; acc = acc + prior_logdens<0>(d, e, d.P[e], t); }
  acc = wave_tree64(acc);
  if (lane == 0) d.lpPn[n] = acc;
}
__global__ __launch_bounds__(64) void k_lpp(Dev d, uint32_t t) { lpp_body(d, t, blockIdx.x, threadIdx.x); }
// log-prior partials of E_t under iteration t's prior parameters, same 256-element blocks and tree as k_edraw
// Esrc: the E of iteration t — its record_sample ring slot when recording is on (not overwritten for a whole window; d.E is
// overwritten by the next k_edraw, which is not ordered behind this kernel), else d.E (then the host double-buffers E)
BNMF_DEV void lpe_body(const Dev& d, uint32_t t, int blk, const double* Esrc, double* buf, int tid) {
  const long e = (long)blk * ES_T + tid;
  double lp = 0.0;
  if (e < (long)d.lenE) lp = prior_logdens<1>(d, (int)e, Esrc[e], t);
  const double r = block_tree<ES_T>(lp, buf, tid);
  if (tid == 0) d.lpE_part[blk] = r;
}
__global__ __launch_bounds__(ES_T) void k_lpe(Dev d, uint32_t t, const double* Esrc) {
  __shared__ double buf[ES_T];
  lpe_body(d, t, blockIdx.x, Esrc, buf, threadIdx.x);
}
// k_side and the log-prior workgroups of the iteration before in ONE launch (the fixed-rank Gibbs sweep: two runtime calls fewer
// per iteration — small problems are bound by the host's enqueue rate).  Workgroups [0, first): k_side's (blk0 as there);
// then n_lpp workgroups with k_lpp's work (one wave each), then n_lpe with k_lpe's.  The two kinds are independent of each
// other; side_done counts the k_side workgroups only.
struct SideExtra { int first, n_lpp, n_lpe; uint32_t t_lp; const double* Esrc; int count_all; };   // count_all: the log-prior workgroups count towards sd too
static_assert(ES_T == RT, "k_side_lp: one block size for both kinds of workgroup");
// Round 5: workgroups behind the log-prior ones sum the per-column metric terms of the iteration whose allocation kernel has left its Mhat
// behind (colterms.h; ct.mh != null): four pairs of columns each.  They belong to no flag: k_reduce follows this kernel in stream order.
__global__ __launch_bounds__(RT) __attribute__((amdgpu_num_vgpr(152))) void k_side_lp(Dev d, uint32_t t, int nbP, int blk0, RecDst rec, SideDone sd, SideExtra ex, CtArgs ct) {
  __shared__ double buf[RT];
  const int j = (int)blockIdx.x - ex.first;
  if (j < 0) { side_body(d, t, nbP, blockIdx.x + blk0, ex.first, rec, sd, buf, threadIdx.x); return; }
  if (j >= ex.n_lpp + ex.n_lpe) {
    const int p = (j - ex.n_lpp - ex.n_lpe) * (RT / 64) + ((int)threadIdx.x >> 6);
    if (2 * p < ct.G) colterms_pair(ct, 2 * p, (int)threadIdx.x & 63);
    return;
  }
  if (j < ex.n_lpp) { if (threadIdx.x < 64) lpp_body(d, ex.t_lp, j, threadIdx.x); }
  else lpe_body(d, ex.t_lp, j - ex.n_lpp, ex.Esrc, buf, threadIdx.x);
  // the prior parameters these workgroups read (slot t_lp & 1) are overwritten two iterations on: where the writer is released
  // by sd's flag (the merged draw kernel, which runs the E-side sweep itself), the flag has to cover these reads
  if (ex.count_all) side_done(sd, threadIdx.x);
}

// ---- k_zalloc: the general Z-allocation kernel (any N, any K); the metric configuration runs k_zalloc_reg ----
// sample_Zkg R/sample_params.R:253-265 for every cell, fused with Mhat (R/utils.R:29-49) and the
// per-cell RMSE / KL / Poisson log-lik terms (R/utils.R:62-112, :412-471).
//
// ONE WAVEFRONT PER COLUMN, no workgroup barrier inside the column loop.  A workgroup is ZT/64
// independent waves that only share the running Z counts zacc[n][k] (LDS atomics), flushed to
// ZsumG once per workgroup with global integer atomics (exact, order-independent).
// Per column g a wave does, wave-synchronously:
//  phase 1: lane l owns rows l, l+64, ...: p_n = (P[k,n] A[n]) E[n,g] (A is 0/1, so the product
//           with ae[n] = A[n] E[n,g] is bit-identical), cumulative sums, thresholds
//           thr[n][k] = floor(cum_n 2^32 / Mhat) into the wave's LDS slab ([n][k] layout: lanes
//           hold consecutive k, so threshold reads and zacc atomics of different cells fall in
//           different banks), the cell's metric terms (accumulated per lane in exactly the
//           canonical 64-strided order, then a wave tree) and the cell's quad count.
//  phase 2: the column's counts are flattened into quads (4 counts = one Philox block of the
//           cell's stream); each lane takes a CONTIGUOUS range of quads; the 4 counts of a quad
//           run 4 interleaved branch-free binary searches over the cell's thresholds, then one
//           LDS atomic into zacc[n][k] and one into the lane's private packed 8-bit histogram.
//  phase 3: ZsumK[:,g] = wave reduction of the packed histograms.
// LDS hand-off between lanes of ONE wave: DS operations of a wave are executed in issue order,
// so only the compiler has to be stopped from moving LDS accesses across this point.  (A
// workgroup-scope fence would also wait for every outstanding global store: ~microseconds.)
BNMF_DEV void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Workgroup barrier for hand-offs that go through LDS only: every wave waits for its own LDS (and scalar) operations, not for its
// outstanding global loads and stores, as __syncthreads() does — inside a chain of short dependent steps that wait is a round trip to
// L2 per barrier (a store of the step's result, a prefetch for the next step).
BNMF_DEV void wg_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
BNMF_DEV uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
struct __attribute__((aligned(16))) u4 { uint32_t x, y, z, w; };
struct ZGeom { int KP, HW, TR, KC, slab_words, zacc_words, p_words; };
constexpr int ZH = 68;             // pitch of the per-lane histogram rows (16-byte aligned rows)
// General kernel (any N, any K): the rows of a column are processed in chunks of zg.KC rows (a multiple of
// 64; KC >= K, i.e. one chunk, whenever the whole column's thresholds fit the wave's LDS slab).  With more
// than one chunk the workgroup-level zacc[n][k] does not fit LDS either (zg.zacc_words == 0): the chunk's counts
// go to the wave's zloc[n][k - kbase] and its non-zero entries are flushed to ZsumG with global integer
// atomics at the end of the chunk.  Metric accumulators and the ZsumK histogram run across the chunks of a
// column in row order, so the results are bit-identical to the one-chunk mapping.
template <bool SAVE_Z, int ZT>
__global__ __launch_bounds__(ZT) void k_zalloc(Dev d, uint32_t t, ZGeom zg, int ablate) {
#ifndef BNMF_DIAG
  ablate = 0;                        // switching phases off is for the builder's diagnostic builds (-DBNMF_DIAG), never for libbnmf.so
#endif
  constexpr int ZW = ZT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  const int KP = zg.KP, HW = zg.HW, KC = zg.KC;          // KP: pitch of the [n][row] LDS arrays (>= min(K, KC), odd)
  const bool chunked = zg.zacc_words == 0;
  const bool use_loc = SAVE_Z || chunked;
  uint32_t* zacc = (uint32_t*)smem;                      // [N][KP] shared by the workgroup (one-chunk mode only)
  uint32_t* slab = zacc + zg.zacc_words + (size_t)wave * zg.slab_words;
  uint32_t* hist = slab;                                 // [HW][ZH] per-lane packed 8-bit bucket counts (16-B aligned)
  double* ae = (double*)(hist + HW * ZH);                // [N]  A[n] * E[n,g]
  uint32_t* thr = (uint32_t*)(ae + N);                   // [N-1][KP] thresholds of the current chunk
  uint32_t* qoff = thr + (size_t)(N - 1) * KP;           // [KC+1]  quad offset (22 bits) | nlast << 22
  int* mcnt = (int*)(qoff + KC + 1);                     // [KC]
  uint32_t* zkt = (uint32_t*)(mcnt + KC);                // [N] column totals
  uint32_t* zloc = zkt + N;                              // [N][KP]  (SAVE_Z or chunked)
  for (int i = tid; i < zg.zacc_words; i += ZT) zacc[i] = 0;
  for (int i = lane; i < HW * ZH; i += 64) hist[i] = 0;
  for (int i = lane; i < N; i += 64) zkt[i] = 0;
  if (use_loc) for (int i = lane; i < N * KP; i += 64) zloc[i] = 0;
  __syncthreads();
  const int nthr = N - 1;
  uint32_t* ztarget = use_loc ? zloc : zacc;
  const int gw = blockIdx.x * ZW + wave, nw = gridDim.x * ZW;
  for (int g = gw; g < G; g += nw) {
    double a_sse = 0.0, a_ll = 0.0, a_kl = 0.0;
    const double* Eg = d.E + (size_t)N * g;
    for (int n = lane; n < N; n += 64) ae[n] = d.A[n] * Eg[n];
    wave_lds_fence();
    for (int kbase = 0; kbase < K; kbase += KC) {
      const int kend = min(K, kbase + KC), kc = kend - kbase;
      // ---------------- phase 1: thresholds, metric terms and quad counts of rows [kbase, kend)
      int carry = 0;
      for (int r = 0; (r << 6) < kc; ++r) {
        const int cl = (r << 6) + lane, kk = kbase + cl;   // row within the chunk / in the matrix
        int q = 0, nl = -1;
        if (cl < kc) {
          const double* Pk = d.P + kk;
          const int m = d.M[kk + (size_t)K * g];
          double c = 0.0;
          for (int n0 = 0; n0 < N; n0 += 8) {             // 8 independent P loads in flight
            double pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = Pk[(size_t)K * min(n0 + j, N - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (n0 + j < N) {
                const double p = pv[j] * ae[n0 + j];
                c = c + p;
                if (p > 0.0) nl = n0 + j;
              }
            }
          }
          if (c > 0.0 && m > 0 && nl >= 0) {
            const double scale = 4294967296.0 / c;
            double cc = 0.0;
            for (int n0 = 0; n0 < nthr; n0 += 8) {
              double pv[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) pv[j] = Pk[(size_t)K * min(n0 + j, N - 1)];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                if (n0 + j < nthr) {
                  cc = cc + pv[j] * ae[n0 + j];
                  const double tt = cc * scale;
                  thr[(size_t)(n0 + j) * KP + cl] = (n0 + j >= nl || tt >= 4294967295.0) ? 0xFFFFFFFFu : (uint32_t)tt;
                }
              }
            }
            q = (m + 3) >> 2;
          }
          const double dd = c - (double)m;
          const double mh = c < 1e-6 ? 1e-6 : c;
          const double lmh = dlog(mh);
          const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
          const double mt = m < 1 ? 1e-6 : (double)m;
          a_sse = a_sse + dd * dd;                        // canonical: lane l adds rows l, l+64, ...
          a_ll = a_ll + (((double)m * lmh - mh) - d.lgfact[mi]);
          a_kl = a_kl + mt * (d.logm[mi] - lmh);
          mcnt[cl] = q > 0 ? m : 0;
        }
        int incl = q;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (cl < kc) qoff[cl] = (uint32_t)(carry + incl - q) | ((uint32_t)(nl < 0 ? 0 : nl) << 22);
        carry += __shfl(incl, 63, 64);
      }
      const int Q = carry;
      if (lane == 0) qoff[kc] = (uint32_t)Q;
      wave_lds_fence();
      // ---------------- phase 2: lane takes quads [q0, q1) of the chunk, in sub-chunks of <= 63 quads so that
      // the packed 8-bit per-lane histogram cannot overflow (<= 252 counts per flush)
      const int per = (Q + 63) >> 6;
      for (int cbase = 0; cbase < per; cbase += 63) {
        const int q0 = min(Q, lane * per + cbase);
        const int q1 = min(Q, min(lane * per + per, q0 + 63));
        if (q0 < q1) {
          int cell;
          {  // upper_bound(qoff[0..kc] & mask, q0) - 1, branch-free
            int b = 0, len = kc + 1;
            while (len > 1) { const int half = len >> 1; b = ((int)(qoff[b + half - 1] & 0x3FFFFFu) <= q0) ? b + half : b; len -= half; }
            cell = b + ((int)(qoff[b] & 0x3FFFFFu) <= q0 ? 1 : 0) - 1;
          }
          uint32_t qw = qoff[cell];
          int cstart = (int)(qw & 0x3FFFFFu);
          int cend = (int)(qoff[cell + 1] & 0x3FFFFFu);
          int mc = mcnt[cell];
          for (int qi = q0; qi < q1; ++qi) {
            if (qi >= cend) {
              do { ++cell; cstart = cend; cend = (int)(qoff[cell + 1] & 0x3FFFFFu); } while (qi >= cend);
              mc = mcnt[cell];
            }
            const int j0 = (qi - cstart) << 2;
            const int nd = mc - j0;                        // >= 1; draws of this quad = min(4, nd)
            const u32x4 w = philox4x32_7((uint32_t)(j0 >> 2), (uint32_t)(kbase + cell + (size_t)K * g), t, BNMF_V_Z, d.k0, d.k1);
            const uint32_t* col = thr + cell;
            const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
            int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if (nthr > 0) {
              int len = nthr;
              while (len > 1) {                           // 4 interleaved branch-free searches
                const int half = len >> 1, off = half - 1;
                const uint32_t t0 = col[(b0 + off) * KP], t1 = col[(b1 + off) * KP], t2 = col[(b2 + off) * KP], t3 = col[(b3 + off) * KP];
                b0 = (t0 <= u0) ? b0 + half : b0;
                b1 = (t1 <= u1) ? b1 + half : b1;
                b2 = (t2 <= u2) ? b2 + half : b2;
                b3 = (t3 <= u3) ? b3 + half : b3;
                len -= half;
              }
              b0 += (col[b0 * KP] <= u0) ? 1 : 0;
              b1 += (col[b1 * KP] <= u1) ? 1 : 0;
              b2 += (col[b2 * KP] <= u2) ? 1 : 0;
              b3 += (col[b3 * KP] <= u3) ? 1 : 0;
            }
            uint32_t* zc = ztarget + cell;
            uint32_t* hl = hist + lane;
            atomicAdd(&zc[b0 * KP], 1u); atomicAdd(&hl[(b0 >> 2) * ZH], 1u << ((b0 & 3) << 3));
            if (nd > 1) { atomicAdd(&zc[b1 * KP], 1u); atomicAdd(&hl[(b1 >> 2) * ZH], 1u << ((b1 & 3) << 3)); }
            if (nd > 2) { atomicAdd(&zc[b2 * KP], 1u); atomicAdd(&hl[(b2 >> 2) * ZH], 1u << ((b2 & 3) << 3)); }
            if (nd > 3) { atomicAdd(&zc[b3 * KP], 1u); atomicAdd(&hl[(b3 >> 2) * ZH], 1u << ((b3 & 3) << 3)); }
          }
        }
        wave_lds_fence();
        // flush the packed histograms: lane n sums byte (n&3) of word n>>2 over the 64 lanes
        for (int n = lane; n < N; n += 64) {
          const uint32_t* hr = hist + (n >> 2) * ZH;
          const int sh = (n & 3) << 3;
          uint32_t tot = 0;
#pragma unroll
          for (int l4 = 0; l4 < 64; l4 += 4) {
            const u4 v = *(const u4*)(hr + l4);
            tot += ((v.x >> sh) & 0xFFu) + ((v.y >> sh) & 0xFFu) + ((v.z >> sh) & 0xFFu) + ((v.w >> sh) & 0xFFu);
          }
          zkt[n] += tot;
        }
        wave_lds_fence();
        for (int i = lane; i < HW * ZH; i += 64) hist[i] = 0;
        wave_lds_fence();
      }
      // ---------------- end of chunk: Z[kbase:kend, :, g], and the chunk's counts into ZsumG / zacc
      if (use_loc) {
        for (int i = lane; i < kc * N; i += 64) {          // i = cl + kc*n: coalesced Z store
          const int cl = i % kc, n = i / kc;
          const size_t a = (size_t)n * KP + cl;
          const uint32_t z = zloc[a];
          if (SAVE_Z) d.Z[kbase + cl + (size_t)K * (n + (size_t)N * g)] = (int32_t)z;
          if (z) {
            if (chunked) { if (!(ablate & 1)) atomicAdd(&d.ZsumG[kbase + cl + (size_t)K * n], (int32_t)z); }
            else atomicAdd(&zacc[a], z);
            zloc[a] = 0;
          }
        }
      }
      wave_lds_fence();
    }
    a_sse = wave_tree64(a_sse); a_ll = wave_tree64(a_ll); a_kl = wave_tree64(a_kl);
    if (lane == 0) { d.colsse[g] = a_sse; d.colll[g] = a_ll; d.colkl[g] = a_kl; }
    // ---------------- ZsumK[:,g]
    for (int n = lane; n < N; n += 64) { d.ZsumK[n + (size_t)N * g] = (int32_t)zkt[n]; zkt[n] = 0; }
    wave_lds_fence();
  }
  __syncthreads();
  if (!chunked) {
    for (int i = tid; i < K * N; i += ZT) {
      const int kk = i % K, n = i / K;
      const uint32_t v = zacc[(size_t)n * KP + kk];
      if (v && !(ablate & 1)) atomicAdd(&d.ZsumG[kk + (size_t)K * n], (int32_t)v);
    }
  }
}

// ---- k_reduce: canonical (W = 1024) reductions of one iteration's partial sums ----
// blocks 0..3 reduce colsse, colll, colkl, lpE_part; block 0 also folds the per-factor log-prior of
// P and sum(A).  Raw values go to raw[row][8]; k_compose turns them into metrics rows once per run.
// (the partial sums of ONE iteration: the slot pointers of that iteration — k_mh_tail of the iteration after it runs this body too)
struct RedSlots { const double *colsse, *colll, *colkl, *lpE_part, *lpPn, *accPn, *accE_part; int row, on; };
BNMF_DEV void reduce_body(const Dev& d, const RedSlots& rs, int nblkE, int j, double* buf, int tid) {
  const double* src = j == 0 ? rs.colsse : j == 1 ? rs.colll : j == 2 ? rs.colkl : j == 3 ? rs.lpE_part : rs.accE_part;
  const long len = j < 3 ? d.G : nblkE;
  const double r = canon1024_by256(src, len, 1, buf, tid);
  double* o = d.raw + (size_t)rs.row * 8;
  if (tid == 0) o[j < 4 ? j : 6] = r;
  if (j != 0) return;                     // (block-uniform)
  // the sums over the factors, n ascending from +0.0 as before — but the terms are fetched by the lanes side by side (lane 0 walking
  // global memory was one round trip per factor, three times N of them in a row)
  auto seq_sum = [&](auto term) -> double {
    double acc = 0.0;
    for (int base = 0; base < d.N; base += RT) {
      __syncthreads();
      if (base + tid < d.N) buf[tid] = term(base + tid);
      __syncthreads();
      if (tid == 0) for (int i = 0; i < min(RT, d.N - base); ++i) acc = acc + buf[i];
    }
    return acc;                           // valid on thread 0
  };
  const double lpP = seq_sum([&](int n) { return rs.lpPn[n]; });
  if (tid == 0) o[4] = lpP;
  if (!d.learning_rank) {                 // with rank learning A changes on the main stream: k_sumA writes these
    const double sumA = seq_sum([&](int n) { return d.A[n]; });
    if (tid == 0) o[5] = sumA;
    if (rs.accPn) {
      // (the old loop skipped the factors with A[n] != 1: an absent term adds nothing; here it adds +0.0, which leaves every sum of these
      // non-negative rates unchanged bit for bit — the sum starts at +0.0 and never holds -0.0)
      const double sp = seq_sum([&](int n) { return d.A[n] == 1.0 ? rs.accPn[n] : 0.0; });
      if (tid == 0) o[7] = sp;
    }
  }
}
__global__ __launch_bounds__(RT) void k_reduce(Dev d, int row, int nblkE, const double* accPn, const double* accE_part) {
  __shared__ double buf[RT];
  reduce_body(d, RedSlots{d.colsse, d.colll, d.colkl, d.lpE_part, d.lpPn, accPn, accE_part, row, 1}, nblkE, blockIdx.x, buf, threadIdx.x);
}
// sum(A) (and the A-masked acceptance sum) of the iteration, on the main stream right after the rank update
__global__ void k_sumA(Dev d, int row, const double* accPn, RecDst rec) {
  if (rec.A) for (int n = threadIdx.x; n < d.N; n += blockDim.x) rec.A[n] = d.A[n];
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (rec.R) *rec.R = (double)*d.R;
  double* o = d.raw + (size_t)row * 8;
  double sumA = 0.0;
  for (int n = 0; n < d.N; ++n) sumA = sumA + d.A[n];
  o[5] = sumA;
  if (accPn) { double sp = 0.0; for (int n = 0; n < d.N; ++n) if (d.A[n] == 1.0) sp = sp + accPn[n]; o[7] = sp; }
}
// compute_metrics_ R/utils.R:412-455, update_sample_metrics_ :339-348: one lane per recorded row
__global__ void k_compose(Dev d, int nrows, uint32_t t0) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  const double* r = d.raw + (size_t)row * 8;
  const uint32_t t = t0 + (uint32_t)row;
  const double sse = r[0], ll = r[1], kl = r[2], lpE = r[3], lpP = r[4], sumA = r[5];
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
  o[9] = d.MH ? r[7] / ((double)d.K * sumA) : BNMF_NAN;    // mean acceptance over active factors (R/utils.R:444-452)
  o[10] = d.MH ? r[6] / ((double)d.G * sumA) : BNMF_NAN;
}

// ---- k_record: record_sample (R/bayesNMF_sampler.R:651-672) into the device ring buffer ----
// Copies the iteration's P, E, A, R and prior parameters into slot (t-1) % window of each ring.
struct RecArgs { const double* src[12]; double* dst[12]; size_t len[12]; int n; const int* R; double* Rdst; };
__global__ void k_record(RecArgs a) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (int j = 0; j < a.n; ++j)
    for (size_t i = tid; i < a.len[j]; i += nth) a.dst[j][i] = a.src[j][i];
  if (tid == 0 && a.Rdst) *a.Rdst = (double)*a.R;
}


// ---- MAP window statistics on the device: get_MAP_ (R/utils.R:194-288) ----
// colSums(P) of every used sample (renormalize, R/helpers.R:35-49): one wave per (sample, factor)
__global__ __launch_bounds__(64) void k_map_colsum(const double* ringP, size_t lenP, int K, int N, const int* slots, double* cs) {
  const int s = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const double* Pn = ringP + (size_t)slots[s] * lenP + (size_t)K * n;
  double acc = 0.0;
  for (int k = lane; k < K; k += 64) acc = acc + Pn[k];
  acc = wave_tree64(acc);
  if (lane == 0) cs[(size_t)s * N + n] = acc;
}
// One lane per element of P (SIDE 0: x = P / colsum) or E (SIDE 1: x = E * colsum): mean over the used samples in
// sample order, and the two type-7 quantiles (stats::quantile default) from the kt smallest / kt largest values.
// The lane keeps both sets UNSORTED in LDS (arrays [kt][64], lane-private columns: conflict-free) with the current
// worst member (value, position) in registers: a value that enters replaces the worst one and the set is re-scanned
// with independent loads (a sorted insertion is a chain of dependent LDS round trips, and some lane of the wave
// inserts at nearly every sample).  The two small sets are sorted once at the end.  kt = 0: mean only.
BNMF_DEV void lane_sort(double* a, int n) {           // insertion sort of a[0], a[64], ... (n <= 160)
  for (int i = 1; i < n; ++i) {
    const double x = a[i * 64];
    int j = i - 1;
    while (j >= 0 && a[j * 64] > x) { a[(j + 1) * 64] = a[j * 64]; --j; }
    a[(j + 1) * 64] = x;
  }
}
template <int SIDE>
__global__ __launch_bounds__(64) void k_map_stats(const double* ring, size_t len, int K, int N, const int* slots, int n_used,
                                                  const double* cs, int kt, int jlo, double glo, int jhi, double ghi,
                                                  double* mean, double* lower, double* upper) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const size_t e = (size_t)blockIdx.x * 64 + tid;
  if (e >= len) return;
  const int n = SIDE ? (int)(e % (size_t)N) : (int)(e / (size_t)K);
  double* lo = (double*)smem + tid;                     // lo[i * 64]: the kt smallest so far
  double* hi = (double*)smem + (size_t)kt * 64 + tid;   // hi[i * 64]: the kt largest so far
  int cnt = 0, lop = 0, hip = 0;
  double lov = 0.0, hiv = 0.0;                          // largest of the small set / smallest of the large set
  double sum = 0.0;
  for (int s0 = 0; s0 < n_used; s0 += 8) {
    double v[8], c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                       // 8 independent loads in flight per lane
      const int s = min(s0 + j, n_used - 1);
      v[j] = ring[(size_t)slots[s] * len + e];
      c[j] = cs[(size_t)s * N + n];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (s0 + j >= n_used) break;
      const double x = SIDE ? v[j] * c[j] : v[j] / c[j];
      sum = sum + x;
      if (kt == 0) continue;
      if (cnt < kt) {                                   // filling: both sets hold everything seen so far
        lo[cnt * 64] = x; hi[cnt * 64] = x;
        if (cnt == 0 || x > lov) { lov = x; lop = cnt; }
        if (cnt == 0 || x < hiv) { hiv = x; hip = cnt; }
        ++cnt;
      } else {
        if (x < lov) {
          lo[lop * 64] = x;
          double m = lo[0]; int p = 0;
          for (int i = 1; i < kt; ++i) { const double w = lo[i * 64]; if (w > m) { m = w; p = i; } }
          lov = m; lop = p;
        }
        if (x > hiv) {
          hi[hip * 64] = x;
          double m = hi[0]; int p = 0;
          for (int i = 1; i < kt; ++i) { const double w = hi[i * 64]; if (w < m) { m = w; p = i; } }
          hiv = m; hip = p;
        }
      }
    }
  }
  mean[e] = sum / (double)n_used;
  if (kt) {
    lane_sort(lo, cnt); lane_sort(hi, cnt);
    const int base = n_used - cnt;                      // hi[i] is order statistic base + i
    const int jl1 = min(jlo + 1, n_used - 1), jh1 = min(jhi + 1, n_used - 1);
    lower[e] = (1.0 - glo) * lo[jlo * 64] + glo * lo[jl1 * 64];
    upper[e] = (1.0 - ghi) * hi[(jhi - base) * 64] + ghi * hi[(jh1 - base) * 64];
  }
}
// ---- k_map_quant: the credible bounds of get_MAP_ (R/utils.R:269-284, quantile type 7) by a wave per element (round 4) ----
// k_map_stats keeps the kt smallest / largest values of an element per LANE, and some lane of the wave replaces one at nearly
// every sample: the whole wave then walks the kt-entry scan (7.5 ms for 202,000 elements x 1,000 samples).  Here a workgroup
// takes 8 consecutive elements: their samples are read as whole 64-byte lines (sample s: elements e0 .. e0 + 7; the slot list sits
// in LDS and a lane has 8 lines in flight), renormalised (the same expression as k_map_stats) and laid out [element][S] in LDS
// (S = 64 R: n_used rounded up, the rest +inf).  A wave takes two of the elements: lane l sorts its own samples l, l + 64, ... (a
// column of R values) in REGISTERS (a bitonic network of min / max pairs) and writes the column back; then the order statistics needed
// — a few dozen from each end at the default interval — are drawn one by one as the minimum (maximum) over the lanes' column heads,
// the four chains (two elements, two ends) side by side in one loop: each is a row of dependent wave reductions.  The same values, the
// same interpolation.
constexpr int MQ_T = 512, MQ_E = 8;
// minimum / maximum over the wave, as a wave-uniform value: the GFX9 DPP reduction (row_shr 1, 2, 4, 8: lane 15 of a row holds the row's;
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's), v_min_f64 / v_max_f64 issued as they are
// (the compiler's fmin quiets both operands first); lanes a shift leaves without a source hold values no later step of lane 63's tree reads
template <int CTRL, int ROWS>
BNMF_DEV double dpp_mov_d(double v) {
  const int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32);
  const int a = __builtin_amdgcn_mov_dpp(lo, CTRL, ROWS, 0xf, true), b = __builtin_amdgcn_mov_dpp(hi, CTRL, ROWS, 0xf, true);
  return __longlong_as_double(((long long)b << 32) | (unsigned)a);
}
BNMF_DEV double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
BNMF_DEV double vmax_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
BNMF_DEV double wave_lane63(double v) {
  const int lo = __builtin_amdgcn_readlane((int)__double_as_longlong(v), 63), hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
#define BNMF_WAVE_EXTREMUM(NAME, OP)                                                                               \
BNMF_DEV double NAME(double v) {                                                                                   \
  v = OP(dpp_mov_d<0x111, 0xf>(v), v);                                                                             \
  v = OP(dpp_mov_d<0x112, 0xf>(v), v);                                                                             \
  v = OP(dpp_mov_d<0x114, 0xf>(v), v);                                                                             \
  v = OP(dpp_mov_d<0x118, 0xf>(v), v);                                                                             \
  v = OP(dpp_mov_d<0x142, 0xa>(v), v);                                                                             \
  v = OP(dpp_mov_d<0x143, 0xc>(v), v);                                                                             \
  return wave_lane63(v);                                                                                           \
}
BNMF_WAVE_EXTREMUM(wave_min_all, vmin_f64)
BNMF_WAVE_EXTREMUM(wave_max_all, vmax_f64)
#undef BNMF_WAVE_EXTREMUM
// ascending sort of R values held in registers (R a power of two; every index below is a compile-time constant after unrolling)
template <int R>
BNMF_DEV void reg_sort(double (&a)[R]) {
#pragma unroll
  for (int k = 2; k <= R; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const double x = a[i], y = a[l];
          const bool sw = ((i & k) == 0) ? (x > y) : (x < y);
          a[i] = sw ? y : x; a[l] = sw ? x : y;
        }
      }
    }
  }
}
template <int SIDE, int R>
__global__ __launch_bounds__(MQ_T) void k_map_quant(const double* ring, size_t len, int K, int N, const int* slots, int n_used,
                                                    const double* cs, int jlo, double glo, int jhi, double ghi, double* mean, double* lower, double* upper) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int S = 64 * R;
  constexpr bool NIOK = (S / (MQ_T / MQ_E)) % 8 == 0;
  double* tile = (double*)smem;                           // [MQ_E][S]
  int* sl = (int*)(tile + (size_t)MQ_E * S);              // [S] ring slots of the samples
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t e0 = (size_t)blockIdx.x * MQ_E;
  for (int s = tid; s < n_used; s += MQ_T) sl[s] = slots[s];
  __syncthreads();
  {
    constexpr int SPR = MQ_T / MQ_E, NI = S / SPR;        // samples per round of the workgroup; rounds
    const int j = tid & (MQ_E - 1), sb = tid >> 3;        // samples sb, sb + SPR, ...: NI of them
    const size_t e = e0 + j;
    const bool live = e < len;
    const size_t ec = live ? e : len - 1;
    const int n = SIDE ? (int)(ec % (size_t)N) : (int)(ec / (size_t)K);
    constexpr int B = 8;
#pragma unroll 1
    for (int i0 = 0; i0 < NI; i0 += B) {
      double v[B], c[B];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int s = sb + SPR * (i0 + b);
        const int sc = min(s, n_used - 1);                  // every load is made (no branch around it): 8 lines in flight
        v[b] = ring[(size_t)sl[sc] * len + ec];
        c[b] = cs[(size_t)sc * N + n];
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int s = sb + SPR * (i0 + b);
        tile[(size_t)j * S + s] = (live && s < n_used) ? (SIDE ? v[b] * c[b] : v[b] / c[b]) : __builtin_inf();
      }
    }
  }
  __syncthreads();
  const int cnt = lane < n_used ? (n_used - lane + 63) >> 6 : 0;      // samples of this lane's column
  const int jl1 = min(jlo + 1, n_used - 1), jh1 = min(jhi + 1, n_used - 1);
  constexpr int EW = MQ_E / (MQ_T / 64);                  // elements of a wave
  static_assert(EW >= 1 && NIOK, "a wave takes whole elements");
  // the means (get_MAP_: the sum in sample order, as k_map_stats forms it) while the samples still lie in sample order: lane q of the
  // wave walks element q of the wave — 2 lanes for a few microseconds against a second pass over the window by another kernel
  if (mean && lane < EW) {
    const double* x = tile + (size_t)(wave * EW + lane) * S;
    double sum = 0.0;
    int s0 = 0;
    for (; s0 + 8 <= n_used; s0 += 8) {
      double v[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) v[b] = x[s0 + b];
#pragma unroll
      for (int b = 0; b < 8; ++b) sum = sum + v[b];
    }
    for (; s0 < n_used; ++s0) sum = sum + x[s0];
    const size_t e = e0 + (size_t)(wave * EW + lane);
    if (e < len) mean[e] = sum / (double)n_used;
  }
  double* col[EW];
#pragma unroll
  for (int q = 0; q < EW; ++q) {
    col[q] = tile + (size_t)(wave * EW + q) * S + lane;   // col[64 r]: the lane's column
    double a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = col[q][64 * r];
    reg_sort<R>(a);
#pragma unroll
    for (int r = 0; r < R; ++r) col[q][64 * r] = a[r];
  }
  wave_lds_fence();
  // the column heads (lov / hiv), the values behind them (lon / hin: loaded when the head moves, a whole reduction before they are
  // needed) and their positions; no branch in a step, so that the chains of a wave interleave
  double x0[EW] = {}, x1[EW] = {}, y0[EW] = {}, y1[EW] = {}, lov[EW], hiv[EW], lon[EW], hin[EW];
  int pl[EW], ph[EW];
  const double INF = __builtin_inf();
#pragma unroll
  for (int q = 0; q < EW; ++q) {
    pl[q] = 0; ph[q] = cnt - 1;
    lov[q] = col[q][0];                                   // +inf where the column is empty
    lon[q] = col[q][64];
    hiv[q] = cnt > 0 ? col[q][64 * ph[q]] : -INF;
    hin[q] = cnt > 1 ? col[q][64 * (ph[q] - 1)] : -INF;
  }
  const int nlo = jl1 + 1, nhi = n_used - jhi;           // order statistics 0 .. jl1 ascending, n_used - 1 .. jhi descending
  auto low_step = [&](int c) {
#pragma unroll
    for (int q = 0; q < EW; ++q) {
      const double m = wave_min_all(lov[q]);
      if (c == jlo) x0[q] = m;
      if (c == jl1) x1[q] = m;
      const unsigned long long eq = __builtin_amdgcn_ballot_w64(lov[q] == m);
      const bool win = lane == (int)__builtin_ctzll(eq);  // the first lane that holds it gives it up
      lov[q] = win ? lon[q] : lov[q];
      pl[q] += win ? 1 : 0;
      const double nx = col[q][64 * min(pl[q] + 1, R - 1)];
      lon[q] = pl[q] + 1 < R ? nx : INF;
    }
  };
  auto high_step = [&](int c) {
    const int os = n_used - 1 - c;
#pragma unroll
    for (int q = 0; q < EW; ++q) {
      const double m = wave_max_all(hiv[q]);
      if (os == jh1) y1[q] = m;
      if (os == jhi) y0[q] = m;
      const unsigned long long eq = __builtin_amdgcn_ballot_w64(hiv[q] == m);
      const bool win = lane == (int)__builtin_ctzll(eq);
      hiv[q] = win ? hin[q] : hiv[q];
      ph[q] -= win ? 1 : 0;
      const double nx = col[q][64 * max(ph[q] - 1, 0)];
      hin[q] = ph[q] - 1 >= 0 ? nx : -INF;
    }
  };
  const int nboth = min(nlo, nhi);
  for (int c = 0; c < nboth; ++c) { low_step(c); high_step(c); }
  for (int c = nboth; c < nlo; ++c) low_step(c);
  for (int c = nboth; c < nhi; ++c) high_step(c);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < EW; ++q) {
      const size_t e = e0 + (size_t)(wave * EW + q);
      if (e < len) {
        lower[e] = (1.0 - glo) * x0[q] + glo * x1[q];
        upper[e] = (1.0 - ghi) * y0[q] + ghi * y1[q];
      }
    }
  }
}
// compute_metrics_(P = MAP$P, A = MAP$A, E = MAP$E, MAP = TRUE) (R/utils.R:412-455): per-column squared error and
// padded KL of Mhat = P diag(A) E; one wave per column
__global__ __launch_bounds__(256) void k_map_fit(const int32_t* M, const double* P, const double* A, const double* E, int K, int N, int G,
                                                  double* colsse, double* colkl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x * 4 + wave;
  if (g >= G) return;
  double sse = 0.0, kl = 0.0;
  for (int k = lane; k < K; k += 64) {
    double c = 0.0;
    for (int n = 0; n < N; ++n) c = c + (P[k + (size_t)K * n] * A[n]) * E[n + (size_t)N * g];
    const double m = (double)M[k + (size_t)K * g];
    const double d = c - m;
    sse = sse + d * d;
    const double mt = m < 1e-6 ? 1e-6 : m, mh = c < 1e-6 ? 1e-6 : c;
    kl = kl + mt * dlog(mt / mh);
  }
  sse = wave_tree64(sse); kl = wave_tree64(kl);
  if (lane == 0) { colsse[g] = sse; colkl[g] = kl; }
}

// ---- posterior reference assignment: pairwise_sim of every used sample's included signatures with the
// reference catalogue (lsa::cosine, R/helpers.R:218-268).  One workgroup per (sample, signature); lane = reference
// signature j (refT is the catalogue stored row-major [k][j], so the lanes read consecutive addresses).
__global__ __launch_bounds__(128) void k_ref_cosine(const double* ringP, size_t lenP, int K, const int* slots, const int* sig, int nk,
                                                     const double* refT, const double* refnorm2, int R, double* out /* [s][i][j] */) {
  const int s = blockIdx.x, i = blockIdx.y;
  const double* Pn = ringP + (size_t)slots[s] * lenP + (size_t)K * sig[i];
  for (int j = threadIdx.x; j < R; j += blockDim.x) {
    double dot = 0.0, nn = 0.0;
    for (int k = 0; k < K; ++k) { const double p = Pn[k]; dot = dot + p * refT[(size_t)k * R + j]; nn = nn + p * p; }
    out[((size_t)s * nk + i) * R + j] = dot / dsqrt(nn * refnorm2[j]);
  }
}

// ---- hungarian_assignment of one sample (R/helpers.R:287-398: RcppHungarian::HungarianSolver(-sim)) — one wave per sample (round 4) ----
// The rectangular assignment problem, n rows <= m columns, minimising sum cost[row][col(row)], by the shortest-augmenting-path method
// with potentials (O(n^2 m)): the columns are spread over the lanes (the scan for the next column and the update of the potentials are
// the two loops over m), ties go to the lowest column index, every floating-point operation is the one the sequential algorithm makes —
// the assignment is the sequential algorithm's.  cost[row][col] = -cos[sig][ref]: rows are the kept signatures (tr = 0), or, with more
// signatures than references, the references (tr = 1).  out[s][row] = column of the row.
__global__ __launch_bounds__(64) void k_hungarian(const double* cosv /* [s][nk][R] */, int nk, int R, int tr, int32_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = tr ? R : nk, m = tr ? nk : R;
  const int lane = threadIdx.x;
  const double* c = cosv + (size_t)blockIdx.x * nk * R;
  double* v = (double*)smem;                               // [m + 1]
  double* minv = v + (m + 1);                              // [m + 1]
  double* u = minv + (m + 1);                              // [n + 1]
  int* p = (int*)(u + (n + 1));                            // [m + 1] row matched to the column (0: none)
  int* way = p + (m + 1);                                  // [m + 1]
  int* usedc = way + (m + 1);                              // [m + 1]
  const double INF = 1e300;
  auto cost = [&](int row, int col) { return tr ? -c[(size_t)col * R + row] : -c[(size_t)row * R + col]; };
  for (int j = lane; j <= m; j += 64) { v[j] = 0.0; p[j] = 0; way[j] = 0; }
  for (int i = lane; i <= n; i += 64) u[i] = 0.0;
  wave_lds_fence();
  for (int i = 1; i <= n; ++i) {
    if (lane == 0) p[0] = i;
    for (int j = lane; j <= m; j += 64) { minv[j] = INF; usedc[j] = 0; }
    wave_lds_fence();
    int j0 = 0;
    do {
      if (lane == 0) usedc[j0] = 1;
      wave_lds_fence();
      const int i0 = p[j0];
      const double ui0 = u[i0];
      double delta = INF;
      int j1 = 0x7fffffff;
      for (int j = lane + 1; j <= m; j += 64)              // ascending j inside the lane: the first minimum stays
        if (!usedc[j]) {
          const double cur = (cost(i0 - 1, j - 1) - ui0) - v[j];
          double mv = minv[j];
          if (cur < mv) { mv = cur; minv[j] = cur; way[j] = j0; }
          if (mv < delta) { delta = mv; j1 = j; }
        }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {                  // the smallest value, the lowest column among equals
        const double d2 = __shfl_xor(delta, o, 64);
        const int k2 = __shfl_xor(j1, o, 64);
        if (d2 < delta || (d2 == delta && k2 < j1)) { delta = d2; j1 = k2; }
      }
      for (int j = lane; j <= m; j += 64) {
        if (usedc[j]) { u[p[j]] = u[p[j]] + delta; v[j] = v[j] - delta; }
        else minv[j] = minv[j] - delta;
      }
      wave_lds_fence();
      if (j1 > m) return;                                  // no finite reduced cost (a cosine is NaN): the rows stay -1, the host reports it
      j0 = j1;
    } while (p[j0] != 0);
    if (lane == 0) { do { const int jn = way[j0]; p[j0] = p[jn]; j0 = jn; } while (j0); }
    wave_lds_fence();
  }
  for (int j = lane + 1; j <= m; j += 64) if (p[j]) out[(size_t)blockIdx.x * n + (p[j] - 1)] = j - 1;
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
  x[e] = rgamma(s, hy(hs, (int)e), hy(hr, (int)e));   // x already points at slot(1)
}

// LUTs lgamma(m+1), log(max(m,1e-6)) for m = 0..maxM
__global__ void k_luts(double* lgfact, double* logm, int maxM) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m > maxM) return;
  lgfact[m] = dlgamma((double)m + 1.0);
  logm[m] = dlog(m < 1 ? 1e-6 : (double)m);
}

// ---- probes for the parity tests ----
// holds its stream for about `us` microseconds (s_memrealtime ticks at 100 MHz): lets a test make a side-stream kernel late
__global__ void k_debug_delay(int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
}
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
    case 6: r = ralpha_fast(s, a[i], b[i], c[i]); break;
    case 4: r = runif(s); break;
    case 5: r = rexp(s, a[i]); break;
    default: r = BNMF_NAN;
  }
  out[i] = r;
}
__global__ void k_test_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out, int rounds) {
  const u32x4 w = rounds == 7 ? philox4x32_7(c0, c1, c2, c3, k0, k1) : philox4x32_10(c0, c1, c2, c3, k0, k1);
  out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = w.w;
}

}  // namespace bnmf
