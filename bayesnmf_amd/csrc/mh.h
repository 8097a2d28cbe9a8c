// bayesnmf_amd/csrc/mh.h — Poisson likelihood with Metropolis-Hastings (truncnormal / exponential prior).
//
// sample_Pn -> sample_Pn_normal(as_proposal = TRUE) -> MH_Pn_poisson (R/sample_Pn.R:11-42, :54-87,
// :132-187, :199-248) and the E mirror (R/sample_En.R).  Factors are updated for n = 1..N in order and
// Mhat = P diag(A) E follows the current state (see the stream spec in DESIGN.md).
//   P side: the conditional of row k needs sums over all columns g, but rows are independent within the sweep
//           -> ONE launch, one 1024-lane workgroup per row doing all N sequential updates (k_mh_prow).
//   E side: the conditional of E[n,g] needs sums over k only -> ONE launch, one wave per column doing
//           all N sequential updates locally (k_mh_ecol), followed by the column's metric terms.
//   Mhat is maintained incrementally inside each sweep (fresh at its start): O(N K G) per iteration.
// All sums follow the canonical orders of the stream spec, so proposals and accept/reject masks are
// bit-identical to the oracle.
#pragma once

namespace bnmf {

constexpr int MH_SEG = 320;             // columns per segment of the canonical row sums (5 per lane)

BNMF_DEV double dpois_log(int m, double lam, double lgf) {      // get_loglik_ poisson branch R/utils.R:98-106
  const double mh = lam < 1e-6 ? 1e-6 : lam;
  return ((double)m * dlog(mh) - mh) - lgf;
}
BNMF_DEV double dnorm_log_sd(double x, double mean, double var) { // dnorm(x, mean, sqrt(var), log = TRUE): metric rows
  const double sd = dsqrt(var);
  const double z = (x - mean) / sd;
  return (-0.91893853320467274178 - dlog(sd)) - 0.5 * (z * z);
}
// The four log-likelihood terms of one cell of the MH ratio (MH_Pn_poisson R/sample_Pn.R:213-231): Poisson at the
// proposed / current Mhat (m1 / m0) and the Normal approximations dnorm(m; mean, var = max(Mhat, 1)) in variance form,
// -(log(2 pi) + log var)/2 - (m - mean)^2 / (2 var).  For Mhat >= 1 the variance IS the clamped Poisson mean, so
// log var is the Poisson term's logarithm; L0 = log(max(m0, 1e-6)) of the current state is carried from step to step
// (after an accepted proposal it is this step's L1): one logarithm per cell and step instead of four.
struct MhTerms { double pn, nold, po, nnew, L1; };
BNMF_DEV MhTerms mh_cell_terms(int m, double m0, double m1, double L0, double lgf, double LOG1) {
  const double mh0 = m0 < 1e-6 ? 1e-6 : m0, mh1 = m1 < 1e-6 ? 1e-6 : m1;
  const double L1 = dlog(mh1);
  const double v0 = m0 < 1.0 ? 1.0 : m0, v1 = m1 < 1.0 ? 1.0 : m1;
  const double lv0 = m0 < 1.0 ? LOG1 : L0, lv1 = m1 < 1.0 ? LOG1 : L1;
  const double d0 = (double)m - m0, d1 = (double)m - m1;
  MhTerms r;
  r.pn = ((double)m * L1 - mh1) - lgf;                                                     // loglik_poisson_new :216-218
  r.nold = (-0.91893853320467274178 - 0.5 * lv1) - 0.5 * ((d0 * d0) / v1);                 // loglik_normal_old  :219-224
  r.po = ((double)m * L0 - mh0) - lgf;                                                     // loglik_poisson_old :213-215
  r.nnew = (-0.91893853320467274178 - 0.5 * lv0) - 0.5 * ((d1 * d1) / v0);                 // loglik_normal_new  :225-231
  r.L1 = L1;
  return r;
}
BNMF_DEV double mh_log_clamped(double mhat) { return dlog(mhat < 1e-6 ? 1e-6 : mhat); }
// proposal from the Normal full conditional (or the prior): get_mu_sigmasq_*_normal :132-187
template <int SIDE>
BNMF_DEV double mh_prior_or_cond(const Dev& d, int e, uint32_t t, bool use_prior, double num1, double den) {
  if (use_prior) return prior_draw<SIDE>(d, e, t);
  double mu, var;
  if (d.prior == BNMF_EXPONENTIAL) {
    const double la = slot<SIDE>(d, SIDE ? d.Lam_e : d.Lam_p, t)[e];
    mu = (num1 - la) / den; var = 1.0 / den;
  } else {
    const double mp = slot<SIDE>(d, SIDE ? d.Mu_e : d.Mu_p, t)[e], sg = slot<SIDE>(d, SIDE ? d.Sig_e : d.Sig_p, t)[e];
    const double den2 = den + 1.0 / sg;
    mu = (num1 + mp / sg) / den2; var = 1.0 / den2;
  }
  Stream s(d.k0, d.k1, SIDE ? BNMF_V_E : BNMF_V_P, (uint32_t)e, t);
  return rtnorm0(s, mu, dsqrt(var));
}

// What a factor's update needs that depends on nothing computed inside the sweep: rt_pre of its proposal stream, its prior
// parameters, and the uniform of its accept / reject step.  Made for all factors of a row / column before the N sequential
// updates start (one factor per lane), so that the chain of dependent steps holds no Philox block, no quantile and no load.
struct DrawPre { RtPre r; double p0, p1, umh, q0, q1; };   // q0 = p0 / p1, q1 = 1 / p1 (truncated-normal prior: the conditional's two divisions by Sigmasq)
template <int SIDE>
BNMF_DEV DrawPre draw_pre(const Dev& d, int e, uint32_t t, bool mhstep) {
  DrawPre q;
  q.r = rt_pre(d.k0, d.k1, SIDE ? BNMF_V_E : BNMF_V_P, (uint32_t)e, t);
  if (d.prior == BNMF_EXPONENTIAL) { q.p0 = slot<SIDE>(d, SIDE ? d.Lam_e : d.Lam_p, t)[e]; q.p1 = 0.0; }
  else { q.p0 = slot<SIDE>(d, SIDE ? d.Mu_e : d.Mu_p, t)[e]; q.p1 = slot<SIDE>(d, SIDE ? d.Sig_e : d.Sig_p, t)[e]; }
  q.q0 = q.q1 = 0.0;
  if (d.prior != BNMF_EXPONENTIAL) { q.q0 = q.p0 / q.p1; q.q1 = 1.0 / q.p1; }
  q.umh = 0.0;
  if (mhstep) { Stream su(d.k0, d.k1, SIDE ? BNMF_V_MHU_E : BNMF_V_MHU_P, (uint32_t)e, t); q.umh = runif(su); }
  return q;
}
BNMF_DEV void pre_store(double* a, const DrawPre& q) { a[0] = q.r.z; a[1] = q.r.lu; a[2] = q.r.u2; a[3] = q.p0; a[4] = q.p1; a[5] = q.umh; a[6] = q.q0; a[7] = q.q1; }
BNMF_DEV DrawPre pre_load(const double* a) { return DrawPre{RtPre{a[0], a[1], a[2]}, a[3], a[4], a[5], a[6], a[7]}; }
constexpr int PRE_W = 8;                                  // doubles per factor
// mh_prior_or_cond (and the prior_draw it falls back to) on a DrawPre: the same operations in the same order
template <int SIDE>
BNMF_DEV double mh_prior_or_cond_pre(const Dev& d, int e, uint32_t t, bool use_prior, double num1, double den, const DrawPre& q) {
  Stream s(d.k0, d.k1, SIDE ? BNMF_V_E : BNMF_V_P, (uint32_t)e, t);
  if (use_prior) {                                        // prior_draw<SIDE>
    if (d.prior == BNMF_EXPONENTIAL) return -q.r.lu / q.p0;
    return rtnorm0_pre(s, q.r, q.p0, dsqrt(q.p1));
  }
  double mu, var;
  if (d.prior == BNMF_EXPONENTIAL) { mu = (num1 - q.p0) / den; var = 1.0 / den; }
  else {
    const double den2 = den + q.q1;
    mu = (num1 + q.q0) / den2; var = 1.0 / den2;
  }
  return rtnorm0_pre(s, q.r, mu, dsqrt(var));
}

// ---- Round 5: what followed the two sweeps (k_mh_tail, below) as workgroups HOSTED by the sweep kernels themselves ----
// k_mh_tail ran alone between the column sweep of t and the row sweep of t + 1 (18 us of config 3's 187).  Nothing in it needs both
// sweeps: what reads P_t (the P-side hyper sweep of t + 1, the log-prior of P_t, its acceptance sums, record_sample's P-side arrays)
// can run BESIDE the column sweep of t, which leaves P alone; what reads E_t (the E-side hyper sweep of t + 1, the log-prior of E_t,
// Esum, record_sample's other arrays) BESIDE the row sweep of t + 1, which leaves E alone and occupies 96 of the 256 CUs.  The
// hand-overs are the kernel boundaries themselves: no flag, no second stream.  What the row sweep needs of E_t — its transpose Et and
// the all(E[n, ] == 0) flags — the column sweep writes as it finishes a column; the flags live in two buffers by iteration parity
// (ecol(t) sets nzE[t & 1], prow(t + 1) reads it; prow(t) counts into nzP[t & 1], ecol(t) reads it) and the hosted workgroups clear
// the buffer of the other parity.  k_reduce's work for iteration t - 1 rides beside the column sweep of t, as it rode in k_mh_tail(t).
struct MhETail {             // E side of iteration t: hosted by k_mh_prow of t + 1 (or k_mh_etail at the end of a call)
  int on;                    // 0: only the clearing of nz_zero
  uint32_t t;
  int nbE, nblkE, nrec;      // 256-lane units of the E-side hyper sweep, of the log-prior sums, of the record copy
  const double* accE; double* accE_part; double* lpE_part;    // (the slots of iteration t: the hosting launch carries those of t + 1)
  RecArgs ra;
  int* nz_zero;              // [N] the flags the NEXT column sweep sets
};
struct MhPTail {             // P side of iteration t: hosted by the column sweep of t
  int on;
  uint32_t t;
  int nbP, nrec;
  const double* accP; double* accPn;
  RecArgs ra;
  RedSlots rs; int nblkE;    // k_reduce's work for the iteration before (rs.on)
  int* nz_zero;              // [N] the counters the NEXT row sweep adds to
};
inline int mh_etail_groups(const MhETail& x, int N, int Q) {   // hosted workgroups of Q 256-lane units each (roles padded to whole workgroups)
  if (!x.on) return 1;
  return (x.nbE + Q - 1) / Q + (x.nblkE + Q - 1) / Q + (N + Q - 1) / Q + (x.nrec + Q - 1) / Q + 1;
}
inline int mh_ptail_blocks(const MhPTail& x, int N, int MH) { return !x.on ? 1 : x.nbP + N + x.nrec + 1 + (x.rs.on ? (MH ? 5 : 4) : 0); }
// T lanes = T / 256 units side by side; every lane of the workgroup takes the same role (the roles' barriers are workgroup barriers)
template <int T>
BNMF_DEV void mh_etail_body(const Dev& d, const MhETail& x, int hw, int tid, double* buf) {
  constexpr int Q = T / ES_T;
  static_assert(ES_T == RT, "one unit width for the hyper sweep's, the log-prior's and the reductions' blocks");
  if (!x.on) { if (tid < d.N) x.nz_zero[tid] = 0; return; }
  const int q = tid / ES_T, lt = tid % ES_T;
  double* qb = buf + q * ES_T;
  const int wE = (x.nbE + Q - 1) / Q, wL = (x.nblkE + Q - 1) / Q, wS = (d.N + Q - 1) / Q, wR = (x.nrec + Q - 1) / Q;
  if (hw < wE) {                                            // hyper sweep of the E-side prior parameters of t + 1 (k_side's E part)
    const long e = ((long)hw * Q + q) * RT + lt;
    if (e < (long)d.lenE) hyper_elem<1>(d, (int)e, x.t + 1, d.E[e], nullptr, nullptr);
    return;
  }
  hw -= wE;
  if (hw < wL) {                                            // log-prior / acceptance partial sums of 256 elements of E_t (k_lp_e)
    const int be = hw * Q + q;
    const long e = (long)be * ES_T + lt;
    double lp = 0.0, ac = 0.0;
    if (be < x.nblkE && e < (long)d.lenE) {
      lp = prior_logdens<1>(d, (int)e, d.E[e], x.t);
      if (x.accE) ac = (d.A[e % d.N] == 1.0) ? x.accE[e] : 0.0;
    }
    const double r = block_tree<ES_T>(lp, qb, lt);
    if (lt == 0 && be < x.nblkE) x.lpE_part[be] = r;
    if (x.accE) {
      __syncthreads();
      const double r2 = block_tree<ES_T>(ac, qb, lt);
      if (lt == 0 && be < x.nblkE) x.accE_part[be] = r2;
    }
    return;
  }
  hw -= wL;
  if (hw < wS) {                                            // Esum of t + 1 (k_side's first blocks)
    const int n = hw * Q + q;
    const bool ok = n < d.N;
    const double r = canon1024_by256(d.E + (ok ? n : 0), ok ? (long)d.G : 0L, d.N, qb, lt);
    if (ok && lt == 0) st_wt(&d.Esum[n], r);
    return;
  }
  hw -= wS;
  if (hw < wR) {                                            // record_sample of t: every array but the P-side ones
    const size_t rt = (size_t)hw * T + tid, nth = (size_t)wR * T;
    for (int j = 0; j < x.ra.n; ++j)
      for (size_t i = rt; i < x.ra.len[j]; i += nth) x.ra.dst[j][i] = x.ra.src[j][i];
    if (rt == 0 && x.ra.Rdst) *x.ra.Rdst = (double)*x.ra.R;
    return;
  }
  if (tid < d.N) x.nz_zero[tid] = 0;
}
BNMF_DEV void mh_ptail_body(const Dev& d, const MhPTail& x, int hb, int tid, double* buf) {   // one 256-lane block
  if (!x.on) { if (tid < d.N) x.nz_zero[tid] = 0; return; }
  if (hb < x.nbP) {                                         // hyper sweep of the P-side prior parameters of t + 1 (k_side's P part)
    const long e = (long)hb * RT + tid;
    if (e < (long)d.lenP) hyper_elem<0>(d, (int)e, x.t + 1, d.P[e], nullptr, nullptr);
    return;
  }
  hb -= x.nbP;
  if (hb < d.N) {                                           // log-prior of column n of P_t and its acceptance sum (k_lp_p)
    if (tid >= 64) return;
    const int n = hb;
    double a = 0.0, b = 0.0;
    for (int k = tid; k < d.K; k += 64) { const int e = k + d.K * n; a = a + prior_logdens<0>(d, e, d.P[e], x.t); if (x.accP) b = b + x.accP[e]; }
    a = wave_tree64(a); b = wave_tree64(b);
    if (tid == 0) { d.lpPn[n] = a; if (x.accPn) x.accPn[n] = b; }
    return;
  }
  hb -= d.N;
  if (hb < x.nrec) {                                        // record_sample of t: the P-side arrays
    const size_t rt = (size_t)hb * ES_T + tid, nth = (size_t)x.nrec * ES_T;
    for (int j = 0; j < x.ra.n; ++j)
      for (size_t i = rt; i < x.ra.len[j]; i += nth) x.ra.dst[j][i] = x.ra.src[j][i];
    return;
  }
  hb -= x.nrec;
  if (hb == 0) { if (tid < d.N) x.nz_zero[tid] = 0; return; }
  if (x.rs.on) reduce_body(d, x.rs, x.nblkE, hb - 1, buf, tid);
}
// the E side of the LAST iteration of a call (no row sweep behind it to host it)
__global__ __launch_bounds__(1024) void k_mh_etail(Dev d, MhETail x) {
  __shared__ double buf[1024];
  mh_etail_body<1024>(d, x, (int)blockIdx.x, (int)threadIdx.x, buf);
}

// nzE[n] = number of non-zero entries in row n of E (all(E[n,] == 0) test of sample_Pn_normal :56); also writes
// the transpose Et[g + G n] = E[n, g] that the P-side kernels read (E is constant during the P updates): there a
// wave's lanes walk consecutive columns g of ONE row, and E[n + N g] / M[k + K g] would be 64 cache lines per load
__global__ void k_mh_nz(Dev d, int* nzE) {
  const int n = blockIdx.x;
  int c = 0;
  for (int g = threadIdx.x; g < d.G; g += blockDim.x) {
    const double e = d.E[n + (size_t)d.N * g];
    d.Et[g + (size_t)d.G * n] = e;
    c += e != 0.0 ? 1 : 0;
  }
  if (c) atomicAdd(&nzE[n], c);
}
// nzP[n] = number of non-zero entries in column n of P (all(P[,n] == 0) test of sample_En_normal :56), after the P sweep
__global__ void k_mh_nzp(Dev d, int* nzP) {
  const int n = blockIdx.x;
  int c = 0;
  for (int k = threadIdx.x; k < d.K; k += blockDim.x) c += d.P[k + (size_t)d.K * n] != 0.0 ? 1 : 0;
  if (c) atomicAdd(&nzP[n], c);
}

// ---- P side: ONE launch, one workgroup of 1024 lanes per row k ----
// Within the P sweep the rows of P are mutually independent (Mhat[k, .] depends on P[k, .] only), so row k runs its
// N sequential factor updates alone: the reductions over the G columns are workgroup reductions, not grid-wide ones.
// The row's Mhat lives in `mhrow` (global scratch [K][G], L2-resident) and is maintained incrementally (stream spec:
// fresh at the start of the sweep, then Mhat_no_n = Mhat - (P[k,n] A[n]) E[n,g] and Mhat = Mhat_no_n + (P_new A[n]) E).
// Canonical row sums: segments of MH_SEG = 320 columns (lane l of the owning wave adds columns l, l+64, ... of the
// segment, then the wave tree), segments added in ascending order by lane 0.  Segment s belongs to wave s % 16; a
// lane owns the same cells for the whole sweep, so Mhat needs no synchronisation.
constexpr int MHP_T = 1024;
constexpr int MHP_W = MHP_T / 64;
constexpr int MH_CPL = MH_SEG / 64;                       // cells per lane and segment
// REG (G <= 16 segments, i.e. one segment per wave): the lane's 5 cells of Mhat, the counts and the current factor's
// exposures stay in registers for the whole sweep, the next factor's exposures are requested one factor ahead; otherwise
// Mhat lives in `mhrow`.
template <bool NORMAL, bool REG, bool MHSTEP /* the Metropolis-Hastings step runs (after convergence) */>
__global__ __launch_bounds__(MHP_T) void k_mh_prow(Dev d, uint32_t t, int S, const int* nzE, int* nzP, double* accP, double* mhrow, double* mhlog, SideWait sw, MhETail et) {
  constexpr int mhstep = MHSTEP ? 1 : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (!NORMAL && (int)blockIdx.x >= d.K) {              // hosted: the E side of what followed the column sweep of t - 1 (the rows come first in the grid)
    if (!NORMAL) mh_etail_body<MHP_T>(d, et, (int)blockIdx.x - d.K, (int)threadIdx.x, (double*)smem);
    return;
  }
#ifdef ZSPROF
  const unsigned long long mhsK0 = __builtin_amdgcn_s_memrealtime();
#endif
  side_wait(sw, threadIdx.x);                           // prior parameters of iteration t (k_side on the side stream)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N, k = blockIdx.x;
  double* part = (double*)smem;                         // [4][S] segment partial sums
  double* pa = part + 4 * (size_t)S;                    // [N] P[k,j] * A[j]
  double* bc = pa + N;                                  // [2] broadcast: proposal, accept flag
  double* pcur = bc + 2;                                // [N] the row's current P[k, .] (thread 0 keeps it): for nzP at the end
  double* prq = pcur + N;                               // [N][PRE_W] the factors' DrawPre
  double* anz = prq + PRE_W * N;                        // [2][N] A[n] and the all(E[n,] == 0) flags: read from LDS inside the chain, not by scalar loads
  double* lgc = anz + 2 * N + tid;                          // REG && MHSTEP: [MH_CPL][MHP_T] this step's candidate logarithms
  double* row = mhrow + (size_t)k * G;
  double* lrow = mhlog + (size_t)k * G;                 // !REG: log(max(Mhat, 1e-6)) of the row; candidates at lrow + K G
  double* lcand = lrow + (size_t)K * G;
  const double LOG1 = dlog(1.0);
  const int32_t* Mk = d.Mt + (size_t)G * k;             // M[k, g] at Mt[g + G k]
  for (int j = tid; j < N; j += MHP_T) {
    const double pj = d.P[k + (size_t)K * j]; pa[j] = pj * d.A[j]; pcur[j] = pj;
    anz[j] = d.A[j]; anz[N + j] = nzE[j] == 0 ? 1.0 : 0.0;
  }
  wg_lds_barrier();                                     // pa, anz are complete
#ifdef ZSPROF
  const unsigned long long mhsKa = __builtin_amdgcn_s_memrealtime();
#endif
  // the factors' DrawPre by the LAST wave, one factor per lane (one pass for N <= 64; one factor per wave took two rounds at N = 20, with
  // every other lane of the workgroup waiting), while the other waves form the row's Mhat below: the first reader (thread 0, behind the first
  // barrier of the first factor step) finds them complete — every wave passes that barrier only after this code
  if (wave == MHP_W - 1) for (int j = lane; j < N; j += 64) pre_store(prq + PRE_W * j, draw_pre<0>(d, k + K * j, t, MHSTEP));
  double mh[REG ? MH_CPL : 1], enr[REG ? MH_CPL : 1], enx[REG ? MH_CPL : 1], sgr[(REG && NORMAL) ? MH_CPL : 1];
  double lg[(REG && MHSTEP) ? MH_CPL : 1];              // log(max(Mhat, 1e-6)) of the lane's cells
  int mr[REG ? MH_CPL : 1];
  const int g0r = wave * MH_SEG + lane;                 // REG: this lane's cells are g0r + 64 i
  // fresh Mhat of the row (factor order), by the lane that owns the cell
  if (REG) {
    // factor by factor with the lane's cells side by side (each cell's sum still runs in factor order) and EVERY load made — a cell beyond the
    // row's end reads column G - 1 and is zeroed afterwards —: MH_CPL loads in flight per factor, four factors unrolled.  Cell by cell, every load
    // waited for the one before it: 12.5 us of the kernel's 87 (stamps)
    bool okc[MH_CPL]; int gi[MH_CPL];
#pragma unroll
    for (int i = 0; i < MH_CPL; ++i) {
      const int g = g0r + 64 * i;
      okc[i] = wave < S && g < min(G, (wave + 1) * MH_SEG);
      gi[i] = okc[i] ? g : G - 1;
      mh[i] = 0.0;
    }
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
      const double paj = pa[j];
      const double* Ej = d.Et + (size_t)G * j;
#pragma unroll
      for (int i = 0; i < MH_CPL; ++i) mh[i] = mh[i] + paj * Ej[gi[i]];
    }
#pragma unroll
    for (int i = 0; i < MH_CPL; ++i) {
      mr[i] = 0; enx[i] = 0.0; enr[i] = 0.0; if (MHSTEP) lg[i] = 0.0; if (NORMAL) sgr[i] = 1.0;
      if (okc[i]) {
        mr[i] = Mk[gi[i]]; enx[i] = d.Et[gi[i]];         // exposures of factor 0
        if (MHSTEP) lg[i] = mh_log_clamped(mh[i]);
        if (NORMAL) sgr[i] = d.sigmasq[gi[i]];
      } else mh[i] = 0.0;
    }
  } else {
    for (int s = wave; s < S; s += MHP_W) {
      const int g0 = s * MH_SEG, gend = min(G, g0 + MH_SEG);
      for (int g = g0 + lane; g < gend; g += 64) {
        double c = 0.0;
        for (int j = 0; j < N; ++j) c = c + pa[j] * d.Et[g + (size_t)G * j];
        row[g] = c;
        if (mhstep) lrow[g] = mh_log_clamped(c);
      }
    }
  }
#ifdef ZSPROF
#define MHSTAMP(i) const unsigned long long mhs##i = __builtin_amdgcn_s_memrealtime()
#else
#define MHSTAMP(i)
#endif
#ifdef ZSPROF
  const unsigned long long mhsK1 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int n = 0; n < N; ++n) {
    MHSTAMP(0);
    const int e = k + K * n;
    const double a_n = anz[n];
    if (REG) {
#pragma unroll
      for (int i = 0; i < MH_CPL; ++i) enr[i] = enx[i];
      if (n + 1 < N) {
#pragma unroll
        for (int i = 0; i < MH_CPL; ++i) { const int g = g0r + 64 * i; if (wave < S && g < min(G, (wave + 1) * MH_SEG)) enx[i] = d.Et[g + (size_t)G * (n + 1)]; }
      }
    }
    if (a_n == 0.0) { if (tid == 0) { const double x = prior_draw<0>(d, e, t); d.P[e] = x; pcur[n] = x; } continue; }          // sample_Pn :12
    const bool allzero = anz[N + n] != 0.0;
    const double pold = pa[n];                                                             // P[k,n] * A[n]
    const double* En = d.Et + (size_t)G * n;
    if (!allzero) {
      if (REG) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < MH_CPL; ++i) {
          const int g = g0r + 64 * i;
          if (wave < S && g < min(G, (wave + 1) * MH_SEG)) {
            const double en = enr[i], mhv = mh[i];
            const double mno = mhv - pold * en;
            const double V = NORMAL ? sgr[i] : mhv;
            const double rV = 1.0 / V;                     // one reciprocal serves both sums (stream spec)
            a0 = a0 + en * (((double)mr[i] - mno) * rV);
            a1 = a1 + (a_n * (en * en)) * rV;
          }
        }
        if (wave < S) {
          a0 = wave_tree64(a0); a1 = wave_tree64(a1);
          if (lane == 0) { part[wave] = a0; part[S + wave] = a1; }
        }
      } else {
        for (int s = wave; s < S; s += MHP_W) {
          const int g0 = s * MH_SEG, gend = min(G, g0 + MH_SEG);
          double a0 = 0.0, a1 = 0.0;
          for (int g = g0 + lane; g < gend; g += 64) {
            const double en = En[g];
            const double mhv = row[g];
            const double mno = mhv - pold * en;                                            // Mhat_no_n
            const double V = NORMAL ? d.sigmasq[g] : mhv;                                  // sigmasq_kg :137-147
            const double rV = 1.0 / V;
            a0 = a0 + en * (((double)Mk[g] - mno) * rV);                                   // :155-161
            a1 = a1 + (a_n * (en * en)) * rV;                                              // :163-169
          }
          a0 = wave_tree64(a0); a1 = wave_tree64(a1);
          if (lane == 0) { part[s] = a0; part[S + s] = a1; }
        }
      }
    }
    MHSTAMP(1);
    wg_lds_barrier();
    MHSTAMP(2);
    if (tid == 0) {
      double num1 = 0.0, den = 0.0;
      if (!allzero) for (int s = 0; s < S; ++s) { num1 = num1 + part[s]; den = den + part[S + s]; }
      bc[0] = mh_prior_or_cond_pre<0>(d, e, t, allzero, num1, den, pre_load(prq + PRE_W * n));
    }
    MHSTAMP(3);
    wg_lds_barrier();
    MHSTAMP(4);
    const double pr = bc[0];
    const double pnew = pr * a_n;
    bool take = true;
    if (mhstep) {                                                                          // MH_Pn_poisson :206-247
      if (REG) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int i = 0; i < MH_CPL; ++i) {
          const int g = g0r + 64 * i;
          if (wave < S && g < min(G, (wave + 1) * MH_SEG)) {
            const double en = enr[i];
            const double m0 = mh[i], m1 = (m0 - pold * en) + pnew * en;
            const int m = mr[i];
            const MhTerms tm = mh_cell_terms(m, m0, m1, lg[MHSTEP ? i : 0], d.lgfact[m], LOG1);
            lgc[i * MHP_T] = tm.L1;
            a0 = a0 + tm.pn; a1 = a1 + tm.nold; a2 = a2 + tm.po; a3 = a3 + tm.nnew;
          }
        }
        if (wave < S) {
          a0 = wave_tree64(a0); a1 = wave_tree64(a1); a2 = wave_tree64(a2); a3 = wave_tree64(a3);
          if (lane == 0) { part[wave] = a0; part[S + wave] = a1; part[2 * S + wave] = a2; part[3 * S + wave] = a3; }
        }
      } else {
        for (int s = wave; s < S; s += MHP_W) {
          const int g0 = s * MH_SEG, gend = min(G, g0 + MH_SEG);
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
          for (int g = g0 + lane; g < gend; g += 64) {
            const double en = En[g];
            const double m0 = row[g], m1 = (m0 - pold * en) + pnew * en;
            const int m = Mk[g];
            const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
            const MhTerms tm = mh_cell_terms(m, m0, m1, lrow[g], d.lgfact[mi], LOG1);
            lcand[g] = tm.L1;
            a0 = a0 + tm.pn; a1 = a1 + tm.nold; a2 = a2 + tm.po; a3 = a3 + tm.nnew;
          }
          a0 = wave_tree64(a0); a1 = wave_tree64(a1); a2 = wave_tree64(a2); a3 = wave_tree64(a3);
          if (lane == 0) { part[s] = a0; part[S + s] = a1; part[2 * S + s] = a2; part[3 * S + s] = a3; }
        }
      }
      wg_lds_barrier();
      if (tid == 0) {
        double A_ = 0.0, B_ = 0.0, C_ = 0.0, D_ = 0.0;
        for (int s = 0; s < S; ++s) { A_ = A_ + part[s]; B_ = B_ + part[S + s]; C_ = C_ + part[2 * S + s]; D_ = D_ + part[3 * S + s]; }
        double ratio = dexp((A_ + B_) - (C_ + D_));
        if (ratio > 1.0) ratio = 1.0;                                                      // pmin(accept_ratio, 1) :239
        accP[e] = ratio;
        bc[1] = (prq[PRE_W * n + 5] < ratio) ? 1.0 : 0.0;                                    // runif of stream (MHU_P, e, t)
      }
      wg_lds_barrier();
      take = bc[1] != 0.0;
    } else if (tid == 0 && accP) accP[e] = 1.0;                                            // :201-204
    if (take) {
      if (REG) {
#pragma unroll
        for (int i = 0; i < MH_CPL; ++i) { mh[i] = (mh[i] - pold * enr[i]) + pnew * enr[i]; if (MHSTEP) lg[i] = lgc[i * MHP_T]; }   // cells beyond G hold zeros: unchanged
      } else {
        for (int s = wave; s < S; s += MHP_W) {
          const int g0 = s * MH_SEG, gend = min(G, g0 + MH_SEG);
          for (int g = g0 + lane; g < gend; g += 64) { const double en = En[g]; row[g] = (row[g] - pold * en) + pnew * en; if (mhstep) lrow[g] = lcand[g]; }
        }
      }
      if (tid == 0) { d.P[e] = pr; pa[n] = pnew; pcur[n] = pr; }
    }
#ifdef ZSPROF
    { MHSTAMP(5);
      if (blockIdx.x == 0 && tid == 0 && n < 32) {
        unsigned long long* o = &g_drprof[8 * (2048 + n)];
        o[0] += mhs1 - mhs0; o[1] += mhs2 - mhs1; o[2] += mhs3 - mhs2; o[3] += mhs4 - mhs3; o[4] += mhs5 - mhs4; o[5] += mhs5 - mhs0; o[7] += 1ull;
      } }
#endif
    // no barrier here: the next factor's writers of `part` / `bc` are behind barriers that every wave reaches only after it has read
    // this factor's values (tid 0 reads `part` before the barrier that publishes bc; bc is rewritten only behind the next one)
  }
#ifdef ZSPROF
  if ((blockIdx.x == 0 || blockIdx.x == 50 || blockIdx.x == 95) && tid == 0) {
    const unsigned long long mhsK2 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = &g_drprof[8 * (2048 + 129 + (blockIdx.x == 0 ? 0 : blockIdx.x == 50 ? 1 : 2))];
    o[0] += mhsK1 - mhsK0; o[1] += mhsK2 - mhsK1; o[2] += mhsKa - mhsK0; o[7] += 1ull;
  }
#endif
  wg_lds_barrier();                                        // pcur is complete
  // nzP[n] = number of non-zero entries of column n of P after the sweep (all(P[,n] == 0) test of sample_En_normal :56);
  // zeroed by k_mh_tail of the previous iteration (or the host before the first one)
  for (int j = tid; j < N; j += MHP_T) if (pcur[j] != 0.0) atomicAdd(&nzP[j], 1);
}

// ---- E side: one wave per column, all factors in order; METRICS_ONLY skips the updates (iteration 1) ----
// The column's Mhat[., g] lives in the wave's LDS array mhc[K] (entry kk belongs to lane kk & 63) and is maintained
// incrementally like the rows of the P side.
constexpr int MHE_T = 256;
template <bool METRICS_ONLY>
__global__ __launch_bounds__(MHE_T) void k_mh_ecol(Dev d, uint32_t t, int mhstep, const int* nzP, double* accE, int draw_sig) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  const int KR = (K + 63) >> 6;
  double* ec = (double*)smem + (size_t)wave * (2 * N + 3 * K);   // [N] current column of E
  double* av = ec + N;                                        // [N] A
  double* mhc = av + N;                                       // [K] Mhat[., g]
  double* l0c = mhc + K;                                      // [K] log(max(Mhat, 1e-6)), carried from step to step (MH)
  double* l1c = l0c + K;                                      // [K] this step's candidates
  const double LOG1 = dlog(1.0);
  const int gw = blockIdx.x * (MHE_T / 64) + wave, nw = gridDim.x * (MHE_T / 64);
  for (int g = gw; g < G; g += nw) {
    for (int j = lane; j < N; j += 64) { ec[j] = d.E[j + (size_t)N * g]; av[j] = d.A[j]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool normal = d.likelihood == BNMF_NORMAL;
    double sg_col = normal ? d.sigmasq[g] : 1.0;
    if (!METRICS_ONLY) {
      for (int r = 0; r < KR; ++r) {                          // fresh Mhat of the column
        const int kk = (r << 6) + lane;
        if (kk < K) {
          double c = 0.0;
          for (int j = 0; j < N; ++j) c = c + (d.P[kk + (size_t)K * j] * av[j]) * ec[j];
          mhc[kk] = c;
          if (mhstep) l0c[kk] = mh_log_clamped(c);
        }
      }
      for (int n = 0; n < N; ++n) {
        const int e = n + N * g;
        const double a_n = av[n];
        if (a_n == 0.0) {                                                                  // sample_En :12
          const double x = prior_draw<1>(d, e, t);
          if (lane == 0) { ec[n] = x; d.E[e] = x; }
          continue;
        }
        const bool allzero = nzP[n] == 0;
        const double eold = ec[n];
        const double* Pn = d.P + (size_t)K * n;
        double s1 = 0.0, s2 = 0.0;
        if (!allzero) {
          for (int r = 0; r < KR; ++r) {
            const int kk = (r << 6) + lane;
            if (kk < K) {
              const double pn = Pn[kk];
              const double mh = mhc[kk];
              const double mno = mh - (pn * a_n) * eold;
              const double V = normal ? sg_col : mh;
              const double rV = 1.0 / V;
              s1 = s1 + pn * (((double)d.M[kk + (size_t)K * g] - mno) * rV);
              s2 = s2 + (a_n * (pn * pn)) * rV;
            }
          }
          s1 = wave_bcast0(wave_tree64(s1)); s2 = wave_bcast0(wave_tree64(s2));
        }
        const double pr = mh_prior_or_cond<1>(d, e, t, allzero, s1, s2);
        bool take = true;
        if (mhstep) {
          double A_ = 0.0, B_ = 0.0, C_ = 0.0, D_ = 0.0;
          for (int r = 0; r < KR; ++r) {
            const int kk = (r << 6) + lane;
            if (kk < K) {
              const double pna = Pn[kk] * a_n;
              const double m0 = mhc[kk], m1 = (m0 - pna * eold) + pna * pr;
              const int m = d.M[kk + (size_t)K * g];
              const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
              const MhTerms tm = mh_cell_terms(m, m0, m1, l0c[kk], d.lgfact[mi], LOG1);
              l1c[kk] = tm.L1;
              A_ = A_ + tm.pn; B_ = B_ + tm.nold; C_ = C_ + tm.po; D_ = D_ + tm.nnew;
            }
          }
          A_ = wave_bcast0(wave_tree64(A_)); B_ = wave_bcast0(wave_tree64(B_));
          C_ = wave_bcast0(wave_tree64(C_)); D_ = wave_bcast0(wave_tree64(D_));
          double ratio = dexp((A_ + B_) - (C_ + D_));
          if (ratio > 1.0) ratio = 1.0;
          if (lane == 0) accE[e] = ratio;
          Stream su(d.k0, d.k1, BNMF_V_MHU_E, (uint32_t)e, t);
          take = runif(su) < ratio;
        } else if (lane == 0 && accE) accE[e] = 1.0;
        if (take) {
          for (int r = 0; r < KR; ++r) {
            const int kk = (r << 6) + lane;
            if (kk < K) { const double pna = Pn[kk] * a_n; mhc[kk] = (mhc[kk] - pna * eold) + pna * pr; if (mhstep) l0c[kk] = l1c[kk]; }
          }
          if (lane == 0) { ec[n] = pr; d.E[e] = pr; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (normal && draw_sig) {
      // sample_sigmasq R/sample_params.R:275-286: sigmasq_g ~ InvGamma(Alpha_g + K/2, Beta_g + sum_k resid^2 / 2)
      double ss = 0.0;
      for (int r = 0; r < KR; ++r) {
        const int kk = (r << 6) + lane;
        if (kk < K) {
          double c = 0.0;
          for (int j = 0; j < N; ++j) c = c + (d.P[kk + (size_t)K * j] * av[j]) * ec[j];
          const double rr = (double)d.M[kk + (size_t)K * g] - c;
          ss = ss + rr * rr;
        }
      }
      ss = wave_bcast0(wave_tree64(ss));
      Stream s(d.k0, d.k1, BNMF_V_SIGMASQ, (uint32_t)g, t);
      sg_col = rinvgamma(s, hy(d.hAlphaS, g) + (double)K / 2.0, hy(d.hBetaS, g) + 0.5 * ss);
      if (lane == 0) d.sigmasq[g] = sg_col;
    }
    // metric terms of the column with the fresh Mhat (R/utils.R:412-471)
    double a_sse = 0.0, a_ll = 0.0, a_kl = 0.0;
    for (int r = 0; r < KR; ++r) {
      const int kk = (r << 6) + lane;
      if (kk < K) {
        double c = 0.0;
        for (int j = 0; j < N; ++j) c = c + (d.P[kk + (size_t)K * j] * av[j]) * ec[j];
        const int m = d.M[kk + (size_t)K * g];
        const double dd = c - (double)m;
        const double mh = c < 1e-6 ? 1e-6 : c;
        const double lmh = dlog(mh);
        const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
        const double mt = m < 1 ? 1e-6 : (double)m;
        a_sse = a_sse + dd * dd;
        if (normal) a_ll = a_ll + dnorm_log_sd((double)m, c, sg_col);           // get_loglik_ normal branch R/utils.R:72-97
        else a_ll = a_ll + (((double)m * lmh - mh) - d.lgfact[mi]);
        a_kl = a_kl + mt * (d.logm[mi] - lmh);
      }
    }
    a_sse = wave_tree64(a_sse); a_ll = wave_tree64(a_ll); a_kl = wave_tree64(a_kl);
    if (lane == 0) { d.colsse[g] = a_sse; d.colll[g] = a_ll; d.colkl[g] = a_kl; }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- E side, K <= 128: SEVERAL columns per wave, GW = 16 or 32 lanes per column ----
// k_mh_ecol spends a whole wave on one column: with K = 96 the second row pass runs half empty and the per-factor
// scalar work (conditional parameters, truncated-normal draw, MH decision) is done by 64 lanes for one column.
// Here a group of GW lanes owns a column: lane j holds rows j, j+GW, ... in registers.  The canonical W = 64 sum over
// rows is kept: accumulator i = row mod 64 belongs to lane i mod GW, slot i / GW; the tree steps that stay inside the
// lane are plain adds, h = 16 of a 32-lane group is a v_permlane16_swap, h = 8..1 DPP row shifts.
constexpr int MHE16_KMAX = 128;
template <int GW>
BNMF_DEV double grp_tree(const double (&acc)[64 / GW]) {  // lane 0 of every GW-lane group: the W = 64 halving tree
  double v;
  if (GW == 16) v = (acc[0] + acc[2 % (64 / GW)]) + (acc[1] + acc[3 % (64 / GW)]);   // h = 32: i + (i + 32); h = 16: i + (i + 16)
  else {
    v = acc[0] + acc[1];                                  // h = 32
    const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = v + __longlong_as_double(((long long)b[1] << 32) | (unsigned)a[1]);            // h = 16: lane i + lane i + 16
  }
#define BNMF_ROW_STEP(CTRL)                                                                                        \
  {                                                                                                                \
    int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32);                              \
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true); \
    v = v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);                                            \
  }
  BNMF_ROW_STEP(0x108) BNMF_ROW_STEP(0x104) BNMF_ROW_STEP(0x102) BNMF_ROW_STEP(0x101)
#undef BNMF_ROW_STEP
  return v;
}
template <int GW>
BNMF_DEV double grp_bcast0(double v, int lane) {          // lane 0 of the group to all its lanes
  int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32);
  if (GW == 16) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x150, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150, 0xf, 0xf, false); }   // row_newbcast:0
  else {
    const int l0 = __builtin_amdgcn_readlane(lo, 0), h0 = __builtin_amdgcn_readlane(hi, 0), l1 = __builtin_amdgcn_readlane(lo, 32), h1 = __builtin_amdgcn_readlane(hi, 32);
    lo = lane < 32 ? l0 : l1; hi = lane < 32 ? h0 : h1;
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// KM: the rows the lanes' register arrays are sized for — 128, or 96 (the 96 trinucleotide contexts: a quarter fewer registers and no empty
// rounds over rows 96..127; the kernel sits at the edge of two waves per SIMD)
template <bool METRICS_ONLY, bool MHSTEP, int GW, int KM = MHE16_KMAX>
__global__ __launch_bounds__(MHE_T) void k_mh_ecol16(Dev d, uint32_t t, const int* nzP, double* accE, int draw_sig, int* nzE_set, int ncolblk, MhPTail pt) {
  constexpr int MHE16_RPL = KM / GW;                      // rows per lane
  constexpr int NS = 64 / GW;                             // accumulator slots per lane
  constexpr int CPW = 64 / GW;                            // columns per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!METRICS_ONLY && (int)blockIdx.x >= ncolblk) {      // hosted: the P side of what followed the sweeps (behind the column blocks in the grid)
    if (!METRICS_ONLY) mh_ptail_body(d, pt, (int)blockIdx.x - ncolblk, tid, (double*)smem);
    return;
  }
  const int j = lane % GW, grp = lane / GW;
  const int K = d.K, G = d.G, N = d.N;
  double* ec = (double*)smem + (size_t)(wave * CPW + grp) * N * (1 + PRE_W);   // [N] current column of E, one per group
  double* prq = ec + N;                                       // [N][PRE_W] the column's DrawPre
  // A[n] and the all(P[,n] == 0) flags of the workgroup: read from LDS inside the factor chain (a scalar load from memory in front of
  // every step was the chain's longest link, as in the row kernel)
  double* anz = (double*)smem + (size_t)(MHE_T / 64) * CPW * N * (1 + PRE_W);   // [2][N]
  for (int i = tid; i < N; i += MHE_T) { anz[i] = d.A[i]; anz[N + i] = (nzP && nzP[i] == 0) ? 1.0 : 0.0; }
  __syncthreads();
  const double LOG1 = dlog(1.0);
  const bool normal = d.likelihood == BNMF_NORMAL;
  const int ngrp = (G + CPW - 1) / CPW;                       // sets of CPW columns
  for (int gq = blockIdx.x * (MHE_T / 64) + wave; gq < ngrp; gq += ncolblk * (MHE_T / 64)) {
    MHSTAMP(S0);
    const int g = gq * CPW + grp;
    const bool live = g < G;                                  // a row beyond G works on column G - 1 and writes nothing
    const int gc = live ? g : G - 1;
    for (int i = j; i < N; i += GW) ec[i] = d.E[i + (size_t)N * gc];
    if (!METRICS_ONLY) for (int i = j; i < N; i += GW) pre_store(prq + PRE_W * i, draw_pre<1>(d, i + N * gc, t, MHSTEP));   // one factor per lane
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double sg_col = normal ? d.sigmasq[gc] : 1.0;
    double mh[MHE16_RPL], l0[MHSTEP ? MHE16_RPL : 1], l1[MHSTEP ? MHE16_RPL : 1];
    int mr[MHE16_RPL];
#pragma unroll
    for (int r = 0; r < MHE16_RPL; ++r) { const int kk = j + GW * r; mr[r] = kk < K ? d.M[kk + (size_t)K * gc] : 0; mh[r] = 0.0; }
    if (!METRICS_ONLY) {
      // fresh Mhat of the column: factor by factor with the lane's rows side by side, every load made (a row beyond K reads row K - 1 and is
      // zeroed afterwards; each row's sum in factor order), see k_mh_prow — in the forms that keep two waves per SIMD with it (the others go
      // to 256 registers and one wave); else row by row
      constexpr bool SIDE_BY_SIDE = GW == 32 || (KM == 96 && !MHSTEP);
      if (!SIDE_BY_SIDE) {
#pragma unroll
        for (int r = 0; r < MHE16_RPL; ++r) {
          const int kk = j + GW * r;
          if (kk < K) {
            double c = 0.0;
            for (int q = 0; q < N; ++q) c = c + (d.P[kk + (size_t)K * q] * anz[q]) * ec[q];
            mh[r] = c;
            if (MHSTEP) l0[MHSTEP ? r : 0] = mh_log_clamped(c);
          }
        }
      } else {
        int kr[MHE16_RPL];
#pragma unroll
        for (int r = 0; r < MHE16_RPL; ++r) kr[r] = min(j + GW * r, K - 1);
#pragma unroll 1
        for (int q = 0; q < N; ++q) {
          const double aq = anz[q], eq = ec[q];
          const double* Pq = d.P + (size_t)K * q;
#pragma unroll
          for (int r = 0; r < MHE16_RPL; ++r) mh[r] = mh[r] + (Pq[kr[r]] * aq) * eq;
        }
#pragma unroll
        for (int r = 0; r < MHE16_RPL; ++r) {
          if (j + GW * r < K) { if (MHSTEP) l0[MHSTEP ? r : 0] = mh_log_clamped(mh[r]); }
          else mh[r] = 0.0;
        }
      }
      double pnx[MHE16_RPL];                                  // the lane's rows of the NEXT factor's column of P: loaded a step ahead
#pragma unroll
      for (int r = 0; r < MHE16_RPL; ++r) { const int kk = j + GW * r; pnx[r] = kk < K ? d.P[kk] : 0.0; }
      MHSTAMP(S1);
#ifdef ZSPROF
      if (blockIdx.x == 0 && tid == 0) { g_drprof[8 * (2048 + 128)] += mhsS1 - mhsS0; g_drprof[8 * (2048 + 128) + 7] += 1ull; }
#endif
      for (int n = 0; n < N; ++n) {
        MHSTAMP(E0);
        const int e = n + N * gc;
        const double a_n = anz[n];
        double pn[MHE16_RPL];
#pragma unroll
        for (int r = 0; r < MHE16_RPL; ++r) pn[r] = pnx[r];
        if (n + 1 < N) {
          const double* Pq = d.P + (size_t)K * (n + 1);
#pragma unroll
          for (int r = 0; r < MHE16_RPL; ++r) { const int kk = j + GW * r; if (kk < K) pnx[r] = Pq[kk]; }
        }
        if (a_n == 0.0) {                                                                  // sample_En :12
          const double x = prior_draw<1>(d, e, t);
          if (j == 0) { ec[n] = x; if (live) d.E[e] = x; }
          continue;
        }
        const bool allzero = anz[N + n] != 0.0;
        const double eold = ec[n];
        double s1 = 0.0, s2 = 0.0;
        if (!allzero) {
          double a1[NS] = {}, a2[NS] = {};
#pragma unroll
          for (int r = 0; r < MHE16_RPL; ++r) {
            if (j + GW * r < K) {
              const double mno = mh[r] - (pn[r] * a_n) * eold;
              const double V = normal ? sg_col : mh[r];
              const double rV = 1.0 / V;
              a1[r % NS] = a1[r % NS] + pn[r] * (((double)mr[r] - mno) * rV);
              a2[r % NS] = a2[r % NS] + (a_n * (pn[r] * pn[r])) * rV;
            }
          }
          MHSTAMP(E1a);
          s1 = grp_bcast0<GW>(grp_tree<GW>(a1), lane); s2 = grp_bcast0<GW>(grp_tree<GW>(a2), lane);
#ifdef ZSPROF
          if (blockIdx.x == 0 && tid == 0) { g_drprof[8 * (2048 + 64 + n)] += mhsE1a - mhsE0; }
#endif
        }
        MHSTAMP(E2);
        const double pr = mh_prior_or_cond_pre<1>(d, e, t, allzero, s1, s2, pre_load(prq + PRE_W * n));
        MHSTAMP(E3);
        bool take = true;
        if (MHSTEP) {
          double tA[NS] = {}, tB[NS] = {}, tC[NS] = {}, tD[NS] = {};
#pragma unroll
          for (int r = 0; r < MHE16_RPL; ++r) {
            if (j + GW * r < K) {
              const double pna = pn[r] * a_n;
              const double m0 = mh[r], m1 = (m0 - pna * eold) + pna * pr;
              const int m = mr[r];
              const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
              const MhTerms tm = mh_cell_terms(m, m0, m1, l0[MHSTEP ? r : 0], d.lgfact[mi], LOG1);
              l1[MHSTEP ? r : 0] = tm.L1;
              tA[r % NS] = tA[r % NS] + tm.pn; tB[r % NS] = tB[r % NS] + tm.nold; tC[r % NS] = tC[r % NS] + tm.po; tD[r % NS] = tD[r % NS] + tm.nnew;
            }
          }
          const double A_ = grp_bcast0<GW>(grp_tree<GW>(tA), lane), B_ = grp_bcast0<GW>(grp_tree<GW>(tB), lane);
          const double C_ = grp_bcast0<GW>(grp_tree<GW>(tC), lane), D_ = grp_bcast0<GW>(grp_tree<GW>(tD), lane);
          double ratio = dexp((A_ + B_) - (C_ + D_));
          if (ratio > 1.0) ratio = 1.0;
          if (j == 0 && live) accE[e] = ratio;
          take = prq[PRE_W * n + 5] < ratio;                                               // runif of stream (MHU_E, e, t)
        } else if (j == 0 && live && accE) accE[e] = 1.0;
        if (take) {
#pragma unroll
          for (int r = 0; r < MHE16_RPL; ++r) {
            if (j + GW * r < K) { const double pna = pn[r] * a_n; mh[r] = (mh[r] - pna * eold) + pna * pr; if (MHSTEP) l0[MHSTEP ? r : 0] = l1[MHSTEP ? r : 0]; }
          }
          if (j == 0) { ec[n] = pr; if (live) d.E[e] = pr; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef ZSPROF
        { MHSTAMP(E4);
          if (blockIdx.x == 0 && tid == 0) { unsigned long long* o = &g_drprof[8 * (2048 + 64 + n)]; o[1] += mhsE2 - mhsE0; o[2] += mhsE3 - mhsE2; o[3] += mhsE4 - mhsE3; o[5] += mhsE4 - mhsE0; o[7] += 1ull; } }
#endif
      }
    }
    // (round 5) what the NEXT row sweep needs of the finished column: its row of the transpose and the all(E[n, ] == 0) flags (k_mh_tail's
    // Et blocks counted the non-zero entries; the sweeps only ask whether there are any)
    if (!METRICS_ONLY && nzE_set && live)
      for (int i = j; i < N; i += GW) { const double x = ec[i]; d.Et[g + (size_t)G * i] = x; if (x != 0.0) nzE_set[i] = 1; }
    // fresh Mhat of the (updated) column: residuals for sigmasq, then the metric terms (R/utils.R:412-471)
    double cfresh[MHE16_RPL];
#pragma unroll
    for (int r = 0; r < MHE16_RPL; ++r) {
      const int kk = j + GW * r;
      cfresh[r] = 0.0;
      if (kk < K) {
        double c = 0.0;
        for (int q = 0; q < N; ++q) c = c + (d.P[kk + (size_t)K * q] * anz[q]) * ec[q];
        cfresh[r] = c;
      }
    }
    if (normal && draw_sig) {
      // sample_sigmasq R/sample_params.R:275-286: sigmasq_g ~ InvGamma(Alpha_g + K/2, Beta_g + sum_k resid^2 / 2)
      double sa[NS] = {};
#pragma unroll
      for (int r = 0; r < MHE16_RPL; ++r) if (j + GW * r < K) { const double rr = (double)mr[r] - cfresh[r]; sa[r % NS] = sa[r % NS] + rr * rr; }
      const double ss = grp_bcast0<GW>(grp_tree<GW>(sa), lane);
      Stream s(d.k0, d.k1, BNMF_V_SIGMASQ, (uint32_t)gc, t);
      sg_col = rinvgamma(s, hy(d.hAlphaS, gc) + (double)K / 2.0, hy(d.hBetaS, gc) + 0.5 * ss);
      if (j == 0 && live) d.sigmasq[g] = sg_col;
    }
    double qs[NS] = {}, ql[NS] = {}, qk[NS] = {};
#pragma unroll
    for (int r = 0; r < MHE16_RPL; ++r) {
      if (j + GW * r < K) {
        const double c = cfresh[r];
        const int m = mr[r];
        const double dd = c - (double)m;
        const double mhv = c < 1e-6 ? 1e-6 : c;
        const double lmh = dlog(mhv);
        const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
        const double mt = m < 1 ? 1e-6 : (double)m;
        qs[r % NS] = qs[r % NS] + dd * dd;
        if (normal) ql[r % NS] = ql[r % NS] + dnorm_log_sd((double)m, c, sg_col);           // get_loglik_ normal branch R/utils.R:72-97
        else ql[r % NS] = ql[r % NS] + (((double)m * lmh - mhv) - d.lgfact[mi]);
        qk[r % NS] = qk[r % NS] + mt * (d.logm[mi] - lmh);
      }
    }
    const double a_sse = grp_tree<GW>(qs), a_ll = grp_tree<GW>(ql), a_kl = grp_tree<GW>(qk);
    if (j == 0 && live) { d.colsse[g] = a_sse; d.colll[g] = a_ll; d.colkl[g] = a_kl; }
    __builtin_amdgcn_wave_barrier();
  }
}

// log-prior and mean-acceptance partial sums for the models whose P/E are not drawn by k_pdraw/k_edraw
__global__ void k_lp_p(Dev d, uint32_t t, const double* accP, double* accPn) {
  const int n = blockIdx.x, lane = threadIdx.x;      // 64 lanes
  double a = 0.0, b = 0.0;
  for (int k = lane; k < d.K; k += 64) { const int e = k + d.K * n; a = a + prior_logdens<0>(d, e, d.P[e], t); if (accP) b = b + accP[e]; }
  a = wave_tree64(a); b = wave_tree64(b);
  if (lane == 0) { d.lpPn[n] = a; if (accPn) accPn[n] = b; }
}
__global__ __launch_bounds__(ES_T) void k_lp_e(Dev d, uint32_t t, const double* accE, double* accE_part) {
  __shared__ double buf[ES_T];
  const int tid = threadIdx.x;
  const long e = (long)blockIdx.x * ES_T + tid;
  double lp = 0.0, ac = 0.0;
  if (e < (long)d.lenE) {
    lp = prior_logdens<1>(d, (int)e, d.E[e], t);
    if (accE) ac = (d.A[e % d.N] == 1.0) ? accE[e] : 0.0;
  }
  const double r = block_tree<ES_T>(lp, buf, tid);
  if (tid == 0) d.lpE_part[blockIdx.x] = r;
  if (accE) {
    __syncthreads();
    const double r2 = block_tree<ES_T>(ac, buf, tid);
    if (tid == 0) accE_part[blockIdx.x] = r2;
  }
}

// ---- k_mh_tail: what follows the P / E sweeps of the MH / Normal models, in ONE launch (they were five) ----
//   blocks [0, N)                 : log-prior of column n of P and its acceptance sum (k_lp_p)
//   blocks [N, N + nblkE)         : log-prior / acceptance partial sums of 256 elements of E (k_lp_e)
//   blocks [N + nblkE, 2N + nblkE): for the NEXT iteration's P sweep: Et = transpose of E, nzE[n] = number of non-zero
//                                   entries of row n of E (k_mh_nz), and nzP[n] = 0 (k_mh_prow accumulates it)
//   blocks [2N + nblkE, ...)      : record_sample of the iteration (k_record's copy into the rings), when ra.n > 0
//   the last blocks (rs.on)       : k_reduce's work for the iteration BEFORE (its partial sums lie in another slot): on a side stream it
//                                   was ordered behind its inputs only, and nothing kept the kernels of three iterations on — which write
//                                   its slot again — behind it once the hyper sweep had moved to the main stream (round 4)
//   the FIRST blocks (sx.n > 0)   : k_side's work for the NEXT iteration (Esum, both hyper sweeps: they read P_t, E_t and the prior parameters of t, and
//                                   write the other slot) — the two kernels ran one behind the other on the main stream, each too small for the device
struct SideInTail { int n, nbP; uint32_t t; RecDst rec; };
static_assert(ES_T == RT, "k_mh_tail: one block size for its own blocks, k_side's and k_reduce's");
__global__ __launch_bounds__(ES_T) void k_mh_tail(Dev d, uint32_t t, const double* accP, double* accPn, const double* accE, double* accE_part, int* nzE, int* nzP, int nblkE, RecArgs ra,
                                                   int nrec, RedSlots rs, SideInTail sx) {
  __shared__ double buf[ES_T];
  __shared__ int cnt;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < sx.n) { side_body(d, sx.t, sx.nbP, blockIdx.x, sx.n, sx.rec, SideDone{}, buf, tid); return; }
  const int blk = (int)blockIdx.x - sx.n;
  if (blk >= 2 * d.N + nblkE + nrec) { reduce_body(d, rs, nblkE, blk - (2 * d.N + nblkE + nrec), buf, tid); return; }
  if (blk < d.N) {
    if (tid >= 64) return;
    const int n = blk;
    double a = 0.0, b = 0.0;
    for (int k = tid; k < d.K; k += 64) { const int e = k + d.K * n; a = a + prior_logdens<0>(d, e, d.P[e], t); if (accP) b = b + accP[e]; }
    a = wave_tree64(a); b = wave_tree64(b);
    if (tid == 0) { d.lpPn[n] = a; if (accPn) accPn[n] = b; }
  } else if (blk < d.N + nblkE) {
    const int be = blk - d.N;
    const long e = (long)be * ES_T + tid;
    double lp = 0.0, ac = 0.0;
    if (e < (long)d.lenE) {
      lp = prior_logdens<1>(d, (int)e, d.E[e], t);
      if (accE) ac = (d.A[e % d.N] == 1.0) ? accE[e] : 0.0;
    }
    const double r = block_tree<ES_T>(lp, buf, tid);
    if (tid == 0) d.lpE_part[be] = r;
    if (accE) {
      __syncthreads();
      const double r2 = block_tree<ES_T>(ac, buf, tid);
      if (tid == 0) accE_part[be] = r2;
    }
  } else if (blk >= 2 * d.N + nblkE) {
    const size_t rt = (size_t)(blk - 2 * d.N - nblkE) * ES_T + tid, nth = (size_t)nrec * ES_T;
    for (int j = 0; j < ra.n; ++j)
      for (size_t i = rt; i < ra.len[j]; i += nth) ra.dst[j][i] = ra.src[j][i];
    if (rt == 0 && ra.Rdst) *ra.Rdst = (double)*ra.R;
  } else {
    const int n = blk - d.N - nblkE;
    if (tid == 0) cnt = 0;
    __syncthreads();
    int c = 0;
    for (int g = tid; g < d.G; g += ES_T) {
      const double e = d.E[n + (size_t)d.N * g];
      d.Et[g + (size_t)d.G * n] = e;
      c += e != 0.0 ? 1 : 0;
    }
    if (c) atomicAdd(&cnt, c);
    __syncthreads();
    if (tid == 0) { nzE[n] = cnt; nzP[n] = 0; }
  }
}

// constructor draws of the truncated-normal prior parameters (R/sample_priors.R:32-61)
template <int SIDE>
__global__ void k_init_tn(Dev d, double* x, int is_mu, uint32_t var, const int* redraw) {
  const long len = SIDE ? (long)d.lenE : (long)d.lenP;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  const int n = SIDE ? (int)(e % d.N) : (int)(e / d.K);
  if (!redraw[n]) return;
  Stream s(d.k0, d.k1, var, (uint32_t)e, 0u);
  if (is_mu) x[e] = hy(SIDE ? d.hM_e : d.hM_p, (int)e) + dsqrt(hy(SIDE ? d.hS_e : d.hS_p, (int)e)) * rnorm_std(s);
  else x[e] = rinvgamma(s, hy(SIDE ? d.hA_e : d.hA_p, (int)e), hy(SIDE ? d.hB_e : d.hB_p, (int)e));
}

}  // namespace bnmf
