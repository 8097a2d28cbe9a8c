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
// Every decision needs a sum over ALL cells, so the factors are separated by a grid-wide exchange (hand-written:
// tagged-granule all-gather, bounded spin; the grid is sized to be co-resident: 512-lane workgroups, one per CU,
// which also keeps the waves' compute time even).  Stream spec: Mhat fresh at the start of the sweep, then maintained per cell; for factor n only
// the alternative state alt = Mhat -/+ P[k,n] E[n,g] is evaluated, the log-likelihood of the current state is carried.
// Per-column sums (64-strided over k + tree) are added in blocks of 8 consecutive columns (one wave owns whole
// blocks) into blkbuf[parity][b]; after the barrier every workgroup reduces the block sums canonically (W = 1024)
// and takes the same tempered Bernoulli decision.
constexpr int RK_T = 512;
constexpr int RK_W = RK_T / 64;
constexpr int RK_REP = 1;                                // copies of the granule buffers (see rank_publish)
constexpr int RK_CW = RK_W - 1;                          // waves that evaluate cells; the last wave of a workgroup gathers, sums and decides
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
  // h >= 1e-6: dlog's special cases reduce to "not finite" (dlog(inf) = inf, dlog(NaN) = NaN: h itself); for every other h the branch-free
  // dlog_fin is the same operations, hence the same bits, and twelve of them interleave
  const double lh = (h <= 1.7976931348623157e308) ? dlog_fin(h) : h;
  return ((double)m * lh - h) - lgf;
}
BNMF_DEV double rank_lgf(const Dev& d, int m) { return d.lgfact[m < 0 ? 0 : (m > d.maxM ? d.maxM : m)]; }
// All-gather of the block sums without a separate barrier: a value is published as two 8-byte granules
// {tag, low word}, {tag, high word} (one agent-scope relaxed store each: write-through, never torn), tag = a number
// unique to (iteration, phase).  The DECISION WAVE of every workgroup (below) sweeps all granules with agent-scope relaxed
// loads (which bypass its L1) until every tag matches: the data is its own flag, no fences needed.  Bounded: a time-out
// sets *err and makes every workgroup leave.
// Every workgroup reads every block sum at every factor: with ONE copy of the buffer all (179 at G = 10,000) decision waves ask the
// same few memory channels for the same 20 KB at the same time, and the gather took 4.3 us however the loads were issued (8- or
// 16-byte, one or two rounds).  So a block sum is published RK_REP times (lanes 0 .. RK_REP - 1 of the publishing wave, one copy
// each, in ONE store instruction per granule) and a workgroup reads copy blockIdx % RK_REP.
BNMF_DEV void rank_publish(unsigned long long* gran /* this lane's copy */, int b, unsigned tag, double v) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(gran + 2 * (size_t)b, ((unsigned long long)tag << 32) | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(gran + 2 * (size_t)b + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The gather and the canonical W = 1024 sum of the NB block sums by ONE wave, without LDS and without a workgroup barrier:
// lane l owns accumulators l + 64 j (j = 0..15) — accumulator i adds x[i], x[i + 1024], ... in order —, so tree levels
// 512 .. 64 are additions inside the lane (i + 512 is j + 8, ...) and levels 32 .. 1 are wave_tree64: the same additions in
// the same order as canon_sum(x, NB, 1024).  The granules of a lane's first 32 blocks are requested together (one round trip
// to the memory side; two in a row — 16 blocks at a time — took 4.1 us per factor) and only those whose tags do not match yet are asked for again.  The sum is valid on lane 0.
typedef unsigned int __attribute__((ext_vector_type(4))) rk_uv4;
template <int GB /* blocks per lane and round: 32 (NB <= 2,048: ONE round trip; more: a round per 2,048), or 24 where registers are short (NB <= 1,536 only) */>
BNMF_DEV bool rank_gather_sum(const unsigned long long* gran, int NB, unsigned tag, int* err, int lane, double& sum, unsigned& rounds) {
  static_assert(GB == 32 || GB == 24, "blocks per round");
  // a block's two granules in ONE 16-byte load (agent scope: sc1, as the 8-byte atomic loads; each granule carries its own tag, so
  // the two halves may be torn against each other).  The buffer descriptor bounds the loads: a lane beyond the last block reads
  // zeros — tag 0 is never used, the value +0.0 adds nothing — so the first round is straight-line code: the requests back to back,
  // one wait, ONE accumulated tag check (vector instructions only), the additions.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gran, 0, NB * 16, 0x00020000);
  double acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.0;
  bool bad = false;
  // (GB = 24 serves NB <= 1,536 only — one round; api.hip takes the half-block sweep, its one user, for at most 5 blocks per CU —: a second
  // round of 1,536 would start half-way through the 1,024 accumulators)
  for (int base = 0; base < NB && !bad; base += 64 * GB) {
    rk_uv4 hv[GB];
    const int nj = min(GB, (NB - base + 63) >> 6);        // wave-uniform: rounds of 64 blocks that hold a block at all
    const bool lastv = base + lane + 64 * (nj - 1) < NB;  // the lane holds a block in the last round (every lane does in the others)
    unsigned spins = 0;
    while (true) {
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        hv[j] = rk_uv4{tag, tag, tag, tag};               // (not requested: passes the check; its value is replaced by +0.0 below)
        if (j < nj) hv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + lane + 64 * j) * 16, 0, 16 /* sc1 */);
      }
      unsigned miss = 0u;
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        const unsigned m = (hv[j].y ^ tag) | (hv[j].w ^ tag);
        miss |= (j == nj - 1 && !lastv) ? 0u : m;
      }
      rounds += 1u;
      if (__builtin_amdgcn_ballot_w64(miss != 0u) == 0ull) break;
      // a block has not arrived: the whole round again (the requests cost one round trip however many they are)
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      const bool giveup = spins > RK_SPIN_LIMIT || ((spins & 1023u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
      if (__builtin_amdgcn_ballot_w64(giveup) != 0ull) { bad = true; break; }
    }
    if (bad) break;
#pragma unroll
    for (int j = 0; j < GB; ++j) {                        // accumulator l + 64 (j & 15) takes its blocks in ascending order (no block: + 0.0)
      const double x = __longlong_as_double((long long)(((unsigned long long)hv[j].z << 32) | hv[j].x));
      acc[j & 15] = acc[j & 15] + ((j < nj) ? x : 0.0);
    }
  }
  if (bad) { if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
#pragma unroll
  for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
    for (int j = 0; j < h; ++j) acc[j] = acc[j] + acc[j + h];
  sum = wave_tree64(acc[0]);
  return true;
}
// Eight canonical W = 64 trees at once (lane i adds lane i + h, h = 32 .. 1: the same additions as eight wave_tree64).  After
// the first level only half of the lanes of a column hold live values, so two columns share a register (v_permlane32_swap
// brings both columns' upper halves down in one exchange), after the second level four (v_permlane16_swap), and the DPP
// row shifts of levels 8 .. 1 work on four columns per instruction: 42 instructions instead of 160.  Results: column c of
// v[] in lane (c & 1) * 32 + ((c >> 1) & 1) * 16 of r[c >> 2].
BNMF_DEV void wave_tree64x8(const double* v, double* r) {
  auto sw32 = [](double& a, double& b) {                  // a <- {a.lo, b.lo}, b <- {a.hi, b.hi}
    const unsigned alo = (unsigned)__double_as_longlong(a), ahi = (unsigned)(__double_as_longlong(a) >> 32);
    const unsigned blo = (unsigned)__double_as_longlong(b), bhi = (unsigned)(__double_as_longlong(b) >> 32);
    const auto x = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __longlong_as_double(((long long)y[0] << 32) | (unsigned)x[0]);
    b = __longlong_as_double(((long long)y[1] << 32) | (unsigned)x[1]);
  };
  auto sw16 = [](double& a, double& b) {                  // a <- {a.r0, b.r0, a.r2, b.r2}, b <- {a.r1, b.r1, a.r3, b.r3}
    const unsigned alo = (unsigned)__double_as_longlong(a), ahi = (unsigned)(__double_as_longlong(a) >> 32);
    const unsigned blo = (unsigned)__double_as_longlong(b), bhi = (unsigned)(__double_as_longlong(b) >> 32);
    const auto x = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto y = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __longlong_as_double(((long long)y[0] << 32) | (unsigned)x[0]);
    b = __longlong_as_double(((long long)y[1] << 32) | (unsigned)x[1]);
  };
  double p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { double a = v[2 * i], b = v[2 * i + 1]; sw32(a, b); p[i] = a + b; }   // lanes 0..31: column 2i, 32..63: column 2i + 1
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    double a = p[2 * i], b = p[2 * i + 1];
    sw16(a, b);
    double q = a + b;                                     // rows: columns 4i, 4i + 2, 4i + 1, 4i + 3
#define BNMF_TREE_STEP(CTRL)                                                                                       \
    {                                                                                                              \
      int lo = (int)__double_as_longlong(q), hi = (int)(__double_as_longlong(q) >> 32);                            \
      lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true); \
      q = q + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);                                          \
    }
    BNMF_TREE_STEP(0x108) BNMF_TREE_STEP(0x104) BNMF_TREE_STEP(0x102) BNMF_TREE_STEP(0x101)   // row_shl:8, 4, 2, 1
#undef BNMF_TREE_STEP
    r[i] = q;
  }
}
// ... and four trees (a half block): column c of v[] in lane (c & 1) * 32 + (c >> 1) * 16 of the result
BNMF_DEV double wave_tree64x4(const double* v) {
  auto sw = [](double& a, double& b, bool half32) {
    const unsigned alo = (unsigned)__double_as_longlong(a), ahi = (unsigned)(__double_as_longlong(a) >> 32);
    const unsigned blo = (unsigned)__double_as_longlong(b), bhi = (unsigned)(__double_as_longlong(b) >> 32);
    if (half32) {
      const auto x = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
      const auto y = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
      a = __longlong_as_double(((long long)y[0] << 32) | (unsigned)x[0]); b = __longlong_as_double(((long long)y[1] << 32) | (unsigned)x[1]);
    } else {
      const auto x = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
      const auto y = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
      a = __longlong_as_double(((long long)y[0] << 32) | (unsigned)x[0]); b = __longlong_as_double(((long long)y[1] << 32) | (unsigned)x[1]);
    }
  };
  double a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
  sw(a0, b0, true); sw(a1, b1, true);
  double p0 = a0 + b0, p1 = a1 + b1;                      // lanes 0..31: columns 0 / 2, lanes 32..63: columns 1 / 3
  sw(p0, p1, false);
  double q = p0 + p1;                                     // rows: columns 0, 2, 1, 3
#define BNMF_TREE_STEP(CTRL)                                                                                       \
  {                                                                                                                \
    int lo = (int)__double_as_longlong(q), hi = (int)(__double_as_longlong(q) >> 32);                              \
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true); \
    q = q + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);                                            \
  }
  BNMF_TREE_STEP(0x108) BNMF_TREE_STEP(0x104) BNMF_TREE_STEP(0x102) BNMF_TREE_STEP(0x101)
#undef BNMF_TREE_STEP
  return q;
}
BNMF_DEV double lane_of(double v, int l) {                // lane l's value (wave-uniform l) through the scalar unit
  const int lo = __builtin_amdgcn_readlane((int)__double_as_longlong(v), l), hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
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

// REG (K <= 96): one block of 8 columns per COMPUTE wave (waves 0 .. RK_CW - 1), its cells in registers for the whole sweep:
// rows 0..63 of every column (slot 0: lane = row), and rows 64..95 of TWO columns per register (slot 1: lanes 0..31 column 2p,
// lanes 32..63 column 2p + 1, row 64 + (lane & 31)), so that no lane idles on a half-empty second pass; the upper half's terms
// are brought down with v_permlane32_swap before they are added to their column's accumulators (lane i = rows i, i + 64, as
// the canonical sum demands).  Otherwise a compute wave walks its blocks b = w, w + Wt, ... and Mhat lives in the global scratch.
// The last wave of every workgroup is its DECISION WAVE (round 4): it gathers the block sums of factor n, adds them
// canonically and takes the tempered Bernoulli decision WHILE the compute waves evaluate the alternative of factor n + 1
// (before: 4.1 us of evaluation, then 2.1 us of gather, then 1.2 us of tree and decision, one after the other, per factor).
// One workgroup barrier per factor hands the decision over.
// HALF (REG only; round 4): a compute wave holds HALF a block — 4 columns — and the two waves of a block chain its sum through LDS (the first
// adds its four column sums from +0.0, the second continues with its own: the same additions in the same order as one wave adding
// eight).  Ten half-block waves and the decision wave per workgroup (704 lanes, <= 168 registers): five blocks per CU instead of seven, on 250
// CUs instead of 179 at G = 10,000, and at most three half-block waves on a SIMD where two whole-block waves finished one after the other.
constexpr int RK_TH = 704, RK_CWH = 10;
template <bool REG, bool NORMAL, bool HALF = false>
__global__ __launch_bounds__(HALF ? RK_TH : RK_T) void k_rank_sweep(Dev d, uint32_t t, unsigned long long* granbuf /* [RK_REP][4][2 NB] */, int NB, int* err, double* mhg, unsigned long long* dbg,
                                                         int row /* metrics row, or -1 */, double* recA, double* recR) {
  static_assert(!HALF || REG, "half blocks: register variant only");
  constexpr int T = HALF ? RK_TH : RK_T, CW = HALF ? RK_CWH : RK_CW;   // lanes, compute waves
  constexpr int CPW = HALF ? RK_MAXC / 2 : RK_MAXC;                     // columns per compute wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double bc[2];
  __shared__ double dst[6];
  // HALF: the first half block's running sum, per block of the workgroup, and the number of the publication it belongs to.  Four slots
  // (publication number & 3): between two workgroup barriers a wave publishes at most three times — in front of the first barrier of the
  // launch: the current state, the alternative of factor 0 and the step ahead for factor 1; later: the redo of a flipped factor and the next
  // step ahead — so the first half is never more than three publications ahead of the second.  (With two slots the three publications at the
  // start could overwrite a sum the second half had not read: a time-out once in a few hundred chains, found by tools/fuzz_parity.py.)
  __shared__ double xchv[RK_CWH / 2][4];
  __shared__ unsigned xcht[RK_CWH / 2][4];
  // the wave's number in a scalar register and the lane from the execution mask: the thread index itself need not stay in a register
  // through the sweep (at 256 registers it was the one value spilled)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid = wave * 64 + lane;
  const int K = d.K, G = d.G, N = d.N;
  const int KR = (K + 63) >> 6;
  const bool decider = wave == CW;
  // block (REG) / first block (otherwise) of this wave: wave-major, so that a grid wider than NB / CW spreads the
  // blocks over all CUs
  const int wblk = HALF ? (wave >> 1) : wave, whalf = HALF ? (wave & 1) : 0;
  const int wg = wblk * gridDim.x + blockIdx.x, Wt = gridDim.x * CW;
  const int cb = wg * RK_MAXC + whalf * CPW;               // the wave's first column (REG)
  constexpr bool normal = NORMAL;
  // tags of this launch, unique over the chain: tag0 + 1 = log-likelihood of the current state, tag0 + 2 + 2n + redo = the
  // alternative of factor n (redo = 1: evaluated again after factor n-1 flipped, see the factor loop)
  const unsigned tag0 = t * (unsigned)(2 * N + 4);
  // the workgroup's own copy of A: every workgroup takes every decision itself, so A is never read across
  // workgroups inside the launch (the global A is written for the kernels that follow)
  double* Ash = (double*)smem;                           // [N]
  double* wR = Ash + N;                                  // [N+1] weights of sample_R
  double* uA = wR + (N + 1);                             // [N] the factors' uniforms (sample_An's rbinom), drawn up front by N lanes
  __shared__ int Rsh;
  for (int j = tid; j < N; j += T) { Ash[j] = d.A[j]; Stream sa(d.k0, d.k1, BNMF_V_A, (uint32_t)j, t); uA[j] = runif(sa); }
  if (tid < 2 * RK_CWH) xcht[tid >> 2][tid & 3] = 0u;
  __syncthreads();
  // sample_R :217-241 (was a launch of its own): every workgroup draws the same R from the same stream; the N+1 weights
  // are evaluated one per lane of the decision wave, then added and scanned in r order by its lane 0
  if (decider) {
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
#define RKSTAMP(i) if (dbg && lane == 0 && n < 16) dbg[(blockIdx.x * 16 + n) * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
  const size_t rstride = (size_t)4 * 2 * NB;               // granules per copy
  unsigned long long* const mycopy = granbuf + (size_t)(blockIdx.x % RK_REP) * rstride;       // the copy this workgroup reads
  unsigned long long* const pubcopy = granbuf + (size_t)(lane % RK_REP) * rstride;            // the copy this lane writes when it publishes
  auto gbuf = [&](int n) { return (size_t)((n + 1) & 3) * 2 * NB; };
  auto tagof = [&](int n, unsigned redo) { return tag0 + 2u + 2u * (unsigned)n + redo; };
  if (decider) {
    // ---------------------------------------------------------------- the decision wave
    double ll_cur = 0.0;
    // few instructions against the compute waves' long fp64 chains, but every other wave of the workgroup waits for their result
    __builtin_amdgcn_s_setprio(3);
    unsigned rounds0 = 0u;
    bool ok = rank_gather_sum<HALF ? 24 : 32>(mycopy, NB, tag0 + 1u, err, lane, ll_cur, rounds0);   // buffer 0: the current state (lane 0 carries it)
    // lane 0's state between the factors lives in LDS (dst: ll_cur, sumA, the sweep's constants): nothing but addresses stays in
    // registers across the gather, whose 32 blocks per lane are 128 of them
    if (lane == 0) {
      const double pi1 = prior_prob_1((double)Rsh, (double)N);   // written by this lane above
      double sumA = 0.0;
      for (int j = 0; j < N; ++j) sumA = sumA + Ash[j];
      dst[0] = ll_cur; dst[1] = sumA; dst[2] = dlog(1.0 - pi1); dst[3] = dlog(pi1); dst[4] = dlog((double)G); dst[5] = temp_at(d, t);
    }
    unsigned redo = 0;                                     // which publication of factor n's alternative is the valid one
    for (int n = 0; n < N; ++n) {
      RKSTAMP(0);
      double ll_alt = 0.0;
      unsigned rounds = 0u;
      if (ok) ok = rank_gather_sum<HALF ? 24 : 32>(mycopy + gbuf(n), NB, tagof(n, redo), err, lane, ll_alt, rounds);
      RKSTAMP(2);
      if (dbg && lane == 0 && n < 16) dbg[(blockIdx.x * 16 + n) * 8 + 5] = rounds;
      const double a_old = Ash[n];
      double a_new = -1.0;                                 // -1: the exchange timed out, every wave leaves
      if (ok && lane == 0) {
        const double ll_cur_ = dst[0], sumA = dst[1];
        const RankConst rc{dst[2], dst[3], dst[4], dst[5]};
        const double ll0 = (a_old == 1.0) ? ll_alt : ll_cur_, ll1 = (a_old == 1.0) ? ll_cur_ : ll_alt;
        a_new = rank_decide(d, uA[n], ll0, ll1, a_old, sumA, rc);
        if (a_new != a_old) { dst[0] = ll_alt; dst[1] = (sumA - a_old) + a_new; }
      }
      if (lane == 0) bc[n & 1] = a_new;
      RKSTAMP(3);
      wg_lds_barrier();
      a_new = wave_bcast0(a_new);
      if (a_new < 0.0) return;
      redo = (a_new != a_old) ? 1u : 0u;
      if (lane == 0) { Ash[n] = a_new; if (blockIdx.x == 0) d.A[n] = a_new; }
    }
  } else {
  // ---------------------------------------------------------------- the compute waves
  constexpr int RK_P = CPW / 2;                           // column pairs of the second row slot
  double mh0[REG ? CPW : 1], mh1[REG ? RK_P : 1], sgc[(REG && NORMAL) ? CPW : 1];
  double lg0[(REG && !NORMAL) ? CPW : 1], lg1[(REG && !NORMAL) ? RK_P : 1];   // lgamma(M + 1) of the wave's cells
  int mm0[REG ? CPW : 1], mm1[REG ? RK_P : 1];
  unsigned xc = 0u;                                        // HALF: publications so far (the same count in both waves of a block)
  bool xdead = false;                                      // HALF: a wait for the first half has timed out: the launch is lost, wait for nothing more
  const int half = lane >> 5, row1 = 64 + (lane & 31);    // slot 1: this lane's column of the pair and its row
  const bool lowv1 = lane < 32 && 64 + lane < K;          // lanes that own an accumulator with a second row
  // block sum of the wave's 8 columns from the lanes' accumulators: 8 canonical trees, then columns in ascending order (lane 0)
  auto block_sum = [&](const double* accv) {
    double bs = 0.0;
    if (!HALF) {
      double r[2];
      wave_tree64x8(accv, r);
#pragma unroll
      for (int c = 0; c < RK_MAXC; ++c) {
        const double tr = lane_of(r[c >> 2], (c & 1) * 32 + ((c >> 1) & 1) * 16);
        bs = (cb + c < G) ? bs + tr : bs;
      }
    } else {
      const double r = wave_tree64x4(accv);
      ++xc;
      if (whalf) {                                         // the second half continues the first half's sum
        unsigned spins = 0;
        // bounded like every wait of the sweep: a first half that never arrives (it cannot, short of a fault) fails the launch, not the chain's bits.
        // after ONE time-out the wave waits for nothing any more (xdead) — the launch is lost (the host poisons the handle), every
        // later publication of this block would otherwise sit through the same bound again — and takes +0.0 instead of a slot that may be stale.
        while (!xdead && __hip_atomic_load(&xcht[wblk][xc & 3u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != xc) {
          if (++spins >= (1u << 26)) { xdead = true; if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          __builtin_amdgcn_s_sleep(0);
        }
        bs = xdead ? 0.0 : xchv[wblk][xc & 3u];
      }
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const double tr = lane_of(r, (c & 1) * 32 + (c >> 1) * 16);
        bs = (cb + c < G) ? bs + tr : bs;
      }
      if (!whalf && lane == 0) {
        xchv[wblk][xc & 3u] = bs;
        __hip_atomic_store(&xcht[wblk][xc & 3u], xc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    return bs;
  };
  // ---- phase 0: fresh Mhat and the log-likelihood of the current state
  unsigned long long* gran = pubcopy;                    // buffer 0 of 4
  const unsigned phase = 1;
  if (REG) {
    double accv[CPW];
    auto fresh = [&](int kk, int g, double sg, double& mhv, int& mv, double& lgv) {   // Mhat, M, lgamma(M+1) and the cell's term
      double cc = 0.0;
      for (int j = 0; j < N; ++j) cc = cc + (d.P[kk + (size_t)K * j] * Ash[j]) * d.E[j + (size_t)N * g];
      const int m = d.M[kk + (size_t)K * g];
      const double lgf = rank_lgf(d, m);
      mhv = cc; mv = m; lgv = lgf;
      return rank_cell_ll(d, m, cc, sg, lgf);
    };
#pragma unroll
    for (int c = 0; c < CPW; ++c) {                       // slot 0: rows 0..63 (cells beyond G / K: harmless values, never added)
      const int g = cb + c;
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
      const int g = cb + 2 * p + half;
      mh1[p] = 0.0; mm1[p] = 0;
      double lgv = 0.0, v = 0.0;
      if (g < G && row1 < K) v = fresh(row1, g, normal ? d.sigmasq[g] : 1.0, mh1[p], mm1[p], lgv);
      if (!NORMAL) lg1[NORMAL ? 0 : p] = lgv;
      const double w = down32(v);
      if (lowv1 && cb + 2 * p < G) accv[2 * p] = accv[2 * p] + v;
      if (lowv1 && cb + 2 * p + 1 < G) accv[2 * p + 1] = accv[2 * p + 1] + w;
    }
    const double bs = block_sum(accv);
    if (lane < RK_REP && wg < NB && (!HALF || whalf)) rank_publish(gran, wg, tag0 + phase, bs);
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
      bs = wave_bcast0(bs);
      if (lane < RK_REP) rank_publish(gran, b, tag0 + phase, bs);
    }
  }
  // ---- factors in order, one step ahead of the decisions.  The alternative log-likelihood of factor n+1 depends on the
  // decision for factor n only through Mhat, and most decisions leave A[n] as it was.  So the compute waves evaluate and
  // publish the alternative of factor n+1 WHILE the decision wave gathers the block sums of factor n (published one step
  // earlier by every workgroup) and decides.  When factor n does flip, Mhat is updated and the alternative of n+1 is
  // evaluated and published again under the `redo` tag; every workgroup takes the same decision, so all of them know which
  // tag to gather.  Same values, same order of operations, same draws as the one-factor-at-a-time sweep.
  // Four granule buffers: a workgroup in step n writes buffer (n+2)&3 while the slowest one may still read (n-1)&3 .. (n+1)&3.
  // REG: column n of P and row n of E (this wave's 8 columns) are requested one factor ahead of their use
  double np0 = 0.0, np1 = 0.0, nen[REG ? CPW : 1];
  auto prefetch = [&](int n) {
    const double* Pq = d.P + (size_t)K * n;
    np0 = lane < K ? Pq[lane] : 0.0; np1 = row1 < K ? Pq[row1] : 0.0;   // slot 1: both half-waves hold rows 64 + (lane & 31)
#pragma unroll
    for (int c = 0; c < (REG ? CPW : 1); ++c) { const int g = cb + c; nen[c] = g < G ? d.E[n + (size_t)N * g] : 0.0; }
  };
  double p0 = 0.0, p1 = 0.0, en_[REG ? CPW : 1];          // REG: the factor whose alternative is evaluated next
  auto take_prefetched = [&](int n_next) {                 // current <- prefetched, request factor n_next
    p0 = np0; p1 = np1;
#pragma unroll
    for (int c = 0; c < (REG ? CPW : 1); ++c) en_[c] = nen[c];
    if (n_next < N) prefetch(n_next);
  };
  // evaluate the alternative of factor f (A[f] flipped) on the current Mhat and publish this wave's block sums
  auto publish_alt = [&](int f, unsigned redo) {
    const double a_f = Ash[f];
    unsigned long long* gr = pubcopy + gbuf(f);
    const unsigned tg = tagof(f, redo);
    if (REG) {
      // straight-line code: the 12 cell terms are independent chains the scheduler can interleave; cells beyond K and
      // columns beyond G hold harmless values and are never added
      // Mhat - t is Mhat + (-t), and (-p) e is -(p e): the sign goes onto the column of P once per factor
      const double q0 = (a_f == 1.0) ? -p0 : p0, q1 = (a_f == 1.0) ? -p1 : p1;
      double accv[CPW];
#pragma unroll
      for (int c = 0; c < CPW; ++c) {                     // slot 0
        const double alt = mh0[c] + q0 * en_[c];
        const double ll = rank_cell_ll_t<NORMAL>(mm0[c], alt, NORMAL ? sgc[NORMAL ? c : 0] : 1.0, NORMAL ? 0.0 : lg0[NORMAL ? 0 : c]);
        accv[c] = (lane < K) ? 0.0 + ll : 0.0;
      }
#pragma unroll
      for (int p = 0; p < RK_P; ++p) {                    // slot 1: two columns per register
        const double alt = mh1[p] + q1 * (half ? en_[2 * p + 1] : en_[2 * p]);
        const double sg = NORMAL ? (half ? sgc[NORMAL ? 2 * p + 1 : 0] : sgc[NORMAL ? 2 * p : 0]) : 1.0;
        const double v = rank_cell_ll_t<NORMAL>(mm1[p], alt, sg, NORMAL ? 0.0 : lg1[NORMAL ? 0 : p]);
        const double w = down32(v);
        accv[2 * p] = lowv1 ? accv[2 * p] + v : accv[2 * p];
        accv[2 * p + 1] = lowv1 ? accv[2 * p + 1] + w : accv[2 * p + 1];
      }
      const double bs = block_sum(accv);
      if (lane < RK_REP && wg < NB && (!HALF || whalf)) rank_publish(gr, wg, tg, bs);
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
        bs = wave_bcast0(bs);
        if (lane < RK_REP) rank_publish(gr, b, tg, bs);
      }
    }
  };
  // Mhat <- Mhat -/+ the term of factor f (its A flipped from a_was): the same operations as the alternative just evaluated
  auto flip_mhat = [&](int f, double a_was) {
    const double* Pf = d.P + (size_t)K * f;
    if (REG) {
      const double q0 = lane < K ? Pf[lane] : 0.0, q1 = row1 < K ? Pf[row1] : 0.0;
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int g = cb + c;
        const double tt = q0 * (g < G ? d.E[f + (size_t)N * g] : 0.0);
        mh0[c] = (a_was == 1.0) ? mh0[c] - tt : mh0[c] + tt;
      }
#pragma unroll
      for (int p = 0; p < RK_P; ++p) {
        const int g = cb + 2 * p + half;
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
  for (int n = 0; n < N; ++n) {
    const double a_old = Ash[n];
    if (n + 1 < N) {                                       // one step ahead: factor n+1 on the Mhat as it is
      if (REG) take_prefetched(n + 2);
      publish_alt(n + 1, 0u);
    }
    if (wave == 0) { RKSTAMP(1); }
    if (wave == CW - 1) { RKSTAMP(7); }                // the youngest compute wave: two waves of a SIMD finish one after the other
    wg_lds_barrier();                                      // the decision of factor n (LDS only: the prefetches and publications stay in flight)
    const double a_new = bc[n & 1];
    if (a_new < 0.0) return;                               // the exchange timed out (the decision wave has set *err)
    if (a_new != a_old) {                                  // the step ahead was taken on a stale Mhat: again
      flip_mhat(n, a_old);                                 // (general variant: a lane re-reads only the Mhat cells it wrote itself)
      if (n + 1 < N) publish_alt(n + 1, 1u);
    }
    if (wave == 0) { RKSTAMP(4); }
  }
  }
#undef RKSTAMP
  __syncthreads();
  // what k_sumA did in a launch of its own (Gibbs sweep): A, R into the ring, sum(A) into the raw metrics row
  if (blockIdx.x == 0 && row >= 0) {
    if (recA) for (int j = tid; j < N; j += T) recA[j] = Ash[j];
    if (tid == 0) {
      if (recR) *recR = (double)Rsh;
      double sA = 0.0;
      for (int j = 0; j < N; ++j) sA = sA + Ash[j];
      d.raw[(size_t)row * 8 + 5] = sA;
    }
  }
}

}  // namespace bnmf
