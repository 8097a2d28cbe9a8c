// bayesnmf_amd/csrc/zalloc_reg.h — k_zalloc_reg: the Z-allocation kernel for N <= 24 (the metric
// configuration has N = 20).
//
// Same stream spec and bit-identical results as k_zalloc (kernels.h), different machine mapping.
// Phase 1 (lane = row) appends each non-empty cell's threshold row to a compact per-column list in the
// wave's LDS slab: a header (end quad offset, row id, pivots = every 4th threshold) and the thresholds
// as 128-bit blocks.  Phase 2 walks the column's quads (4 counts = one Philox block) in contiguous
// per-lane ranges; the bucket of a count is found with a two-level search: pivot compares in registers
// pick the 4-block, ONE 128-bit LDS read fetches it, four more compares finish.  The loop is software
// pipelined (block reads of quad i in flight under the Philox of quad i+1; next cell's header
// prefetched), because with 2 waves per SIMD nothing else hides the LDS round trips.  Each count then
// costs two LDS atomics: zacc[n][k] and the lane's private packed 8-bit histogram used for ZsumK.
// (sample_Zkg R/sample_params.R:253-265; metrics R/utils.R:412-471)
#pragma once

namespace bnmf {

// -DZPROF: in-kernel section timers (s_memtime).  Selector sel = ablate >> 12 (1 phase 1, 2 range set-up,
// 3 quad loop, 4 histogram flush, 5 phase 3, 6 whole column, 7 kernel prologue on the wave's first column);
// the cycles of that section replace the column's squared-error term, so the host reads
// sum over columns = RMSE^2 K G.  Diagnostics only; never defined in the product build.
#ifdef ZPROF
#define ZTIC(i) const uint64_t ztic_##i = __builtin_amdgcn_s_memtime()
#define ZTOC(i) zprof[i] += __builtin_amdgcn_s_memtime() - ztic_##i
#else
#define ZTIC(i)
#define ZTOC(i)
#endif

constexpr int ZNMAX = 24;          // max N of the register path (<= 23 thresholds: the last slot of a row stays a pad)

// b_j += (T <= u_j) for four counts at once.  Measured on gfx950 (tools/ubench.hip): the plain C++
// form (v_cmp -> VCC -> v_addc, with the compiler's hazard nops) issues at ~2.2 cycles per
// compare-accumulate per SIMD; hand-written e64 forms through SGPR pairs are 2x SLOWER (4.2).
BNMF_DEV void cmp_acc4(uint32_t T, uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3,
                       uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
  b0 += (T <= u0) ? 1u : 0u;
  b1 += (T <= u1) ? 1u : 0u;
  b2 += (T <= u2) ? 1u : 0u;
  b3 += (T <= u3) ? 1u : 0u;
}

// full-rate 24-bit multiply (v_mul_u32_u24): operands are bucket indices / LDS pitches, far below 2^24;
// HIP's __umul24 lowers to the quarter-rate v_mul_lo_u32 here
BNMF_DEV uint32_t mul24(uint32_t a, int b) {
  uint32_t r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// LDS atomics by byte offset: address = base + index * pitch as ONE v_mad_u32_u24 (index and pitch are far below 2^24; the
// pointer form costs a multiply and a shift-add)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
BNMF_DEV uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)(lds_u32*)p; }
BNMF_DEV uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
BNMF_DEV void lds_add(uint32_t off, uint32_t v) { __hip_atomic_fetch_add((lds_u32*)(uintptr_t)off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// v_cvt_u32_f64 saturates (x >= 2^32 -> 0xFFFFFFFF, x < 0 / NaN -> 0): C's (uint32_t)x is undefined there
BNMF_DEV uint32_t cvt_u32_sat(double x) {
  uint32_t r;
  asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS round trips)
BNMF_DEV int wave_incl_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// only what this kernel needs (the full Dev by value costs ~60 SGPRs and made the hot loop spill SGPRs)
struct ZArgs {
  int K, G, N, maxM;
  uint32_t k0, k1;
  const int32_t* M;
  const double *P, *E, *A;
  int32_t *ZsumK, *ZsumG, *Z;
  double *colsse, *colll, *colkl;
  const double *lgfact, *logm;
  // gate (fixed-rank sweep, merged draw kernel): ONE lane of the launch waits, before the kernel ends, for the flags of the next
  // iteration's hyper sweep — the draw kernel behind this one then waits for nothing outside itself
  const unsigned *gate0, *gate1; unsigned gate_epoch; int* gate_err;
};
// Row geometry of the compact per-column cell list (host and device must agree: api.hip sizes the slab)
constexpr int zreg_hdr_u4(int trc) { return (trc / 4 - 1) > 4 ? 3 : 2; }           // header 128-bit words: link + pivots
constexpr int zreg_row_words(int trc) { return 4 * (zreg_hdr_u4(trc) + trc / 4); } // header + threshold blocks

template <bool SAVE_Z, int ZT, int TRC /* threshold slots: multiple of 4, >= N-1 */, bool DIAG = false /* honour `ablate` */>
__global__ __launch_bounds__(ZT, 4) void k_zalloc_reg(ZArgs d, uint32_t t, ZGeom zg, int ablate) {
  constexpr int ZW = ZT / 64;
  constexpr int NC = TRC;                                // factors covered by this instantiation: at most TRC - 1 thresholds,
                                                         // so slot TRC - 1 of every row is a pad (2^32 - 1 = "never")
  constexpr int NMIN = TRC == 8 ? 1 : TRC == 16 ? 9 : TRC - 3;    // smallest N routed here (api.hip: zg.TR)
  constexpr int NB = TRC / 4;                            // threshold blocks per cell
  constexpr int NPV = NB - 1;                            // pivots: last threshold of every block but the last
  constexpr int HB = zreg_hdr_u4(TRC);
  constexpr int RP = zreg_row_words(TRC);                // row pitch in words
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N;
  const int KP = zg.KP, HW = zg.HW;
  const int KR = (K + 63) >> 6;
  uint32_t* zacc = (uint32_t*)smem;                      // [N][KP] shared by the workgroup
  double* Pl = (double*)(zacc + zg.zacc_words);          // [N][K] P, shared by the workgroup
  uint32_t* slab = zacc + zg.zacc_words + zg.p_words + (size_t)wave * zg.slab_words;
  uint32_t* hist = slab;                                 // [HW][ZH] per-lane packed 8-bit bucket counts
  uint32_t* rows = hist + HW * ZH;                       // [K+1][RP] compact list of the column's non-empty cells:
                                                         //   u4 {cend, k | pad << 30, next cend, next k | pad}, u4 pivots
                                                         //   (+ u4 when > 4 pivots), then NB u4 threshold blocks;
                                                         //   cells in row order; + sentinel
  double* ae = (double*)(rows + (size_t)(K + 1) * RP);   // [N]  A[n] * E[n,g]
  uint32_t* zkt = (uint32_t*)(ae + N);                   // [N] column totals
  uint32_t* zloc = zkt + N;                              // [N][KP]  (SAVE_Z only)
#ifdef ZPROF
  uint64_t zprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  ZTIC(7);
  for (int i = tid; i < N * KP; i += ZT) zacc[i] = 0;
  for (int i = tid; i < K * N; i += ZT) Pl[i] = d.P[i];
  for (int i = lane; i < HW * ZH; i += 64) hist[i] = 0;
  for (int i = lane; i < N; i += 64) zkt[i] = 0;
  if (SAVE_Z) for (int i = lane; i < N * KP; i += 64) zloc[i] = 0;
  __syncthreads();
  ZTOC(7);
  const int nthr = N - 1;
  uint32_t* ztarget = SAVE_Z ? zloc : zacc;
  const int gw = blockIdx.x * ZW + wave, nw = gridDim.x * ZW;
  // Global-memory latency (~1-2 k cycles) is hidden by loading one column ahead: the counts of the first two
  // row passes and the exposures E[:,g] of the NEXT column are requested at the start of this column.
  const double a_l = lane < N ? d.A[lane] : 0.0;          // N <= ZNMAX < 64
  int mpre0 = 0, mpre1 = 0;
  double epre = 0.0;
  if (gw < G) {
    if (lane < K) mpre0 = d.M[lane + (size_t)K * gw];
    if (64 + lane < K) mpre1 = d.M[64 + lane + (size_t)K * gw];
    if (lane < N) epre = d.E[lane + (size_t)N * gw];
  }
  for (int g = gw; g < G; g += nw) {
    if (DIAG && (ablate & 32)) continue;                  // launch + prologue + epilogue only
    ZTIC(6);
#ifdef ZPROF
    const uint64_t zrt0 = __builtin_amdgcn_s_memrealtime();
#endif
    ZTIC(1);
    // ---------------- phase 1: lane = row.  Mhat, metric terms, and the cell's threshold row appended
    // to the compact list (position = number of non-empty cells before it: a wave prefix scan).
    double a_sse = 0.0, a_ll = 0.0, a_kl = 0.0;
    int Q = 0, J = 0;                                     // wave-uniform: quads and non-empty cells so far
    if (lane < N) ae[lane] = a_l * epre;
    const int mc0 = mpre0, mc1 = mpre1;
    {
      const int gn = g + nw;
      if (gn < G) {
        if (lane < K) mpre0 = d.M[lane + (size_t)K * gn];
        if (64 + lane < K) mpre1 = d.M[64 + lane + (size_t)K * gn];
        if (lane < N) epre = d.E[lane + (size_t)N * gn];
      }
    }
    wave_lds_fence();
    for (int r = 0; r < KR; ++r) {
      const int kk = (r << 6) + lane;
      int q = 0, m = 0;
      double c = 0.0;
      const double* Pk = Pl + min(kk, K - 1);             // P[kk, n] = Pl[kk + K n] (workgroup copy in LDS)
      // the row of P: requested in one go (one LDS round trip under the other waves' quad-loop traffic instead of one per group
      // of four) and kept for the threshold pass below
      double pk[NC];
#pragma unroll
      for (int n = 0; n < NC; ++n) pk[n] = (n < NMIN || n < N) ? Pk[(size_t)K * n] : 0.0;
      if (kk < K) {
        m = r == 0 ? mc0 : r == 1 ? mc1 : d.M[kk + (size_t)K * g];
        const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
        const double lgf = d.lgfact[mi], lgm = d.logm[mi];   // requested now, used after the factor loop
        // Mhat = sum_n P[k,n] (A[n] E[n,g]) in factor order
#pragma unroll
        for (int n = 0; n < NC; ++n) if (n < NMIN || n < N) c = c + pk[n] * ae[n];
        if (c > 0.0 && m > 0) q = (m + 3) >> 2;
        const double dd = c - (double)m;
        const double mh = c < 1e-6 ? 1e-6 : c;
        const double lmh = dlog(mh);
        const double mt = m < 1 ? 1e-6 : (double)m;
        a_sse = a_sse + dd * dd;                          // canonical: lane l adds rows l, l+64, ...
        a_ll = a_ll + (((double)m * lmh - mh) - lgf);
        a_kl = a_kl + mt * (lgm - lmh);
      }
      const int inclQ = wave_incl_scan_dpp(q);
      const int inclJ = wave_incl_scan_dpp(q > 0 ? 1 : 0);
      if (q > 0) {
        // thr_n = floor(cum_n 2^32 / Mhat), saturating at 2^32-1 = "never".  Factors at/after the last
        // positive one have cum_n == Mhat, and fl(Mhat * fl(2^32/Mhat)) >= 2^32 (1 - 2^-52) > 2^32 - 1,
        // so they saturate by themselves (the oracle's explicit `n >= nlast` test is the same set).
        // The running sums are recomputed (same operations in the same order, hence the same bits) rather than
        // kept: 21 live fp64 values made the compiler serialise every LDS read of the sum above behind a wait.
        const double scale = 4294967296.0 / c;
        u4* row = (u4*)(rows + (size_t)(J + inclJ - 1) * RP);
        uint32_t pvt[NB];
        double c2 = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          uint32_t tv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int n = 4 * j + i;
            tv[i] = 0xFFFFFFFFu;
            if (n < NC && (n < NMIN - 1 || n < nthr)) { c2 = c2 + pk[n < NC ? n : 0] * ae[n]; tv[i] = cvt_u32_sat(c2 * scale); }
          }
          row[HB + j] = u4{tv[0], tv[1], tv[2], tv[3]};
          pvt[j] = tv[3];
        }
        row[0] = u4{(uint32_t)(Q + inclQ), (uint32_t)kk | ((uint32_t)((4 - (m & 3)) & 3) << 30), 0u, 0u};   // .z .w: link, below
        row[1] = u4{NPV > 0 ? pvt[0] : 0u, NPV > 1 ? pvt[NPV > 1 ? 1 : 0] : 0u, NPV > 2 ? pvt[NPV > 2 ? 2 : 0] : 0u, NPV > 3 ? pvt[NPV > 3 ? 3 : 0] : 0u};
        if (HB == 3) row[2] = u4{pvt[NPV > 4 ? 4 : 0], 0u, 0u, 0u};
      }
      Q += __builtin_amdgcn_readlane(inclQ, 63);
      J += __builtin_amdgcn_readlane(inclJ, 63);
    }
    if (lane == 0) { rows[(size_t)J * RP] = 0x7FFFFFFFu; rows[(size_t)J * RP + 1] = 0u; }   // sentinel after the last cell
    wave_lds_fence();
    // link: every header also carries the (end offset, row id) of the NEXT cell, so that on a cell switch the
    // Philox counter of the new cell is known without waiting for an LDS read
    for (int j = lane; j < J; j += 64) {
      const uint32_t* nx = rows + (size_t)(j + 1) * RP;
      uint32_t* me = rows + (size_t)j * RP;
      me[2] = nx[0]; me[3] = nx[1];
    }
    a_sse = wave_tree64(a_sse); a_ll = wave_tree64(a_ll); a_kl = wave_tree64(a_kl);
    if (lane == 0) { d.colsse[g] = a_sse; d.colll[g] = a_ll; d.colkl[g] = a_kl; }
    wave_lds_fence();
    ZTOC(1);
    // ---------------- phase 2: the column's Q quads in contiguous ranges per lane (balanced whatever the
    // count distribution); chunks of <= 63 quads so that the packed 8-bit per-lane histogram cannot overflow.
    // The loop is software-pipelined: while the 128-bit threshold blocks of quad i are in flight the lane
    // generates the Philox block and pivot compares of quad i+1; the header (end offset, row id, pivots)
    // of the NEXT cell of the list is prefetched when a cell is entered, so a cell switch costs no LDS wait.
    const int per = (Q + 63) >> 6;
    for (int cbase = 0; cbase < per; cbase += 63) {
      const int q0 = min(Q, lane * per + cbase);
      const int q1 = min(Q, min(lane * per + per, q0 + 63));
      ZTIC(2);
      if (q0 < q1 && !(DIAG && (ablate & 2))) {
        int lo = 0;
        {  // first cell of the list whose end offset is > q0 (exists: q0 < Q = end of the last cell), branch-free
          int len = J;
          while (len > 1) { const int half = len >> 1; lo = ((int)rows[(size_t)(lo + half - 1) * RP] <= q0) ? lo + half : lo; len -= half; }
          lo += ((int)rows[(size_t)lo * RP] <= q0) ? 1 : 0;
        }
        int cstart = lo > 0 ? (int)rows[(size_t)(lo - 1) * RP] : 0;
        const uint32_t* rowp = rows + (size_t)lo * RP;
        const u4 h0 = *(const u4*)rowp;                    // {cend, k | pad << 30, next cend, next k | pad}
        int cend = (int)h0.x;
        uint32_t kkpad = h0.y, lnkz = h0.z, lnkw = h0.w;   // link = (end offset, row id) of the next cell of the list
        const uint32_t hlb = lds_off(hist + lane), ztb = lds_off(ztarget), KP4 = (uint32_t)KP * 4u;
        const uint32_t gK = (uint32_t)K * (uint32_t)g;
        // two LDS atomics per count: zacc[n][k] and the lane's packed histogram.  Branch-free: counts beyond the
        // cell's last one (ND < 4, only in a cell's last quad) add 0
#define ZATOM(B0, B1, B2, B3, CELL, ND)                                                                            \
        if (!(DIAG && (ablate & 4))) {                                                                             \
          const uint32_t zb_ = ztb + ((CELL) << 2);                                                                \
          const uint32_t a0_ = (ND) > 0 ? 1u : 0u, a1_ = (ND) > 1 ? 1u : 0u, a2_ = (ND) > 2 ? 1u : 0u, a3_ = (ND) > 3 ? 1u : 0u; \
          lds_add(mad24(B0, KP4, zb_), a0_); lds_add(mad24((B0) >> 2, ZH * 4, hlb), a0_ << (((B0) & 3) << 3));      \
          lds_add(mad24(B1, KP4, zb_), a1_); lds_add(mad24((B1) >> 2, ZH * 4, hlb), a1_ << (((B1) & 3) << 3));      \
          lds_add(mad24(B2, KP4, zb_), a2_); lds_add(mad24((B2) >> 2, ZH * 4, hlb), a2_ << (((B2) & 3) << 3));      \
          lds_add(mad24(B3, KP4, zb_), a3_); lds_add(mad24((B3) >> 2, ZH * 4, hlb), a3_ << (((B3) & 3) << 3));      \
        }
        u4 pa = *(const u4*)(rowp + 4);                    // pivots 0..3
        uint32_t pb = HB == 3 ? rowp[8] : 0u;              // pivot 4
        // quad -> its random words (clamped) and the 4-block of each count (pivot compares)
#define ZQUAD(QI, U0, U1, U2, U3, J0, J1, J2, J3, CELL, ND)                                                        \
        {                                                                                                          \
          const int jq_ = (QI) - cstart;                                                                           \
          CELL = kkpad & 0x3FFFFFFFu;                                                                              \
          ND = ((cend - cstart) << 2) - (int)(kkpad >> 30) - (jq_ << 2);       /* counts left in the cell: >= 1 */  \
          u32x4 w_;                                                                                                \
          if (DIAG && (ablate & 16)) w_ = u32x4{(uint32_t)(QI) * 2654435761u, (uint32_t)(QI) * 40503u, (uint32_t)(QI), ~(uint32_t)(QI)}; \
          else w_ = philox4x32_7((uint32_t)jq_, CELL + gK, t, BNMF_V_Z, d.k0, d.k1);                              \
          U0 = min(w_.x, 0xFFFFFFFEu); U1 = min(w_.y, 0xFFFFFFFEu); U2 = min(w_.z, 0xFFFFFFFEu); U3 = min(w_.w, 0xFFFFFFFEu); \
          J0 = J1 = J2 = J3 = 0;                                                                                   \
          if (NPV > 0) cmp_acc4(pa.x, U0, U1, U2, U3, J0, J1, J2, J3);                                             \
          if (NPV > 1) cmp_acc4(pa.y, U0, U1, U2, U3, J0, J1, J2, J3);                                             \
          if (NPV > 2) cmp_acc4(pa.z, U0, U1, U2, U3, J0, J1, J2, J3);                                             \
          if (NPV > 3) cmp_acc4(pa.w, U0, U1, U2, U3, J0, J1, J2, J3);                                             \
          if (NPV > 4) cmp_acc4(pb, U0, U1, U2, U3, J0, J1, J2, J3);                                               \
        }
        uint32_t u0, u1, u2, u3, j0, j1, j2, j3, cell; int nd;
        ZQUAD(q0, u0, u1, u2, u3, j0, j1, j2, j3, cell, nd);
        uint32_t pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0, pcell = cell; int pnd = 0;   // previous quad's buckets: its atomics are
                                                                                  // issued one iteration late (see (3))
        ZTOC(2);
        ZTIC(3);
        for (int qi = q0; qi < q1; ++qi) {
          // (1) this quad's threshold blocks: one 128-bit read per count, consumed in (5)
          const uint32_t* blk = rowp + 4 * HB;
          u4 k0 = u4{0, 0, 0, 0}, k1 = k0, k2 = k0, k3 = k0;
          if (!(DIAG && (ablate & 8))) {
            k0 = *(const u4*)(blk + 4 * j0); k1 = *(const u4*)(blk + 4 * j1);
            k2 = *(const u4*)(blk + 4 * j2); k3 = *(const u4*)(blk + 4 * j3);
          }
          // (2) if quad qi+1 starts the next cell of the list: its end offset and row id are already in
          // lnkz/lnkw; its own link and pivots are requested now and land under the Philox rounds
          uint32_t tz = lnkz, tw = lnkw;
          const bool more = qi + 1 < q1;
          if (more && qi + 1 >= cend) {
            cstart = cend; rowp += RP;
            // explicit moves (not PHI copies the allocator places after the loads), pinned before the loads: the
            // link registers are then dead and the loads below land in them directly, with no copy behind them
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(cend), "=&v"(kkpad) : "v"(lnkz), "v"(lnkw));
            __builtin_amdgcn_sched_barrier(0);
            tz = rowp[2]; tw = rowp[3];
            pa = *(const u4*)(rowp + 4); if (HB == 3) pb = rowp[8];
          }
          // (3) the atomics of quad qi-1, AFTER this quad's reads in LDS issue order: the LDS queue of a wave is
          // served in order, so reads issued behind 8 (bank-conflicting) atomics would wait for all of them
          ZATOM(pb0, pb1, pb2, pb3, pcell, pnd);
          // (4) quad qi+1 while all that is in flight: Philox, pivot compares
          uint32_t v0 = u0, v1 = u1, v2 = u2, v3 = u3, i0 = j0, i1 = j1, i2 = j2, i3 = j3, ncell = cell; int nnd = nd;
          if (more) ZQUAD(qi + 1, v0, v1, v2, v3, i0, i1, i2, i3, ncell, nnd);
          __builtin_amdgcn_sched_barrier(0);
          // (5) three compares per count finish the search of quad qi
          uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
          if (!(DIAG && (ablate & 8))) {
            // three, not four: the block's last slot is the pivot that already stopped the count (thresholds ascend,
            // so pivot j > u for the block j chosen), or, in the last block, the pad
            b0 = 4 * j0 + (k0.x <= u0) + (k0.y <= u0) + (k0.z <= u0);
            b1 = 4 * j1 + (k1.x <= u1) + (k1.y <= u1) + (k1.z <= u1);
            b2 = 4 * j2 + (k2.x <= u2) + (k2.y <= u2) + (k2.z <= u2);
            b3 = 4 * j3 + (k3.x <= u3) + (k3.y <= u3) + (k3.z <= u3);
          }
          if (DIAG && (ablate & 4)) { if (b0 + b1 + b2 + b3 == 0xFFFFFFF0u) ztarget[cell] = u0; }
          // (6) the link loaded in (2) is committed only here: a copy right behind the load (what the compiler
          // does for a loop-carried load result) would expose a full LDS round trip in every iteration in which
          // any lane of the wave changes cell, i.e. nearly every one
          asm volatile("" : "+v"(tz), "+v"(tw));
          lnkz = tz; lnkw = tw;
          pb0 = b0; pb1 = b1; pb2 = b2; pb3 = b3; pcell = cell; pnd = nd;
          u0 = v0; u1 = v1; u2 = v2; u3 = v3; j0 = i0; j1 = i1; j2 = i2; j3 = i3; cell = ncell; nd = nnd;
        }
        ZATOM(pb0, pb1, pb2, pb3, pcell, pnd);
#undef ZQUAD
#undef ZATOM
        ZTOC(3);
      }
      wave_lds_fence();
      ZTIC(4);
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
      ZTOC(4);
    }
    ZTIC(5);
    // ---------------- phase 3: ZsumK[:,g] (and Z[:,:,g])
    for (int n = lane; n < N; n += 64) { d.ZsumK[n + (size_t)N * g] = (int32_t)zkt[n]; zkt[n] = 0; }
    if (SAVE_Z) {
      int32_t* Zg = d.Z + (size_t)K * N * g;              // Z[:, :, g], element (kk, n) at kk + K n: coalesced over kk
      for (int n = 0; n < N; ++n) {
        for (int kk = lane; kk < K; kk += 64) {
          const size_t a = (size_t)n * KP + kk;
          const uint32_t z = zloc[a];
          Zg[kk + (size_t)K * n] = (int32_t)z;
          if (z) { atomicAdd(&zacc[a], z); zloc[a] = 0; }
        }
      }
    }
    wave_lds_fence();
    ZTOC(5);
    ZTOC(6);
#ifdef ZPROF
    zprof[0] += __builtin_amdgcn_s_memrealtime() - zrt0;   // whole column in 100 MHz ticks (selector 8 -> index 0)
#endif
#ifdef ZPROF
    if (ablate >> 12) {
      if (lane == 0) d.colsse[g] = (double)zprof[(ablate >> 12) & 7];
      for (int i = 0; i < 8; ++i) zprof[i] = 0;
    }
#endif
  }
  __syncthreads();
  for (int i = tid; i < K * N; i += ZT) {
    const int kk = i % K, n = i / K;
    const uint32_t v = zacc[(size_t)n * KP + kk];
    if (v && !(DIAG && (ablate & 1))) atomicAdd(&d.ZsumG[kk + (size_t)K * n], (int32_t)v);
  }
  if (d.gate0 && blockIdx.x == 0 && tid == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(d.gate0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < d.gate_epoch ||
           __hip_atomic_load(d.gate1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < d.gate_epoch) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) { __hip_atomic_store(d.gate_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
}

}  // namespace bnmf
