// bayesnmf_amd/csrc/zalloc_tile.h — k_zalloc_tile: the Z-allocation kernel for N > 24 (and for N <= 24 when K is
// too large for the register kernel's per-column list).
//
// Same stream spec and bit-identical results as k_zalloc (kernels.h); different machine mapping.  k_zalloc gives a
// wave a whole column: with N = 50..100 its threshold table plus a private count table is 24-60 KB of LDS per
// wave, i.e. 2-4 waves per CU, and every cell re-reads its row of P from global memory twice.  Here a WORKGROUP owns
// a chunk of ZTR = 32 rows of M for a slice of the columns:
//   * the chunk's rows of P are staged in LDS once per workgroup (fp64, [n][row]) and the chunk's ZsumG counts
//     accumulate in one LDS table shared by the workgroup (flushed once with global integer atomics);
//   * a wave takes one column at a time: a tile of 32 cells, threshold table (N-1) x 32 words;
//     a wave works on PAIRS of columns: pass A of phase 1 (Mhat, lane = row: a sequential sum per cell) runs for both
//     at once, one per half-wave; pass B (thresholds) and phase 2 then take the two columns in turn.  In pass B the two
//     half-waves take the two halves of the factors, the upper one continuing from the partial sum pass A recorded, so
//     the sequential sum of the spec is kept; phase 2 (lane = contiguous quad range) is k_zalloc's search;
//   * the per-factor totals of a tile come from a packed 8-bit histogram with 8 replicas (0.9 KB instead of the
//     6.8 KB of one row per lane): N = 100 fits 8 waves per CU;
//   * ZsumK[:, g] is accumulated across the row chunks with coalesced global integer atomics (zeroed by the host before
//     the launch), Mhat[k, g] is written out and the per-column metric terms are formed by k_colmetrics afterwards in
//     the canonical order (lane l adds rows l, l + 64, ...; wave tree), exactly as the one-wave-per-column kernels do.
// (sample_Zkg R/sample_params.R:253-265; metrics R/utils.R:412-471)
#pragma once

namespace bnmf {

constexpr int ZTR = 32;            // rows per tile = pitch of the [n][row] LDS arrays of a tile
constexpr int ZTHR = 8;            // replicas of the packed bucket histogram (lane l uses replica l & 7)
constexpr int ZTHP = 9;            // pitch of a histogram row
constexpr int ZTSUB = 7;           // quads per lane between histogram flushes: 8 lanes x 7 x 4 counts <= 255 per 8-bit field
struct ZTGeom { int HW, nch, nslice, slab_words, zacc_words, p_words; unsigned long long* dbg; };   // dbg: BNMF_ZTDBG diagnostics (null otherwise)

// host and device agree on the slab layout through these
constexpr int ztile_np8(int N) { return (N + 7) & ~7; }              // factors padded to whole groups of 8 (zero rows of P, zero ae)
constexpr int ztile_ae_words(int N) { return 4 * ztile_np8(N); }      // two columns (a wave works on a pair)
constexpr int ztile_hist_words(int HW) { return (HW * ZTHP + 1) & ~1; }
inline size_t ztile_slab_words(int N, int HW, bool save_Z) {
  size_t w = (size_t)ztile_ae_words(N) + (size_t)ztile_hist_words(HW) + (size_t)(N - 1) * ZTR + (ZTR + 1) + ZTR + (save_Z ? (size_t)N * ZTR : 0);
  return (w + 1) & ~(size_t)1;
}

// LEAN: single-buffered phase 1 and at most 128 VGPRs, for shapes whose LDS need allows 16 waves per CU (N <= ~50):
// four waves per SIMD hide the latency the double buffering hides at two
template <bool SAVE_Z, int ZT, bool LEAN = false>
__global__ __launch_bounds__(ZT, LEAN ? 4 : 2) void k_zalloc_tile(ZArgs d, double* __restrict__ Mhat, uint32_t t, ZTGeom zg) {
  constexpr int ZW = ZT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d.K, G = d.G, N = d.N, HW = zg.HW;
  const int NP8 = ztile_np8(N);
  double* Pc = (double*)smem;                             // [NP8][ZTR] rows k0.. of P (zero beyond N), shared by the workgroup
  uint32_t* zacc = (uint32_t*)(Pc + (size_t)zg.p_words / 2);   // [N][ZTR] ZsumG counts of the chunk
  uint32_t* slab = zacc + zg.zacc_words + (size_t)wave * zg.slab_words;
  double* ae = (double*)slab;                             // [2][NP8]  A[n] * E[n,g] of the wave's two columns (zero beyond N)
  uint32_t* hist = slab + ztile_ae_words(N);              // [HW][ZTHP] packed 8-bit bucket counts, 16 replicas
  uint32_t* thr = hist + ztile_hist_words(HW);            // [N-1][ZTR] thresholds of the tile
  uint32_t* qoff = thr + (size_t)(N - 1) * ZTR;           // [ZTR+1]  quad offsets
  int* mcnt = (int*)(qoff + ZTR + 1);                     // [ZTR]
  uint32_t* zloc = (uint32_t*)(mcnt + ZTR);               // [N][ZTR]  (SAVE_Z only)
  const int chunk = blockIdx.x % zg.nch, slice = blockIdx.x / zg.nch;
  const int k0 = chunk * ZTR, kc = min(ZTR, K - k0);
  const int gs0 = (int)((long)G * slice / zg.nslice), gs1 = (int)((long)G * (slice + 1) / zg.nslice);
  for (int i = tid; i < zg.zacc_words; i += ZT) zacc[i] = 0;
  for (int i = tid; i < NP8 * ZTR; i += ZT) { const int cl = i & (ZTR - 1), n = i / ZTR; Pc[i] = (cl < kc && n < N) ? d.P[k0 + cl + (size_t)K * n] : 0.0; }
  for (int n = N + lane; n < NP8; n += 64) { ae[n] = 0.0; ae[NP8 + n] = 0.0; }
  for (int i = lane; i < HW * ZTHP; i += 64) hist[i] = 0;
  if (SAVE_Z) for (int i = lane; i < N * ZTR; i += 64) zloc[i] = 0;
  __syncthreads();
  const int nthr = N - 1;
  // pass B of phase 1 splits the factors between the two half-waves at H (a multiple of 8, at most nthr / 2); pass A
  // records the running sum at H, so both halves continue the SAME sequential sum.  All LDS reads of phase 1 are
  // unpredicated with constant offsets (zero padding adds exactly 0.0): groups of 8 factors, double-buffered
  const int H = 8 * (nthr / 16);
  const int LB = nthr - H;                                // >= H
  uint32_t* ztarget = SAVE_Z ? zloc : zacc;
  const int cellL = lane & (ZTR - 1), half = lane >> 5;   // both half-waves mirror the tile's rows in pass A
  const bool cellok = cellL < kc;
  const double* Pl = Pc + cellL;
  unsigned long long tp[6] = {0, 0, 0, 0, 0, 0};           // diagnostics: cycles per section (only summed when zg.dbg)
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#define ZT_TIC(i) const unsigned long long tic_##i = zg.dbg ? __builtin_amdgcn_s_memtime() : 0ull
#define ZT_TOC(i) if (zg.dbg) tp[i] += __builtin_amdgcn_s_memtime() - tic_##i
  // A wave works on PAIRS of columns: pass A of phase 1 (lane = row, a sequential sum per cell) runs for both at once,
  // lanes 0..31 on column g, lanes 32..63 on column g + 1; pass B and phase 2 then take the two columns in turn.
  // Counts and exposures of the NEXT pair are requested while this one is processed.
  const double* aeh = ae + (size_t)half * NP8;            // this half-wave's column in pass A
  int mpre = 0;
  double epre[2][2] = {{0.0, 0.0}, {0.0, 0.0}};           // [column of the pair][n < 64, n >= 64]; larger N reloads below
  auto request = [&](int g) {
    const int gm = min(g + half, gs1 - 1);
    if (cellok) mpre = d.M[k0 + cellL + (size_t)K * gm];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int gs = min(g + s, gs1 - 1);
      if (lane < N) epre[s][0] = d.E[lane + (size_t)N * gs];
      if (64 + lane < N) epre[s][1] = d.E[64 + lane + (size_t)N * gs];
    }
  };
  if (gs0 + 2 * wave < gs1) request(gs0 + 2 * wave);
  for (int g = gs0 + 2 * wave; g < gs1; g += 2 * ZW) {
    const bool liveB = g + 1 < gs1;                       // an odd tail: the upper half-wave repeats column g and stores nothing
    const int gmine = (half && liveB) ? g + 1 : g;
    const int m_own = mpre;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int gs = min(g + s, gs1 - 1);
      if (lane < N) ae[s * NP8 + lane] = d.A[lane] * epre[s][0];
      if (64 + lane < N) ae[s * NP8 + 64 + lane] = d.A[64 + lane] * epre[s][1];
      for (int n = 128 + lane; n < N; n += 64) ae[s * NP8 + n] = d.A[n] * d.E[n + (size_t)N * gs];
    }
    if (g + 2 * ZW < gs1) request(g + 2 * ZW);
    wave_lds_fence();
    ZT_TIC(1);
    // ---------------- phase 1, pass A: Mhat = sum_n P[k,n] ae[n] in factor order; the LDS reads of the next 8 factors are
    // in flight while the current 8 are added
    double c_own = 0.0, ch_own = 0.0;
    int nl_own = -1;
    {
      double pa[8], aa[8], pb[8], ab[8];
      auto ld = [&](int n0, double* pv, double* av) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { pv[j] = Pl[(n0 + j) * ZTR]; av[j] = aeh[n0 + j]; }
      };
      auto acc = [&](int n0, const double* pv, const double* av) {
        if (n0 == H) ch_own = c_own;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double p = pv[j] * av[j]; c_own = c_own + p; nl_own = p > 0.0 ? n0 + j : nl_own; }
      };
      if (LEAN) {
        for (int n0 = 0; n0 < N; n0 += 8) { ld(n0, pa, aa); acc(n0, pa, aa); }
      } else {
        ld(0, pa, aa);
        for (int n0 = 0; n0 < N; n0 += 16) {
          if (n0 + 8 < N) ld(n0 + 8, pb, ab);
          acc(n0, pa, aa);
          if (n0 + 16 < N) ld(n0 + 16, pa, aa);
          if (n0 + 8 < N) acc(n0 + 8, pb, ab);
        }
      }
    }
    if (cellok && (half == 0 || liveB)) Mhat[k0 + cellL + (size_t)K * gmine] = c_own;
    ZT_TOC(1);
   for (int s = 0; s < (liveB ? 2 : 1); ++s) {             // the two columns of the pair in turn
    const int gcol = g + s;
    const double* aes = ae + (size_t)s * NP8;
    ZT_TIC(4);
    // the column's pass-A results, mirrored to both half-waves
    const double c = __shfl(c_own, cellL + 32 * s, 64), ch = __shfl(ch_own, cellL + 32 * s, 64);
    const int nl = __shfl(nl_own, cellL + 32 * s, 64), m = __shfl(m_own, cellL + 32 * s, 64);
    const bool act = cellok && c > 0.0 && m > 0 && nl >= 0;
    // ---------------- pass B: thresholds; lanes 0..31 take factors [0, H), lanes 32..63 continue from the sum at H
    if (act) {
      const double scale = 4294967296.0 / c;
      const int nb = half ? H : 0, ne = half ? nthr : H;
      double cc = half ? ch : 0.0;
      uint32_t* tcol = thr + cellL;
      double pa[8], aa[8], pb[8], ab[8];
      auto ld = [&](int i0, double* pv, double* av) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { pv[j] = Pl[(nb + i0 + j) * ZTR]; av[j] = aes[nb + i0 + j]; }
      };
      auto put = [&](int i0, const double* pv, const double* av) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = nb + i0 + j;
          cc = cc + pv[j] * av[j];
          const double tt = cc * scale;
          if (n < ne) tcol[n * ZTR] = (n >= nl || tt >= 4294967295.0) ? 0xFFFFFFFFu : (uint32_t)tt;
        }
      };
      if (LEAN) {
        for (int i0 = 0; i0 < LB; i0 += 8) { ld(i0, pa, aa); put(i0, pa, aa); }
      } else {
        ld(0, pa, aa);
        for (int i0 = 0; i0 < LB; i0 += 16) {
          if (i0 + 8 < LB) ld(i0 + 8, pb, ab);
          put(i0, pa, aa);
          if (i0 + 16 < LB) ld(i0 + 16, pa, aa);
          if (i0 + 8 < LB) put(i0 + 8, pb, ab);
        }
      }
    }
    const int q = (act && half == 0) ? (m + 3) >> 2 : 0;
    if (half == 0 && cellok) mcnt[cellL] = q > 0 ? m : 0;
    const int incl = wave_incl_scan_dpp(q);
    if (half == 0 && cellok) qoff[cellL] = (uint32_t)(incl - q);
    const int Q = __builtin_amdgcn_readlane(incl, 63);
    if (lane == 0) qoff[kc] = (uint32_t)Q;
    wave_lds_fence();
    ZT_TOC(4);
    // ---------------- phase 2: lane takes quads [q0, q1) of the tile, in sub-chunks of <= ZTSUB quads so that the packed
    // 8-bit histogram fields (shared by 4 lanes) cannot overflow
    const int per = (Q + 63) >> 6;
    for (int cbase = 0; cbase < per; cbase += ZTSUB) {
      const int q0 = min(Q, lane * per + cbase);
      const int q1 = min(Q, min(lane * per + per, q0 + ZTSUB));
      ZT_TIC(2);
      if (q0 < q1) {
        int cell;
        {  // upper_bound(qoff[0..kc], q0) - 1, branch-free
          int b = 0, len = kc + 1;
          while (len > 1) { const int hf = len >> 1; b = ((int)qoff[b + hf - 1] <= q0) ? b + hf : b; len -= hf; }
          cell = b + ((int)qoff[b] <= q0 ? 1 : 0) - 1;
        }
        int cstart = (int)qoff[cell];
        int cend = (int)qoff[cell + 1];
        int mc = mcnt[cell];
        uint32_t* hl = hist + (lane & (ZTHR - 1));
        for (int qi = q0; qi < q1; ++qi) {
          if (qi >= cend) {
            do { ++cell; cstart = cend; cend = (int)qoff[cell + 1]; } while (qi >= cend);
            mc = mcnt[cell];
          }
          const int j0 = (qi - cstart) << 2;
          const int nd = mc - j0;                          // >= 1; draws of this quad = min(4, nd)
          const u32x4 w = philox4x32_7((uint32_t)(j0 >> 2), (uint32_t)(k0 + cell) + (uint32_t)K * (uint32_t)gcol, t, BNMF_V_Z, d.k0, d.k1);
          const uint32_t* col = thr + cell;
          const uint32_t u0 = min(w.x, 0xFFFFFFFEu), u1 = min(w.y, 0xFFFFFFFEu), u2 = min(w.z, 0xFFFFFFFEu), u3 = min(w.w, 0xFFFFFFFEu);
          int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
          if (nthr > 0) {
            int len = nthr;
            while (len > 1) {                             // 4 interleaved branch-free searches
              const int hf = len >> 1, off = hf - 1;
              const uint32_t t0 = col[(b0 + off) * ZTR], t1 = col[(b1 + off) * ZTR], t2 = col[(b2 + off) * ZTR], t3 = col[(b3 + off) * ZTR];
              b0 = (t0 <= u0) ? b0 + hf : b0;
              b1 = (t1 <= u1) ? b1 + hf : b1;
              b2 = (t2 <= u2) ? b2 + hf : b2;
              b3 = (t3 <= u3) ? b3 + hf : b3;
              len -= hf;
            }
            b0 += (col[b0 * ZTR] <= u0) ? 1 : 0;
            b1 += (col[b1 * ZTR] <= u1) ? 1 : 0;
            b2 += (col[b2 * ZTR] <= u2) ? 1 : 0;
            b3 += (col[b3 * ZTR] <= u3) ? 1 : 0;
          }
          uint32_t* zc = ztarget + cell;
          atomicAdd(&zc[b0 * ZTR], 1u); atomicAdd(&hl[(b0 >> 2) * ZTHP], 1u << ((b0 & 3) << 3));
          if (nd > 1) { atomicAdd(&zc[b1 * ZTR], 1u); atomicAdd(&hl[(b1 >> 2) * ZTHP], 1u << ((b1 & 3) << 3)); }
          if (nd > 2) { atomicAdd(&zc[b2 * ZTR], 1u); atomicAdd(&hl[(b2 >> 2) * ZTHP], 1u << ((b2 & 3) << 3)); }
          if (nd > 3) { atomicAdd(&zc[b3 * ZTR], 1u); atomicAdd(&hl[(b3 >> 2) * ZTHP], 1u << ((b3 & 3) << 3)); }
        }
      }
      wave_lds_fence();
      ZT_TOC(2);
      ZT_TIC(3);
      // flush the packed histograms into the tile's share of ZsumK[:, g]: lane n sums byte (n&3) of word n>>2 over the replicas
      for (int n = lane; n < N; n += 64) {
        const uint32_t* hr = hist + (n >> 2) * ZTHP;
        const int sh = (n & 3) << 3;
        uint32_t tot = 0;
#pragma unroll
        for (int r = 0; r < ZTHR; ++r) tot += (hr[r] >> sh) & 0xFFu;
        if (tot) atomicAdd(&d.ZsumK[n + (size_t)N * gcol], (int32_t)tot);
      }
      wave_lds_fence();
      for (int i = lane; i < HW * ZTHP; i += 64) hist[i] = 0;
      wave_lds_fence();
      ZT_TOC(3);
    }
    // ---------------- Z[k0:k0+kc, :, g]
    if (SAVE_Z) {
      for (int i = lane; i < kc * N; i += 64) {            // i = cl + kc*n: coalesced Z store
        const int cl = i % kc, n = i / kc;
        const size_t a = (size_t)n * ZTR + cl;
        const uint32_t z = zloc[a];
        d.Z[k0 + cl + (size_t)K * (n + (size_t)N * gcol)] = (int32_t)z;
        if (z) { atomicAdd(&zacc[a], z); zloc[a] = 0; }
      }
      wave_lds_fence();
    }
   }
  }
  const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  for (int i = tid; i < kc * N; i += ZT) {
    const int cl = i % kc, n = i / kc;
    const uint32_t v = zacc[(size_t)n * ZTR + cl];
    if (v) atomicAdd(&d.ZsumG[k0 + cl + (size_t)K * n], (int32_t)v);
  }
  if (zg.dbg && lane == 0) {                               // [0] waves, [1..3] sections, [4] column loop, [5] whole kernel
    atomicAdd(&zg.dbg[0], 1ull); atomicAdd(&zg.dbg[1], tp[1] + tp[4]); atomicAdd(&zg.dbg[2], tp[2]); atomicAdd(&zg.dbg[3], tp[3]);
    atomicAdd(&zg.dbg[4], tk1 - tk0); atomicAdd(&zg.dbg[5], __builtin_amdgcn_s_memtime() - tk0);
  }
#undef ZT_TIC
#undef ZT_TOC
}

// per-column metric terms from Mhat (written by k_zalloc_tile) in the canonical order: wave per column, lane l adds
// rows l, l + 64, ..., then the wave tree (the same operations as the fused form in k_zalloc / k_zalloc_reg)
template <int T>
__global__ __launch_bounds__(T) void k_colmetrics(ZArgs d, const double* __restrict__ Mhat) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = d.K, g = blockIdx.x * (T / 64) + wave;
  if (g >= d.G) return;
  double a_sse = 0.0, a_ll = 0.0, a_kl = 0.0;
  for (int kk = lane; kk < K; kk += 64) {
    const int m = d.M[kk + (size_t)K * g];
    const double c = Mhat[kk + (size_t)K * g];
    const int mi = m < 0 ? 0 : (m > d.maxM ? d.maxM : m);
    const double dd = c - (double)m;
    const double mh = c < 1e-6 ? 1e-6 : c;
    const double lmh = dlog(mh);
    const double mt = m < 1 ? 1e-6 : (double)m;
    a_sse = a_sse + dd * dd;
    a_ll = a_ll + (((double)m * lmh - mh) - d.lgfact[mi]);
    a_kl = a_kl + mt * (d.logm[mi] - lmh);
  }
  a_sse = wave_tree64(a_sse); a_ll = wave_tree64(a_ll); a_kl = wave_tree64(a_kl);
  if (lane == 0) { d.colsse[g] = a_sse; d.colll[g] = a_ll; d.colkl[g] = a_kl; }
}

}  // namespace bnmf
